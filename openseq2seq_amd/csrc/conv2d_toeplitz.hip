// conv2d over (time, frequency) of DeepSpeech2 (tf.layers.conv2d in conv_bn_actv,
// open_seq2seq/encoders/ds2_encoder.py:252-266, kernels [11,41] s[2,2] and [11,21] s[1,2])
// expressed on the 1-D implicit-GEMM kernel: with activations flattened to
// [B, T, F*C] the convolution over frequency becomes a banded (Toeplitz) channel mixing
//   W'[kt][(fo,co)][(fi,ci)] = w[kt][fi - fo*sF + padF][ci][co]   (0 outside the band)
// and the convolution over time is the K = KT tap loop of os2s_conv1d_fwd / wgrad / dgrad.
// These two kernels build W' (bf16) from the fp32 master kernel [KT,KF,Cin,Cout] (TF
// layout) and fold the gradient of W' back onto the master gradient.
#include "os2s_common.hpp"

namespace os2s {

__global__ __launch_bounds__(256) void toeplitz_expand_kernel(
    const float* __restrict__ w, int KT, int KF, int Cin, int Cout, int Fi, int Fo, int sF, int padF,
    bf16_t* __restrict__ wexp) {
  const long long total = (long long)KT * Fo * Cout * Fi * Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    long long r = i;
    const int ci = (int)(r % Cin); r /= Cin;
    const int fi = (int)(r % Fi); r /= Fi;
    const int co = (int)(r % Cout); r /= Cout;
    const int fo = (int)(r % Fo); r /= Fo;
    const int kt = (int)r;
    const int kf = fi - fo * sF + padF;
    float v = 0.f;
    if (kf >= 0 && kf < KF) v = w[(((long long)kt * KF + kf) * Cin + ci) * Cout + co];
    wexp[i] = f2bf(v);
  }
}

// dw[kt][kf][ci][co] += sum_fo dwexp[kt][(fo,co)][(fo*sF + kf - padF, ci)]
__global__ __launch_bounds__(256) void toeplitz_reduce_kernel(
    const float* __restrict__ dwexp, int KT, int KF, int Cin, int Cout, int Fi, int Fo, int sF,
    int padF, float* __restrict__ dw) {
  const long long total = (long long)KT * KF * Cin * Cout;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    long long r = i;
    const int co = (int)(r % Cout); r /= Cout;
    const int ci = (int)(r % Cin); r /= Cin;
    const int kf = (int)(r % KF); r /= KF;
    const int kt = (int)r;
    float s = 0.f;
    for (int fo = 0; fo < Fo; ++fo) {
      const int fi = fo * sF + kf - padF;
      if (fi >= 0 && fi < Fi)
        s += dwexp[(((long long)kt * Fo + fo) * Cout + co) * ((long long)Fi * Cin) + (long long)fi * Cin + ci];
    }
    dw[i] += s;
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_conv2d_toeplitz_expand(os2s_stream_t stream, const float* w, int KT, int KF,
                                           int Cin, int Cout, int Fi, int Fo, int sF, int padF,
                                           uint16_t* wexp) {
  OS2S_REQUIRE(w && wexp && KT >= 1 && KF >= 1 && Cin >= 1 && Cout >= 1 && Fi >= 1 && Fo >= 1);
  const long long total = (long long)KT * Fo * Cout * Fi * Cin;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  OS2S_LAUNCH(toeplitz_expand_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w, KT, KF,
              Cin, Cout, Fi, Fo, sF, padF, wexp);
  return OS2S_OK;
}

extern "C" int os2s_conv2d_toeplitz_reduce(os2s_stream_t stream, const float* dwexp, int KT, int KF,
                                           int Cin, int Cout, int Fi, int Fo, int sF, int padF,
                                           float* dw) {
  OS2S_REQUIRE(dwexp && dw);
  const long long total = (long long)KT * KF * Cin * Cout;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  OS2S_LAUNCH(toeplitz_reduce_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dwexp, KT,
              KF, Cin, Cout, Fi, Fo, sF, padF, dw);
  return OS2S_OK;
}
