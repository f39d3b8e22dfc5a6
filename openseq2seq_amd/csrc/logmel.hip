// Log-mel ("logfbank") speech front end on the GPU, gfx950.
//
// Reference: get_speech_features_librosa (open_seq2seq/data/speech2text/
// speech_utils.py:322-441): gain-normalise (:354, :216-222) -> dither (:364-365) ->
// pre-emphasis 0.97 (:397, :271-272) -> librosa.core.stft(n_fft=512, hop=160,
// win=320, center=True, window=np.hanning) (:398-401) -> |.|^2 -> mel_basis . S
// (:402-406) -> log(. + 1e-20) -> per-feature mean/std over time (:411-417).
// In the reference this is NumPy on the host inside tf.py_func; here it is three
// kernels, HBM-bound by design (~0.9 KB per frame in, 128 B out):
//   1. per-utterance max|x| (gain)                 [grid-stride + atomicMax]
//   2. frames: one WAVE computes TWO frames as one complex 512-point FFT
//      (frame A -> real part, frame B -> imaginary part), radix-8 x 8 x 8 with
//      the 8-point DFTs in registers and two conflict-free LDS exchanges; the
//      window, reflect padding, gain, dither and pre-emphasis are applied while
//      gathering the samples (no intermediate signal is materialised); the two
//      power spectra are separated by Hermitian symmetry; the mel projection uses
//      a compact per-filter table (start, length, weights) — one mel bin per lane;
//   3. per-utterance, per-feature mean / std (two-pass, fp64 accumulators) +
//      normalise + bf16 store into the zero-padded [B, Tpad, F] batch.
#include "os2s_common.hpp"

namespace os2s {

constexpr int kNfft = 512;

__device__ __forceinline__ void dft4(float& r0, float& i0, float& r1, float& i1, float& r2,
                                     float& i2, float& r3, float& i3) {
  const float ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
  const float cr = r1 + r3, ci = i1 + i3, dr = r1 - r3, di = i1 - i3;
  r0 = ar + cr; i0 = ai + ci;           // Y0
  r2 = ar - cr; i2 = ai - ci;           // Y2
  r1 = br + di; i1 = bi - dr;           // Y1 = (y0-y2) - i (y1-y3)
  r3 = br - di; i3 = bi + dr;           // Y3 = (y0-y2) + i (y1-y3)
}

// forward 8-point DFT, natural order in and out
__device__ __forceinline__ void dft8(float (&r)[8], float (&i)[8]) {
  const float c = 0.70710678118654752440f;
  float ar[4], ai[4], br[4], bi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ar[k] = r[k] + r[k + 4]; ai[k] = i[k] + i[k + 4];
    br[k] = r[k] - r[k + 4]; bi[k] = i[k] - i[k + 4];
  }
  // b_k *= W8^k
  { const float tr = c * (br[1] + bi[1]), ti = c * (bi[1] - br[1]); br[1] = tr; bi[1] = ti; }
  { const float tr = bi[2], ti = -br[2]; br[2] = tr; bi[2] = ti; }
  { const float tr = c * (bi[3] - br[3]), ti = -c * (br[3] + bi[3]); br[3] = tr; bi[3] = ti; }
  dft4(ar[0], ai[0], ar[1], ai[1], ar[2], ai[2], ar[3], ai[3]);
  dft4(br[0], bi[0], br[1], bi[1], br[2], bi[2], br[3], bi[3]);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    r[2 * m] = ar[m]; i[2 * m] = ai[m];
    r[2 * m + 1] = br[m]; i[2 * m + 1] = bi[m];
  }
}

__device__ __forceinline__ float gauss_noise(unsigned long long seed, int b, long long i) {
  const uint32_t h1 = hash_u32(seed, ((unsigned long long)b << 40) ^ (unsigned long long)(2 * i));
  const uint32_t h2 = hash_u32(seed, ((unsigned long long)b << 40) ^ (unsigned long long)(2 * i + 1));
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}

struct LogmelArgs {
  const void* signal;      // [B, Nmax] float32 or int16
  const int32_t* n_samples;
  int sample_is_int16;
  int B;
  long long Nmax;
  int hop, n_mels, n_bins;
  const float* window;     // [512] (symmetric Hann of win_length, centred, zero padded)
  const int32_t* mel_start; const int32_t* mel_len; const float* mel_wt;  // [maxlen][n_mels]
  int mel_maxlen;
  float preemph, dither, fixed_gain, log_floor;
  unsigned long long seed;
  const float* absmax;     // [B]
  float* raw;              // [B, Tmax, n_mels] fp32 log-mel
  int Tmax;
};

__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ signal,
                                                     const int32_t* __restrict__ n_samples,
                                                     int is_i16, long long Nmax,
                                                     float* __restrict__ absmax) {
  const int b = blockIdx.y;
  const long long n = min((long long)n_samples[b], Nmax);
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    const float v = is_i16 ? (float)reinterpret_cast<const int16_t*>(signal)[b * Nmax + i]
                           : reinterpret_cast<const float*>(signal)[b * Nmax + i];
    m = fmaxf(m, fabsf(v));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned int*>(absmax + b), __builtin_bit_cast(unsigned int, m));
}

__global__ __launch_bounds__(256) void logmel_frames_kernel(LogmelArgs p) {
  // per-wave LDS: 2 x (8*72) floats exchange area + 2 x 260 power spectra
  __shared__ float lds[4][2 * 576 + 2 * 260];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const long long N = min((long long)p.n_samples[b], p.Nmax);
  const int Tb = 1 + (int)(N / p.hop);
  const int tA = (blockIdx.x * 4 + wid) * 2, tB = tA + 1;
  float* zr = lds[wid];
  float* zi = lds[wid] + 576;
  float* pa = lds[wid] + 1152;
  float* pb = lds[wid] + 1152 + 260;
  const bool active = tA < Tb;   // wave-uniform
  const float gain = p.fixed_gain > 0.f ? p.fixed_gain : 1.0f / (p.absmax[b] + 1e-5f);

  auto sample = [&](long long i) -> float {   // s(i) = gain*x[i] + dither*noise
    float v = p.sample_is_int16
                  ? (float)reinterpret_cast<const int16_t*>(p.signal)[b * p.Nmax + i]
                  : reinterpret_cast<const float*>(p.signal)[b * p.Nmax + i];
    v *= gain;
    if (p.dither > 0.f) v += p.dither * gauss_noise(p.seed, b, i);
    return v;
  };
  auto pre = [&](long long pidx) -> float {   // pre-emphasised signal at reflect-padded index
    long long i = pidx < 0 ? -pidx : (pidx >= N ? 2 * (N - 1) - pidx : pidx);
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
    const float s0 = sample(i);
    return i > 0 ? s0 - p.preemph * sample(i - 1) : s0;
  };

  float xr[8], xi[8];
  if (active) {
    // ---- stage A: lane = b0, points j = 64a + b0 ----------------------------
    const long long baseA = (long long)tA * p.hop - kNfft / 2;
    const long long baseB = (long long)tB * p.hop - kNfft / 2;
    const bool hasB = tB < Tb;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int j = 64 * a + lane;
      const float w = p.window[j];
      float ra = 0.f, rb = 0.f;
      if (w != 0.f) {
        ra = w * pre(baseA + j);
        if (hasB) rb = w * pre(baseB + j);
      }
      xr[a] = ra; xi[a] = rb;
    }
    dft8(xr, xi);
#pragma unroll
    for (int c = 1; c < 8; ++c) {   // twiddle W512^(lane*c)
      float s, co;
      sincospif(-(float)(lane * c) * (1.0f / 256.0f), &s, &co);
      const float tr = xr[c] * co - xi[c] * s, ti = xr[c] * s + xi[c] * co;
      xr[c] = tr; xi[c] = ti;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { zr[c * 72 + lane] = xr[c]; zi[c * 72 + lane] = xi[c]; }
  }
  __syncthreads();
  const int c = lane >> 3, q = lane & 7;   // stage B: (c, b') ; stage C: (c, c')
  if (active) {
#pragma unroll
    for (int a = 0; a < 8; ++a) { xr[a] = zr[c * 72 + 8 * a + q]; xi[a] = zi[c * 72 + 8 * a + q]; }
    dft8(xr, xi);
#pragma unroll
    for (int cc = 1; cc < 8; ++cc) {   // twiddle W64^(b'*c')
      float s, co;
      sincospif(-(float)(q * cc) * (1.0f / 32.0f), &s, &co);
      const float tr = xr[cc] * co - xi[cc] * s, ti = xr[cc] * s + xi[cc] * co;
      xr[cc] = tr; xi[cc] = ti;
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) { zr[c * 72 + 9 * q + cc] = xr[cc]; zi[c * 72 + 9 * q + cc] = xi[cc]; }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) { xr[bb] = zr[c * 72 + 9 * bb + q]; xi[bb] = zi[c * 72 + 9 * bb + q]; }
    dft8(xr, xi);   // lane (c, c'=q) now holds Z[c + 8q + 64 d'], d' = 0..7
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int d = 0; d < 8; ++d) { zr[c + 8 * q + 64 * d] = xr[d]; zi[c + 8 * q + 64 * d] = xi[d]; }
  }
  __syncthreads();
  if (active) {
    // ---- separate the two real spectra: A = (Zk + conj Zn)/2, B = (Zk - conj Zn)/(2i)
    for (int k = lane; k <= 256; k += 64) {
      const int n = (kNfft - k) & (kNfft - 1);
      const float kr = zr[k], ki = zi[k], nr = zr[n], ni = zi[n];
      const float Ar = 0.5f * (kr + nr), Ai = 0.5f * (ki - ni);
      const float Br = 0.5f * (ki + ni), Bi = -0.5f * (kr - nr);
      pa[k] = Ar * Ar + Ai * Ai;
      pb[k] = Br * Br + Bi * Bi;
    }
  }
  __syncthreads();
  if (active && lane < p.n_mels) {
    const int st = p.mel_start[lane], ln = p.mel_len[lane];
    float sa = 0.f, sb = 0.f;
    for (int j = 0; j < ln; ++j) {
      const float w = p.mel_wt[j * p.n_mels + lane];
      sa += w * pa[st + j];
      sb += w * pb[st + j];
    }
    float* out = p.raw + ((long long)b * p.Tmax + tA) * p.n_mels + lane;
    out[0] = logf(sa + p.log_floor);
    if (tB < Tb) out[p.n_mels] = logf(sb + p.log_floor);
  }
}

// per-utterance whitening + bf16/fp32 store with zero padding up to Tpad. One workgroup per
// utterance walks its frames three times (mean, variance, store) with fp64 running sums: the
// walk is a serial chain per thread, so the workgroup is as wide as it can be (16 frame lanes x
// 64 features; with 4 lanes the kernel was half of the front end's time at B = 32).
constexpr int kNormLanes = 16;
__global__ __launch_bounds__(64 * kNormLanes) void logmel_normalize_kernel(
    const float* __restrict__ raw, const int32_t* __restrict__ n_samples, long long Nmax, int hop,
    int Tmax, int Tpad, int F, int norm_per_feature, bf16_t* __restrict__ out_bf16,
    float* __restrict__ out_f32, int32_t* __restrict__ out_len) {
  __shared__ double red[64 * kNormLanes];
  __shared__ double s_mean[64], s_rstd[64];
  const int b = blockIdx.x;
  const long long N = min((long long)n_samples[b], Nmax);
  const int Tb = min(1 + (int)(N / hop), Tmax);
  const int m = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const float* x = raw + (long long)b * Tmax * F;
  auto lane_sum = [&]() {          // red[m] <- sum over the frame lanes, fixed order
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kNormLanes; ++k) t += red[k * 64 + m];
    return t;
  };
  // pass 1: mean
  double s = 0.0;
  if (m < F)
    for (int t = tl; t < Tb; t += kNormLanes) s += (double)x[(long long)t * F + m];
  red[threadIdx.x] = s;
  __syncthreads();
  double tot1 = 0.0;
  if (tl == 0) tot1 = lane_sum();
  __syncthreads();
  if (tl == 0) red[m] = tot1;
  __syncthreads();
  if (!norm_per_feature) {
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int k = 0; k < F; ++k) tot += red[k];
      for (int k = 0; k < 64; ++k) s_mean[k] = tot / ((double)Tb * F);
    }
  } else if (tl == 0) {
    s_mean[m] = red[m] / (double)Tb;
  }
  __syncthreads();
  // pass 2: population variance (np.std, ddof = 0)
  const double mu = s_mean[m];
  double q = 0.0;
  if (m < F)
    for (int t = tl; t < Tb; t += kNormLanes) {
      const double d = (double)x[(long long)t * F + m] - mu;
      q += d * d;
    }
  __syncthreads();
  red[threadIdx.x] = q;
  __syncthreads();
  double tot2 = 0.0;
  if (tl == 0) tot2 = lane_sum();
  __syncthreads();
  if (tl == 0) red[m] = tot2;
  __syncthreads();
  if (!norm_per_feature) {
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int k = 0; k < F; ++k) tot += red[k];
      for (int k = 0; k < 64; ++k) s_rstd[k] = 1.0 / sqrt(tot / ((double)Tb * F));
    }
  } else if (tl == 0) {
    s_rstd[m] = 1.0 / sqrt(red[m] / (double)Tb);
  }
  __syncthreads();
  const float fm = (float)mu, fr = (float)s_rstd[m];
  if (m < F)
    for (int t = tl; t < Tpad; t += kNormLanes) {
      const float v = t < Tb ? (x[(long long)t * F + m] - fm) * fr : 0.f;
      if (out_bf16) out_bf16[((long long)b * Tpad + t) * F + m] = f2bf(v);
      if (out_f32) out_f32[((long long)b * Tpad + t) * F + m] = v;
    }
  if (threadIdx.x == 0 && out_len) out_len[b] = Tb;
}

}  // namespace os2s

using namespace os2s;

extern "C" size_t os2s_logmel_workspace_bytes(int B, int Tmax, int n_mels) {
  return (size_t)B * Tmax * n_mels * 4 + (size_t)B * 4 + 256;
}

extern "C" int os2s_logmel(os2s_stream_t stream_, const void* signal, const int32_t* n_samples,
                           int sample_is_int16, int B, long long Nmax, int n_fft, int hop,
                           int n_mels, const float* window, const int32_t* mel_start,
                           const int32_t* mel_len, const float* mel_wt, int mel_maxlen,
                           float preemph, float dither, unsigned long long seed,
                           float fixed_gain, float log_floor, int norm_per_feature, int Tmax,
                           int Tpad, uint16_t* out_bf16, float* out_f32, int32_t* out_len,
                           void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(signal && n_samples && window && mel_start && mel_len && mel_wt && workspace);
  OS2S_REQUIRE(B >= 1 && Nmax >= 1 && hop >= 1 && Tmax >= 1 && Tpad >= Tmax);
  if (n_fft != kNfft || n_mels < 1 || n_mels > 64) return OS2S_ERR_UNSUPPORTED;
  if (workspace_bytes < os2s_logmel_workspace_bytes(B, Tmax, n_mels)) return OS2S_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* raw = (float*)workspace;
  float* absmax = (float*)((char*)workspace + (size_t)B * Tmax * n_mels * 4);
  if (hipMemsetAsync(absmax, 0, (size_t)B * 4, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (fixed_gain <= 0.f) {
    int nb = ceil_div(Nmax, 256 * 16);
    if (nb > 64) nb = 64;
    OS2S_LAUNCH(absmax_kernel, dim3(nb, B), dim3(256), 0, stream, signal, n_samples,
                sample_is_int16, Nmax, absmax);
  }
  LogmelArgs a;
  a.signal = signal; a.n_samples = n_samples; a.sample_is_int16 = sample_is_int16; a.B = B;
  a.Nmax = Nmax; a.hop = hop; a.n_mels = n_mels; a.n_bins = n_fft / 2 + 1; a.window = window;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_wt = mel_wt; a.mel_maxlen = mel_maxlen;
  a.preemph = preemph; a.dither = dither; a.fixed_gain = fixed_gain; a.log_floor = log_floor;
  a.seed = seed; a.absmax = absmax; a.raw = raw; a.Tmax = Tmax;
  OS2S_LAUNCH(logmel_frames_kernel, dim3(ceil_div(Tmax, 8), B), dim3(256), 0, stream, a);
  OS2S_LAUNCH(logmel_normalize_kernel, dim3(B), dim3(64 * kNormLanes), 0, stream, raw, n_samples, Nmax, hop,
              Tmax, Tpad, n_mels, norm_per_feature, out_bf16, out_f32, out_len);
  return OS2S_OK;
}
