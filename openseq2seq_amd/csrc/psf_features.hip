// 'spectrogram' features of the python_speech_features backend (the DeepSpeech2 configs:
// input_type 'spectrogram', 160 bins of a 320-point spectrum), gfx950 — the per-utterance
// arithmetic of get_speech_features_psf (open_seq2seq/data/speech2text/speech_utils.py:444-535):
//   s16 = int16(trunc(x / (max|x| + 1e-5) * 32767))                 (normalize_signal, :216-222)
//   frames of n_win samples every n_step, zero-padded tail, SYMMETRIC Hann (np.hanning)
//   lps = 10 log10(max(|rfft(frame, n_win)|^2 / n_win, 1e-30)) [- max over the utterance]
//   features = (lps[:, :F] - mean) / std over the whole utterance, frames rounded up to pad_to
// The global-max term of logpowspec(norm=1) shifts every value of the utterance by the same
// constant and cancels in the mean / std normalisation: it is not computed. The zero frames added
// to reach a multiple of pad_to (-300 dB rows) DO enter mean and std, as in the reference.
// n_win = 320 is not a power of two: direct real DFT in fp32 with the twiddles in LDS, one
// workgroup per frame, one thread per bin (a data-layer op: ~51 k MAC per frame).
#include "os2s_common.hpp"

namespace os2s {

__global__ __launch_bounds__(256) void psf_absmax_kernel(const void* __restrict__ signal, int is_i16,
                                                        long long sig_stride,
                                                        const int32_t* __restrict__ n_samples,
                                                        float* __restrict__ gain) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int n = (int)min((long long)n_samples[b], sig_stride);   // never past the row (as logmel.hip)
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = is_i16 ? (float)reinterpret_cast<const int16_t*>(signal)[(long long)b * sig_stride + i]
                           : reinterpret_cast<const float*>(signal)[(long long)b * sig_stride + i];
    m = fmaxf(m, fabsf(v));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) gain[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + 1e-5f;
}

// frames_out[b] = frames of utterance b incl. the pad_to rounding; plane[b, t, :] = lps; per-frame
// partial (sum, sum^2) of the F kept bins for the utterance statistics
__global__ __launch_bounds__(256) void psf_logpowspec_kernel(
    const void* __restrict__ signal, int is_i16, long long sig_stride, const int32_t* __restrict__ n_samples,
    const float* __restrict__ denom, int n_win, int n_step, int pad_to, int F, int T,
    float* __restrict__ plane, double* __restrict__ partial, int32_t* __restrict__ frames_out) {
  extern __shared__ float sm[];
  float* x = sm;                 // [n_win] windowed frame
  float* cs = sm + n_win;        // [n_win]
  float* sn = sm + 2 * n_win;    // [n_win]
  __shared__ double red[2][4];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int n = (int)min((long long)n_samples[b], sig_stride);   // never past the row
  int frames = n <= n_win ? 1 : 1 + (n - n_win + n_step - 1) / n_step;
  if (pad_to > 0 && frames % pad_to) frames += pad_to - frames % pad_to;
  if (t == 0 && tid == 0) frames_out[b] = frames;
  if (t >= frames) {
    if (tid == 0) { partial[((long long)b * T + t) * 2] = 0.0; partial[((long long)b * T + t) * 2 + 1] = 0.0; }
    return;
  }
  const float d = denom[b];
  for (int i = tid; i < n_win; i += 256) {
    const long long j = (long long)t * n_step + i;
    float v = 0.f;
    if (j < n) {
      const float raw = is_i16 ? (float)reinterpret_cast<const int16_t*>(signal)[(long long)b * sig_stride + j]
                               : reinterpret_cast<const float*>(signal)[(long long)b * sig_stride + j];
      v = truncf((raw / d) * 32767.0f);                       // astype(np.int16): toward zero
    }
    // np.hanning(M)[i] = 0.5 - 0.5 cos(2 pi i / (M - 1))
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(n_win - 1));
    x[i] = v * w;
    float s, c;
    sincospif(2.0f * (float)i / (float)n_win, &s, &c);
    cs[i] = c;
    sn[i] = s;
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < F; k += 256) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int i = 0; i < n_win; ++i) {
      re += x[i] * cs[idx];
      im -= x[i] * sn[idx];
      idx += k;
      if (idx >= n_win) idx -= n_win;
    }
    const float ps = fmaxf((re * re + im * im) / (float)n_win, 1e-30f);
    const float lps = 10.0f * log10f(ps);
    plane[((long long)b * T + t) * F + k] = lps;
    s1 += (double)lps;
    s2 += (double)lps * (double)lps;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    partial[((long long)b * T + t) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[((long long)b * T + t) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// mean / std over the utterance (fixed summation order), then (x - mean) / std -> bf16 (+ fp32)
__global__ __launch_bounds__(256) void psf_normalize_kernel(const float* __restrict__ plane,
                                                           const double* __restrict__ partial,
                                                           const int32_t* __restrict__ frames_out, int F,
                                                           int T, bf16_t* __restrict__ out16,
                                                           float* __restrict__ out32) {
  __shared__ double sh[2];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int frames = frames_out[b];
  if (tid == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < frames; ++t) { s1 += partial[((long long)b * T + t) * 2]; s2 += partial[((long long)b * T + t) * 2 + 1]; }
    const double cnt = (double)frames * F;
    const double mean = s1 / cnt;
    const double var = fmax(s2 / cnt - mean * mean, 0.0);
    sh[0] = mean;
    sh[1] = 1.0 / sqrt(var);
  }
  __syncthreads();
  const float mean = (float)sh[0], rstd = (float)sh[1];
  const long long per = (long long)T * F;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < per; i += (long long)gridDim.x * 256) {
    const int t = (int)(i / F);
    const float v = t < frames ? (plane[(long long)b * per + i] - mean) * rstd : 0.f;
    out16[(long long)b * per + i] = f2bf(v);
    if (out32) out32[(long long)b * per + i] = v;
  }
}

// 'logfbank' features of the python_speech_features backend (the toy Wave2Letter / residual-TDNN test
// configurations of the reference: backend psf by default, input_type 'logfbank', 40 filters) —
// get_speech_features_psf (speech_utils.py:517-535) -> psf.logfbank(signal, samplerate, winlen, winstep,
// nfilt, nfft = 512, lowfreq = 0, highfreq = sr / 2, preemph = 0.97) of python_speech_features 0.6:
//   s16 as above, zero-padded to the pad_to frame count;  y[j] = s16[j] - 0.97 s16[j-1]  (y[0] = s16[0]; the
//   first padding sample is -0.97 s16[n-1], not zero)
//   frames of n_win samples every n_step, RECTANGULAR window, zero-padded to nfft
//   pspec = |rfft(frame, nfft)|^2 / nfft;  feat = pspec . fb^T  (fb: triangular HTK-mel filters, host table
//   [nfilt][nfft/2 + 1]);  feat == 0 -> 2.220446e-16 (numpy double eps);  features = ln(feat)
//   then (features - mean) / std over the whole utterance, pad frames included (ln(eps) rows)
// One workgroup per frame: direct real DFT of the n_win live samples against a 512-entry twiddle table in
// LDS (nfft = 512 has a fast transform, but this is a data-layer op of ~82 k MAC per frame), power spectrum
// in LDS, one thread per filter.
__global__ __launch_bounds__(256) void psf_logfbank_kernel(
    const void* __restrict__ signal, int is_i16, long long sig_stride, const int32_t* __restrict__ n_samples,
    const float* __restrict__ denom, int n_win, int n_step, int pad_to, int nfilt, int nfft,
    const float* __restrict__ fb, int T, float* __restrict__ plane, double* __restrict__ partial,
    int32_t* __restrict__ frames_out) {
  extern __shared__ float sm[];
  float* x = sm;                 // [n_win] pre-emphasised frame
  float* cs = sm + n_win;        // [nfft]
  float* sn = cs + nfft;         // [nfft]
  float* ps = sn + nfft;         // [nfft / 2 + 1]
  __shared__ double red[2][4];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int nbins = nfft / 2 + 1;
  const int n = (int)min((long long)n_samples[b], sig_stride);
  int frames = n <= n_win ? 1 : 1 + (n - n_win + n_step - 1) / n_step;
  if (pad_to > 0 && frames % pad_to) frames += pad_to - frames % pad_to;
  if (t == 0 && tid == 0) frames_out[b] = frames;
  if (t >= frames) {
    if (tid == 0) { partial[((long long)b * T + t) * 2] = 0.0; partial[((long long)b * T + t) * 2 + 1] = 0.0; }
    return;
  }
  const float d = denom[b];
  auto s16 = [&](long long j) -> float {
    if (j < 0 || j >= n) return 0.f;
    const float raw = is_i16 ? (float)reinterpret_cast<const int16_t*>(signal)[(long long)b * sig_stride + j]
                             : reinterpret_cast<const float*>(signal)[(long long)b * sig_stride + j];
    return truncf((raw / d) * 32767.0f);                      // astype(np.int16): toward zero
  };
  // the reference pads the SIGNAL (by whole strides, to the pad_to frame count) before psf.logfbank runs its
  // pre-emphasis over it; framesig then zero-pads the pre-emphasised signal to complete the last frame. So
  // y[n] = -0.97 s16[n-1] exists only when that signal padding happened (plen > n), and y[j] = 0 past plen.
  long long plen = n;
  {
    const int length = n <= n_win ? 1 : 1 + (n - n_win + n_step - 1) / n_step;   // 1 + ceil((n - n_win) / n_step)
    // (python: 1 + int(ceil((n - n_win) / n_step)) is <= 1 for n <= n_win: ceil of a non-positive quotient;
    // for n < n_win - n_step it is <= 0 — such clips (< 10 ms) do not occur: frames >= 1 is kept)
    if (pad_to > 0 && length % pad_to) plen = (long long)n + (long long)(pad_to - length % pad_to) * n_step;
  }
  for (int i = tid; i < n_win; i += 256) {
    const long long j = (long long)t * n_step + i;
    x[i] = j >= plen ? 0.f : (j == 0 ? s16(0) : s16(j) - 0.97f * s16(j - 1));
  }
  for (int i = tid; i < nfft; i += 256) {
    float s, c;
    sincospif(2.0f * (float)i / (float)nfft, &s, &c);
    cs[i] = c;
    sn[i] = s;
  }
  __syncthreads();
  for (int k = tid; k < nbins; k += 256) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int i = 0; i < n_win; ++i) {
      re += x[i] * cs[idx];
      im -= x[i] * sn[idx];
      idx = (idx + k) & (nfft - 1);
    }
    ps[k] = (re * re + im * im) / (float)nfft;
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  for (int m = tid; m < nfilt; m += 256) {
    float e = 0.f;
    const float* const f = fb + (long long)m * nbins;
    for (int k = 0; k < nbins; ++k) e += ps[k] * f[k];
    const float v = logf(e == 0.f ? 2.220446049250313e-16f : e);
    plane[((long long)b * T + t) * nfilt + m] = v;
    s1 += (double)v;
    s2 += (double)v * (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    partial[((long long)b * T + t) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[((long long)b * T + t) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

}  // namespace os2s

using namespace os2s;

// 'logfbank' of the psf backend (see psf_logfbank_kernel): fb = [nfilt][nfft/2 + 1] fp32 filter table on the
// device (python_speech_features.get_filterbanks; the host layer builds it). nfft: a power of two >= n_win.
// Workspace as os2s_psf_spectrogram_workspace_bytes(B, Tpad, nfilt).
extern "C" int os2s_psf_logfbank(os2s_stream_t stream_, const void* signal, int sample_is_int16,
                                 const int32_t* n_samples, int B, long long Nmax, int n_win, int n_step,
                                 int pad_to, int nfilt, int nfft, const float* fb, int Tpad,
                                 uint16_t* out_bf16, float* out_f32, int32_t* out_len, void* workspace,
                                 size_t workspace_bytes) {
  OS2S_REQUIRE(signal && n_samples && fb && out_bf16 && out_len && workspace);
  OS2S_REQUIRE(B >= 1 && n_win >= 16 && n_step >= 1 && Tpad >= 1 && nfilt >= 1 && nfilt <= 4096);
  OS2S_REQUIRE(nfft >= n_win && (nfft & (nfft - 1)) == 0);
  if (workspace_bytes < os2s_psf_spectrogram_workspace_bytes(B, Tpad, nfilt)) return OS2S_ERR_WORKSPACE;
  const size_t lds = ((size_t)n_win + 2 * (size_t)nfft + nfft / 2 + 1) * sizeof(float);
  if (lds > 48 * 1024) return OS2S_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = reinterpret_cast<char*>(workspace);
  float* denom = reinterpret_cast<float*>(ws);
  size_t off = ((size_t)B * 4 + 63) / 64 * 64;
  double* partial = reinterpret_cast<double*>(ws + off);
  off += (size_t)B * Tpad * 2 * 8;
  float* plane = reinterpret_cast<float*>(ws + off);
  OS2S_LAUNCH(psf_absmax_kernel, dim3(B), dim3(256), 0, stream, signal, sample_is_int16, Nmax, n_samples, denom);
  OS2S_LAUNCH(psf_logfbank_kernel, dim3(Tpad, B), dim3(256), lds, stream, signal, sample_is_int16, Nmax, n_samples,
              denom, n_win, n_step, pad_to, nfilt, nfft, fb, Tpad, plane, partial, out_len);
  OS2S_LAUNCH(psf_normalize_kernel, dim3(64, B), dim3(256), 0, stream, plane, partial, out_len, nfilt, Tpad,
              out_bf16, out_f32);
  return OS2S_OK;
}

extern "C" size_t os2s_psf_spectrogram_workspace_bytes(int B, int T, int F) {
  return (size_t)B * 4 + (size_t)B * T * F * 4 + (size_t)B * T * 2 * 8 + 64;
}

extern "C" int os2s_psf_spectrogram(os2s_stream_t stream_, const void* signal, int sample_is_int16,
                                    const int32_t* n_samples, int B, long long Nmax, int n_win,
                                    int n_step, int pad_to, int num_features, int Tpad,
                                    uint16_t* out_bf16, float* out_f32, int32_t* out_len,
                                    void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(signal && n_samples && out_bf16 && out_len && workspace);
  OS2S_REQUIRE(B >= 1 && n_win >= 16 && n_step >= 1 && Tpad >= 1 && num_features >= 1);
  OS2S_REQUIRE(num_features <= n_win / 2 + 1);   // the reference's assertion (speech_utils.py:501-502)
  if (workspace_bytes < os2s_psf_spectrogram_workspace_bytes(B, Tpad, num_features)) return OS2S_ERR_WORKSPACE;
  if ((size_t)3 * n_win * sizeof(float) > 48 * 1024) return OS2S_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = reinterpret_cast<char*>(workspace);
  float* denom = reinterpret_cast<float*>(ws);
  size_t off = ((size_t)B * 4 + 63) / 64 * 64;
  double* partial = reinterpret_cast<double*>(ws + off);
  off += (size_t)B * Tpad * 2 * 8;
  float* plane = reinterpret_cast<float*>(ws + off);
  OS2S_LAUNCH(psf_absmax_kernel, dim3(B), dim3(256), 0, stream, signal, sample_is_int16, Nmax, n_samples, denom);
  OS2S_LAUNCH(psf_logpowspec_kernel, dim3(Tpad, B), dim3(256), (size_t)3 * n_win * sizeof(float), stream, signal,
              sample_is_int16, Nmax, n_samples, denom, n_win, n_step, pad_to, num_features, Tpad, plane, partial,
              out_len);
  OS2S_LAUNCH(psf_normalize_kernel, dim3(64, B), dim3(256), 0, stream, plane, partial, out_len, num_features,
              Tpad, out_bf16, out_f32);
  return OS2S_OK;
}
