// Text-to-speech spectrogram features, gfx950 — the per-utterance arithmetic of
// get_speech_features (open_seq2seq/data/text2speech/speech_utils.py:98-182):
//   librosa.stft(y, n_fft) (hop n_fft/4, periodic Hann of n_fft, centre + reflect padding)
//   -> |X|^mag_power -> log(clip(mag, data_min_mag))[:, :n_mag]   ("magnitude" features)
//   -> log(clip(mel_basis . mag, data_min_mel))                   ("mel" features; htk basis)
// n_fft is 800 (M-AILABS 16 kHz) or 1024 (LJSpeech): not a power of two in the benchmark
// config, so the transform is a direct real DFT in fp32 — one workgroup per frame, the
// windowed frame and an n_fft-entry twiddle table in LDS, one thread per frequency bin with
// an incremental (k*n mod N) phase index. It is a data-layer op (once per batch,
// ~0.5 M complex MACs per frame), bound by VALU, not by HBM: 2 B read + ~2 KB written/frame.
#include "os2s_common.hpp"

namespace os2s {

__global__ __launch_bounds__(256) void tts_spectrogram_kernel(
    const float* __restrict__ signal, long long sig_stride, const int32_t* __restrict__ n_samples,
    const float* __restrict__ window, int n_fft, int hop, int T, int mag_power, float data_min_mag,
    float data_min_mel, int n_mag, int n_mels, const int32_t* __restrict__ mel_start,
    const int32_t* __restrict__ mel_len, const float* __restrict__ mel_wt, int mel_maxlen,
    float* __restrict__ out_mel, float* __restrict__ out_mag, float pad_mel, float pad_mag) {
  extern __shared__ float sm[];
  float* x = sm;                    // [n_fft] windowed frame
  float* cs = sm + n_fft;           // [n_fft] cos
  float* sn = sm + 2 * n_fft;       // [n_fft] sin
  float* mag = sm + 3 * n_fft;      // [n_fft/2+1]
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int n = n_samples[b];
  const int frames = 1 + n / hop;
  const int nbins = n_fft / 2 + 1;
  float* om = out_mel ? out_mel + ((long long)b * T + t) * n_mels : nullptr;
  float* og = out_mag ? out_mag + ((long long)b * T + t) * n_mag : nullptr;
  if (t >= frames) {      // batch padding
    if (om) for (int m = tid; m < n_mels; m += 256) om[m] = pad_mel;
    if (og) for (int k = tid; k < n_mag; k += 256) og[k] = pad_mag;
    return;
  }
  const float* sig = signal + (long long)b * sig_stride;
  const int start = t * hop - n_fft / 2;
  for (int i = tid; i < n_fft; i += 256) {
    int j = start + i;          // np.pad(mode='reflect'): ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    j = min(max(j, 0), n - 1);
    x[i] = sig[j] * window[i];
    float s, c;
    sincospif(2.0f * (float)i / (float)n_fft, &s, &c);
    cs[i] = c;
    sn[i] = s;
  }
  __syncthreads();
  for (int k = tid; k < nbins; k += 256) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int i = 0; i < n_fft; ++i) {
      re += x[i] * cs[idx];
      im -= x[i] * sn[idx];
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    const float p2 = re * re + im * im;
    mag[k] = mag_power == 2 ? p2 : sqrtf(p2);
  }
  __syncthreads();
  if (og)
    for (int k = tid; k < n_mag; k += 256) og[k] = logf(fmaxf(mag[k], data_min_mag));
  if (om)
    for (int m = tid; m < n_mels; m += 256) {
      float a = 0.f;
      const int s0 = mel_start[m], ln = mel_len[m];
      for (int i = 0; i < ln; ++i) a += mel_wt[(long long)i * n_mels + m] * mag[s0 + i];
      om[m] = logf(fmaxf(a, data_min_mel));
    }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_tts_spectrogram(os2s_stream_t stream, const float* signal, long long sig_stride,
                                    const int32_t* n_samples, const float* window, int B, int n_fft,
                                    int hop, int T, int mag_power, float data_min_mag,
                                    float data_min_mel, int n_mag, int n_mels,
                                    const int32_t* mel_start, const int32_t* mel_len,
                                    const float* mel_wt, int mel_maxlen, float* out_mel,
                                    float* out_mag, float pad_mel, float pad_mag) {
  OS2S_REQUIRE(signal && n_samples && window && B >= 1 && T >= 1 && n_fft >= 16 && n_fft % 2 == 0);
  OS2S_REQUIRE(hop >= 1 && (mag_power == 1 || mag_power == 2) && (out_mel || out_mag));
  if (out_mag) OS2S_REQUIRE(n_mag >= 1 && n_mag <= n_fft / 2 + 1);
  if (out_mel) OS2S_REQUIRE(n_mels >= 1 && mel_start && mel_len && mel_wt);
  const size_t lds = (size_t)(3 * n_fft + n_fft / 2 + 1) * sizeof(float);
  if (lds > 64 * 1024) return OS2S_ERR_UNSUPPORTED;
  OS2S_LAUNCH(tts_spectrogram_kernel, dim3(T, B), dim3(256), lds, (hipStream_t)stream, signal, sig_stride,
              n_samples, window, n_fft, hop, T, mag_power, data_min_mag, data_min_mel, n_mag, n_mels,
              mel_start, mel_len, mel_wt, mel_maxlen, out_mel, out_mag, pad_mel, pad_mag);
  return OS2S_OK;
}
