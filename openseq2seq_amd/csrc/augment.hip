// Audio augmentation of the speech-to-text data layer on the device, gfx950.
//
// Reference (host NumPy inside tf.py_func, open_seq2seq/data/speech2text/speech_utils.py):
//   normalize_signal :225-231        x * gain, gain = 1/(max|x| + 1e-5) unless fixed
//   augment_audio_signal :234-272    speed perturbation = resampy.resample(x, sr, int(sr*a),
//                                    filter='kaiser_best'), then additive Gaussian noise of a
//                                    random level (dB)
//   SpecAugment masks :419-433       n_freq_mask bands [:, f0:f0+fw] and n_time_mask bands
//                                    [t0:t0+tw, :] of the normalised features set to 0
// resampy is a third-party dependency that is not vendored in the reference; its published
// algorithm (band-limited sinc interpolation, Smith; resampy/interpn.py resample_f: a
// half-window table of num_zeros * 2^precision + 1 taps walked with stride int(scale * 2^prec),
// linear interpolation between neighbouring taps) is restated here with the table passed in by
// the caller. The random draws (stretch factor, noise level, mask positions) stay on the host
// RNG of the data layer, in the reference's order; the kernels are deterministic given them.
// All three kernels are HBM-bound streaming passes over the batch.
#include "os2s_common.hpp"

namespace os2s {

__global__ __launch_bounds__(256) void absmax_rows_kernel(const void* __restrict__ signal, int is_i16,
                                                          long long nmax,
                                                          const int32_t* __restrict__ n_samples,
                                                          uint32_t* __restrict__ absmax_bits) {
  const int b = blockIdx.y;
  const long long n = min((long long)n_samples[b], nmax);
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = is_i16 ? (float)reinterpret_cast<const int16_t*>(signal)[b * nmax + i]
                           : reinterpret_cast<const float*>(signal)[b * nmax + i];
    m = fmaxf(m, fabsf(v));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(absmax_bits + b, __float_as_uint(m));   // m >= 0: bit order = value order
}

struct ResampleArgs {
  const void* signal; int is_i16; long long nmax;
  const int32_t* n_in;          // [B]
  const int32_t* n_out;         // [B]  = int(n_in * ratio)
  const double* ratio;          // [B]  sample_ratio = sr_new / sr_orig (1.0 = copy)
  const float* noise_amp;       // [B]  10^(dB/20) or 0
  const uint32_t* absmax_bits;  // [B]
  float fixed_gain;             // > 0: use it; else 1/(absmax + 1e-5)
  const float* win; int nwin, num_table;
  unsigned long long seed;
  float* out; long long nout_max;
};

__device__ __forceinline__ float load_sample(const ResampleArgs& p, long long base, int i) {
  return p.is_i16 ? (float)reinterpret_cast<const int16_t*>(p.signal)[base + i]
                  : reinterpret_cast<const float*>(p.signal)[base + i];
}

__global__ __launch_bounds__(256) void resample_noise_kernel(ResampleArgs p) {
  const int b = blockIdx.y;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.nout_max) return;
  float* dst = p.out + (long long)b * p.nout_max + t;
  const int n_out = p.n_out[b];
  if (t >= n_out) { *dst = 0.f; return; }
  const int n_orig = p.n_in[b];
  const long long base = (long long)b * p.nmax;
  const float gain = p.fixed_gain > 0.f ? p.fixed_gain : 1.f / (__uint_as_float(p.absmax_bits[b]) + 1e-5f);
  const double sample_ratio = p.ratio[b];
  float y;
  if (sample_ratio == 1.0) {
    y = load_sample(p, base, (int)t) * gain;
  } else {
    // resampy.interpn.resample_f, one output sample
    const double scale = sample_ratio < 1.0 ? sample_ratio : 1.0;
    const int index_step = (int)(scale * p.num_table);
    const double time_register = (double)t / sample_ratio;
    const int n = (int)time_register;
    const float wscale = sample_ratio < 1.0 ? (float)sample_ratio : 1.f;   // interp_win *= sample_ratio
    double frac = scale * (time_register - n);
    double index_frac = frac * p.num_table;
    int offset = (int)index_frac;
    float eta = (float)(index_frac - offset);
    float acc = 0.f;
    int i_max = min(n + 1, (p.nwin - offset) / index_step);
    for (int i = 0; i < i_max; ++i) {
      const int idx = offset + i * index_step;
      const float w0 = p.win[idx], w1 = idx + 1 < p.nwin ? p.win[idx + 1] : w0;   // delta of the last tap = 0
      acc += (w0 + eta * (idx + 1 < p.nwin ? w1 - w0 : 0.f)) * load_sample(p, base, n - i);
    }
    frac = scale - frac;
    index_frac = frac * p.num_table;
    offset = (int)index_frac;
    eta = (float)(index_frac - offset);
    int k_max = min(n_orig - n - 1, (p.nwin - offset) / index_step);
    for (int k = 0; k < k_max; ++k) {
      const int idx = offset + k * index_step;
      const float w0 = p.win[idx], w1 = idx + 1 < p.nwin ? p.win[idx + 1] : w0;
      acc += (w0 + eta * (idx + 1 < p.nwin ? w1 - w0 : 0.f)) * load_sample(p, base, n + k + 1);
    }
    y = acc * wscale * gain;
  }
  const float amp = p.noise_amp ? p.noise_amp[b] : 0.f;
  if (amp > 0.f) {     // Box-Muller on two counter-based uniforms
    const unsigned long long idx = ((unsigned long long)b << 40) + (unsigned long long)t;
    const float u1 = ((float)hash_u32(p.seed, 2 * idx) + 1.f) * (1.f / 4294967296.f);
    const float u2 = (float)hash_u32(p.seed, 2 * idx + 1) * (1.f / 4294967296.f);
    y += amp * sqrtf(-2.f * logf(u1)) * cosf(6.2831853071795864f * u2);
  }
  *dst = y;
}

// masks: int32 [B, n_masks, 4] = (t0, t1, f0, f1) half-open boxes; elements inside any box -> 0
__global__ __launch_bounds__(256) void spec_augment_kernel(bf16_t* __restrict__ feats, int T, int F,
                                                           const int32_t* __restrict__ masks,
                                                           int n_masks) {
  const int b = blockIdx.y;
  const int32_t* mk = masks + (long long)b * n_masks * 4;
  for (int m = 0; m < n_masks; ++m) {
    const int t0 = max(mk[4 * m], 0), t1 = min(mk[4 * m + 1], T);
    const int f0 = max(mk[4 * m + 2], 0), f1 = min(mk[4 * m + 3], F);
    const int w = f1 - f0;
    const long long cnt = (long long)max(t1 - t0, 0) * max(w, 0);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (long long)gridDim.x * 256) {
      const int t = t0 + (int)(e / w), f = f0 + (int)(e % w);
      feats[((long long)b * T + t) * F + f] = 0;
    }
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_augment_signal(os2s_stream_t stream, const void* signal, int is_int16, int B,
                                   long long nmax, const int32_t* n_in, const int32_t* n_out,
                                   const double* ratio, const float* noise_amp, float fixed_gain,
                                   const float* interp_win, int nwin, int num_table,
                                   unsigned long long seed, uint32_t* absmax_scratch, float* out,
                                   long long nout_max) {
  OS2S_REQUIRE(signal && n_in && n_out && ratio && interp_win && absmax_scratch && out);
  OS2S_REQUIRE(B >= 1 && nmax >= 1 && nout_max >= 1 && nwin >= 2 && num_table >= 1);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(absmax_scratch, 0, (size_t)B * 4, s) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (fixed_gain <= 0.f)
    OS2S_LAUNCH(absmax_rows_kernel, dim3((unsigned)min((long long)64, (nmax + 255) / 256), B), dim3(256), 0,
                s, signal, is_int16, nmax, n_in, absmax_scratch);
  ResampleArgs a;
  a.signal = signal; a.is_i16 = is_int16; a.nmax = nmax; a.n_in = n_in; a.n_out = n_out;
  a.ratio = ratio; a.noise_amp = noise_amp; a.absmax_bits = absmax_scratch; a.fixed_gain = fixed_gain;
  a.win = interp_win; a.nwin = nwin; a.num_table = num_table; a.seed = seed; a.out = out;
  a.nout_max = nout_max;
  OS2S_LAUNCH(resample_noise_kernel, dim3((unsigned)((nout_max + 255) / 256), B), dim3(256), 0, s, a);
  return OS2S_OK;
}

extern "C" int os2s_spec_augment(os2s_stream_t stream, uint16_t* feats, int B, int T, int F,
                                 const int32_t* masks, int n_masks) {
  OS2S_REQUIRE(feats && B >= 1 && T >= 1 && F >= 1 && n_masks >= 0);
  if (n_masks == 0) return OS2S_OK;
  OS2S_REQUIRE(masks);
  OS2S_LAUNCH(spec_augment_kernel, dim3(32, B), dim3(256), 0, (hipStream_t)stream, feats, T, F, masks,
              n_masks);
  return OS2S_OK;
}
