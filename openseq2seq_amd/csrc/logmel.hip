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
#include <mutex>

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
  float* absmax;           // [B]
  float* raw;              // [B, Tmax, n_mels] fp32 log-mel
  double* partial;         // [B, nblk, 2, 64]: per frame block, per mel bin: sum x, sum x^2
  int Tmax, nblk;
};

// Frames per workgroup of the frames / normalise / absmax kernels: 4 waves x 4 frame pairs.
constexpr int kFB = 32;

// Round 6: ONE pass over the PCM and no re-read of the fp32 feature plane from HBM. The three data passes — max |x|,
// frames, normalise — run over the SAME (utterance, block of 32 frames) units in the SAME order: unit u of the grid
// goes to XCD u % 8 (observed dispatch order; a wrong guess costs speed only), and slot(u) hands every XCD a
// contiguous range of units. What the max pass read and what the frames pass wrote is then in THAT XCD's L2 (the
// batch's PCM is 1.3 - 2.1 MB per XCD, the feature plane 1 MB) when the next pass asks for it: the second read of the
// PCM and the read of the feature plane by the normalisation no longer reach the fabric. The per-feature statistics
// come from per-block partial sums the frames pass emits (fp64, summed in block order by a one-workgroup-per-
// utterance finalize kernel), not from two more walks over the plane.
__device__ __forceinline__ bool logmel_unit(int u, int nblk, int B, int& b, int& fb) {
  const int total = nblk * B;
  const int per = (total + 7) >> 3;
  const int slot = (u & 7) * per + (u >> 3);
  if ((u >> 3) >= per || slot >= total) return false;
  b = slot / nblk;
  fb = slot - b * nblk;
  return true;
}

__global__ __launch_bounds__(256) void absmax_kernel(LogmelArgs p) {
  int b, fb;
  if (!logmel_unit(blockIdx.x, p.nblk, p.B, b, fb)) return;
  const long long n = min((long long)p.n_samples[b], p.Nmax);
  // the samples of frames [fb * 32, fb * 32 + 32): each sample of the utterance belongs to exactly one unit (the last
  // unit takes the tail)
  const long long lo = (long long)fb * kFB * p.hop;
  const long long hi = fb == p.nblk - 1 ? n : min(n, lo + (long long)kFB * p.hop);
  float m = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = p.sample_is_int16 ? (float)reinterpret_cast<const int16_t*>(p.signal)[b * p.Nmax + i]
                                      : reinterpret_cast<const float*>(p.signal)[b * p.Nmax + i];
    m = fmaxf(m, fabsf(v));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(reinterpret_cast<unsigned int*>(p.absmax + b), __builtin_bit_cast(unsigned int, m));
}

// Frames: a workgroup owns 32 consecutive frames of one utterance. The gained + dithered signal of its sample range
// is generated ONCE into LDS (round 5 regenerated every sample — two counter hashes, a log, a square root and a cosine
// for the dither — in each of the ~4 taps that touch it: 1 200 of the ~2 000 instructions a lane spent per frame
// pair), the FFT twiddles are computed once per wave, and a wave walks 4 frame pairs. Same arithmetic per element as
// round 5: the features are bit-identical.
__global__ __launch_bounds__(256) void logmel_frames_kernel(LogmelArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  int b, fb;
  if (!logmel_unit(blockIdx.x, p.nblk, p.B, b, fb)) return;     // (workgroup-uniform: no barrier is skipped)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long N = min((long long)p.n_samples[b], p.Nmax);
  const int Tb = 1 + (int)(N / p.hop);
  const int f0 = fb * kFB;
  float* const fft = lds_dyn + wid * (2 * 576 + 2 * 260);        // per-wave: 2 x (8*72) exchange + 2 x 260 power spectra
  float* const sig = lds_dyn + 4 * (2 * 576 + 2 * 260);          // the block's signal: original indices [s0, s0 + L)
  float* zr = fft;
  float* zi = fft + 576;
  float* pa = fft + 1152;
  float* pb = fft + 1152 + 260;
  const int L = kFB * p.hop + 768;
  long long s0 = (long long)f0 * p.hop - 512;
  s0 = s0 < 0 ? 0 : s0;
  const float gain = p.fixed_gain > 0.f ? p.fixed_gain : 1.0f / (p.absmax[b] + 1e-5f);
  if (f0 < Tb) {
    for (int k = threadIdx.x; k < L; k += 256) {
      const long long i = s0 + k;
      float v = 0.f;
      if (i < N) {
        v = p.sample_is_int16 ? (float)reinterpret_cast<const int16_t*>(p.signal)[b * p.Nmax + i]
                              : reinterpret_cast<const float*>(p.signal)[b * p.Nmax + i];
        v *= gain;
        if (p.dither > 0.f) v += p.dither * gauss_noise(p.seed, b, i);
      }
      sig[k] = v;
    }
  }
  __syncthreads();
  auto pre = [&](long long pidx) -> float {   // pre-emphasised signal at reflect-padded index
    long long i = pidx < 0 ? -pidx : (pidx >= N ? 2 * (N - 1) - pidx : pidx);
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
    int k = (int)(i - s0);
    k = k < 0 ? 0 : (k >= L ? L - 1 : k);     // (never taken: the staged range covers every reflected index)
    const float sv = sig[k];
    return (i > 0 && k > 0) ? sv - p.preemph * sig[k - 1] : sv;
  };
  // twiddles of stages A (W512^(lane c)) and B (W64^(q c)): once per wave
  const int c = lane >> 3, q = lane & 7;
  float twa_s[8], twa_c[8], twb_s[8], twb_c[8];
#pragma unroll
  for (int cc = 1; cc < 8; ++cc) {
    sincospif(-(float)(lane * cc) * (1.0f / 256.0f), &twa_s[cc], &twa_c[cc]);
    sincospif(-(float)(q * cc) * (1.0f / 32.0f), &twb_s[cc], &twb_c[cc]);
  }
  double sum1 = 0.0, sum2 = 0.0;                // this lane's mel bin over the wave's frames
  for (int pr = 0; pr < kFB / 8; ++pr) {
    const int tA = f0 + (pr * 4 + wid) * 2, tB = tA + 1;
    const bool active = tA < Tb;              // wave-uniform
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
      for (int cc = 1; cc < 8; ++cc) {   // twiddle W512^(lane*c)
        const float s_ = twa_s[cc], co = twa_c[cc];
        const float tr = xr[cc] * co - xi[cc] * s_, ti = xr[cc] * s_ + xi[cc] * co;
        xr[cc] = tr; xi[cc] = ti;
      }
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) { zr[cc * 72 + lane] = xr[cc]; zi[cc * 72 + lane] = xi[cc]; }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int a = 0; a < 8; ++a) { xr[a] = zr[c * 72 + 8 * a + q]; xi[a] = zi[c * 72 + 8 * a + q]; }
      dft8(xr, xi);
#pragma unroll
      for (int cc = 1; cc < 8; ++cc) {   // twiddle W64^(b'*c')
        const float s_ = twb_s[cc], co = twb_c[cc];
        const float tr = xr[cc] * co - xi[cc] * s_, ti = xr[cc] * s_ + xi[cc] * co;
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
      const float va = logf(sa + p.log_floor);
      out[0] = va;
      sum1 += (double)va; sum2 += (double)va * (double)va;
      if (tB < Tb) {
        const float vb = logf(sb + p.log_floor);
        out[p.n_mels] = vb;
        sum1 += (double)vb; sum2 += (double)vb * (double)vb;
      }
    }
    __syncthreads();                          // the power spectra / exchange area are reused by the next pair
  }
  // ---- the block's partial sums: the four waves' in wave order (fixed: deterministic) --------------------------
  double* red = reinterpret_cast<double*>(sig);   // (the staged signal is dead: every wave passed the last barrier)
  red[(wid * 2 + 0) * 64 + lane] = sum1;
  red[(wid * 2 + 1) * 64 + lane] = sum2;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * 64 + lane];
    p.partial[(((long long)b * p.nblk + fb) * 2 + which) * 64 + lane] = t;
  }
}

// per-utterance, per-feature mean and 1 / std (np.std, ddof = 0) from the block partials, fp64, block order
__global__ __launch_bounds__(64) void logmel_stats_kernel(const double* __restrict__ partial,
                                                          const int32_t* __restrict__ n_samples, long long Nmax, int hop,
                                                          int Tmax, int nblk, int F, int norm_per_feature,
                                                          float* __restrict__ stats) {
  const int b = blockIdx.x, m = threadIdx.x;
  const long long N = min((long long)n_samples[b], Nmax);
  const int Tb = min(1 + (int)(N / hop), Tmax);
  double s1 = 0.0, s2 = 0.0;
  const int live_blocks = (Tb + kFB - 1) / kFB;
  if (m < F)
    for (int k = 0; k < live_blocks; ++k) {
      s1 += partial[(((long long)b * nblk + k) * 2 + 0) * 64 + m];
      s2 += partial[(((long long)b * nblk + k) * 2 + 1) * 64 + m];
    }
  __shared__ double r1[64], r2[64];
  r1[m] = s1; r2[m] = s2;
  __syncthreads();
  double mu, var;
  if (norm_per_feature) {
    mu = s1 / (double)Tb;
    var = s2 / (double)Tb - mu * mu;
  } else {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < F; ++k) { t1 += r1[k]; t2 += r2[k]; }
    mu = t1 / ((double)Tb * F);
    var = t2 / ((double)Tb * F) - mu * mu;
  }
  var = var > 0.0 ? var : 0.0;
  stats[((long long)b * 2 + 0) * 64 + m] = (float)mu;
  stats[((long long)b * 2 + 1) * 64 + m] = (float)(1.0 / sqrt(var));
}

// whitening + bf16 / fp32 store with zero padding up to Tpad: the same (utterance, 32-frame block) units in the same
// order as the frames pass — the fp32 plane is read from the L2 it was written to
__global__ __launch_bounds__(256) void logmel_normalize_kernel(
    const float* __restrict__ raw, const float* __restrict__ stats, const int32_t* __restrict__ n_samples,
    long long Nmax, int hop, int Tmax, int Tpad, int nblk, int B, int F, bf16_t* __restrict__ out_bf16,
    float* __restrict__ out_f32, int32_t* __restrict__ out_len) {
  int b, fb;
  if (!logmel_unit(blockIdx.x, nblk, B, b, fb)) return;
  const long long N = min((long long)n_samples[b], Nmax);
  const int Tb = min(1 + (int)(N / hop), Tmax);
  const int m = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const float fm = stats[((long long)b * 2 + 0) * 64 + m], fr = stats[((long long)b * 2 + 1) * 64 + m];
  const float* x = raw + (long long)b * Tmax * F;
  // the last block of an utterance also writes the zero frames up to Tpad
  const int t_end = fb == nblk - 1 ? Tpad : min(Tpad, (fb + 1) * kFB);
  if (m < F)
    for (int t = fb * kFB + tl; t < t_end; t += 4) {
      const float v = t < Tb ? (x[(long long)t * F + m] - fm) * fr : 0.f;
      if (out_bf16) out_bf16[((long long)b * Tpad + t) * F + m] = f2bf(v);
      if (out_f32) out_f32[((long long)b * Tpad + t) * F + m] = v;
    }
  if (threadIdx.x == 0 && fb == 0 && out_len) out_len[b] = Tb;
}

}  // namespace os2s

using namespace os2s;

static int logmel_nblk(int Tmax, int Tpad) { return os2s::ceil_div(Tpad > Tmax ? Tpad : Tmax, os2s::kFB); }

extern "C" size_t os2s_logmel_workspace_bytes(int B, int Tmax, int n_mels) {
  // fp32 plane | max |x| per utterance | block partials (fp64; Tpad <= Tmax rounded up to pad_to: one more block at most)
  // | mean, 1 / std per utterance and feature
  const size_t nblk = (size_t)os2s::ceil_div(Tmax, os2s::kFB) + 2;
  return (size_t)B * Tmax * n_mels * 4 + (size_t)B * 4 + 256 + (size_t)B * nblk * 2 * 64 * 8 + (size_t)B * 2 * 64 * 4 + 256;
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
  if (n_fft != kNfft || n_mels < 1 || n_mels > 64 || hop > 512) return OS2S_ERR_UNSUPPORTED;
  if (workspace_bytes < os2s_logmel_workspace_bytes(B, Tmax, n_mels)) return OS2S_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int nblk = logmel_nblk(Tmax, Tpad);
  if ((size_t)nblk > (size_t)ceil_div(Tmax, kFB) + 2) return OS2S_ERR_WORKSPACE;     // (Tpad far beyond Tmax)
  char* w = (char*)workspace;
  float* raw = (float*)w;
  w += (size_t)B * Tmax * n_mels * 4;
  float* absmax = (float*)w;
  w += ((size_t)B * 4 + 255) / 256 * 256;
  double* partial = (double*)w;
  w += (size_t)B * nblk * 2 * 64 * 8;
  float* stats = (float*)w;
  LogmelArgs a;
  a.signal = signal; a.n_samples = n_samples; a.sample_is_int16 = sample_is_int16; a.B = B;
  a.Nmax = Nmax; a.hop = hop; a.n_mels = n_mels; a.n_bins = n_fft / 2 + 1; a.window = window;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_wt = mel_wt; a.mel_maxlen = mel_maxlen;
  a.preemph = preemph; a.dither = dither; a.fixed_gain = fixed_gain; a.log_floor = log_floor;
  a.seed = seed; a.absmax = absmax; a.raw = raw; a.partial = partial; a.Tmax = Tmax; a.nblk = nblk;
  const int units = ceil_div(nblk * B, 8) * 8;
  if (fixed_gain <= 0.f) {
    if (hipMemsetAsync(absmax, 0, (size_t)B * 4, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
    OS2S_LAUNCH(absmax_kernel, dim3(units), dim3(256), 0, stream, a);
  }
  const size_t smem = (size_t)(4 * (2 * 576 + 2 * 260) + kFB * hop + 768) * 4;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)logmel_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess || smem > 160 * 1024) return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH(logmel_frames_kernel, dim3(units), dim3(256), smem, stream, a);
  OS2S_LAUNCH(logmel_stats_kernel, dim3(B), dim3(64), 0, stream, partial, n_samples, Nmax, hop, Tmax, nblk, n_mels,
              norm_per_feature, stats);
  OS2S_LAUNCH(logmel_normalize_kernel, dim3(units), dim3(256), 0, stream, raw, stats, n_samples, Nmax, hop,
              Tmax, Tpad, nblk, B, n_mels, out_bf16, out_f32, out_len);
  return OS2S_OK;
}
