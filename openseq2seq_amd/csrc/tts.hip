// Text-to-speech loss and small element-wise pieces of the Tacotron2 path, gfx950.
//
// Text2SpeechLoss (open_seq2seq/losses/text2speech_loss.py:35-209): masked
// tf.losses.mean_squared_error / absolute_difference (reduction SUM_BY_NONZERO_WEIGHTS:
// sum(w * err) / #non-zero broadcast weights = F * sum_b len_b) on the decoder mel, the
// post-net mel and the magnitude prediction, plus the masked sigmoid cross entropy of the
// stop token divided by sum(mask). All are the same HBM-bound pass: read prediction and
// target once, write the gradient once, block partial sums -> one finalize.
#include "os2s_common.hpp"

namespace os2s {

constexpr int kTtsRowsPerBlock = 16;

// mode 0: (p-t)^2, 1: |p-t|, 2: sigmoid cross entropy with logits p and labels t.
// Prediction and target may have different row counts per sample (Tp / Tt): both are padded to
// T = max(Tp, Tt) as text2speech_loss.py:80-117 does — predictions with zeros, targets with target_pad
// (0 for spectrogram rows, 1 for the stop token) — and the mask is sequence_mask(lens, T).
__global__ __launch_bounds__(256) void tts_loss_kernel(
    const bf16_t* __restrict__ pred, long long ld_p, int Tp, const float* __restrict__ target,
    long long ld_t, int Tt, float target_pad, const int32_t* __restrict__ lens, int B, int F, int mode,
    float weight, const float* __restrict__ grad_scale_dev, float* __restrict__ partial,
    bf16_t* __restrict__ dpred) {
  __shared__ float red[4];
  const int T = max(Tp, Tt);
  long long cnt = 0;
  for (int b = 0; b < B; ++b) cnt += lens ? min(max(lens[b], 0), T) : T;
  const float inv_n = cnt > 0 ? 1.f / ((float)cnt * (float)F) : 0.f;
  const float gs = weight * inv_n * (grad_scale_dev ? *grad_scale_dev : 1.f);
  const long long row0 = (long long)blockIdx.x * kTtsRowsPerBlock;
  float sum = 0.f;
  for (int r = 0; r < kTtsRowsPerBlock; ++r) {
    const long long row = row0 + r;
    if (row >= (long long)B * T) break;
    const int b = (int)(row / T), t = (int)(row - (long long)b * T);
    const bool live = !lens || t < lens[b];
    const bool has_p = t < Tp, has_t = t < Tt;
    const long long prow = ((long long)b * Tp + t) * ld_p, trow = ((long long)b * Tt + t) * ld_t;
    for (int f = threadIdx.x; f < F; f += 256) {
      float g = 0.f;
      if (live) {
        const float p = has_p ? bf2f(pred[prow + f]) : 0.f, y = has_t ? target[trow + f] : target_pad;
        if (mode == 0) { const float d = p - y; sum += d * d; g = 2.f * d; }
        else if (mode == 1) { const float d = p - y; sum += fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        else {
          sum += fmaxf(p, 0.f) - p * y + log1pf(__expf(-fabsf(p)));
          g = 1.f / (1.f + __expf(-p)) - y;
        }
      }
      if (dpred && has_p) dpred[prow + f] = f2bf(g * gs);
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void tts_loss_finalize_kernel(
    const float* __restrict__ partial, int nparts, const int32_t* __restrict__ lens, int B, int T,
    int F, float weight, float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long cnt = 0;
    for (int b = 0; b < B; ++b) cnt += lens ? min(max(lens[b], 0), T) : T;
    if (cnt > 0) loss[0] += (float)(red[0] * (double)weight / ((double)cnt * (double)F));
  }
}

// y = exp(x);  dx = dy * y
__global__ void exp_fwd_kernel(const bf16_t* __restrict__ x, long long n8, bf16_t* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(__expf(bflo(v[e])), __expf(bfhi(v[e])));
    *reinterpret_cast<u32x4*>(y + i * 8) = o;
  }
}
__global__ void mul_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long long n8,
                                bf16_t* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(a + i * 8);
    const u32x4 w = *reinterpret_cast<const u32x4*>(b + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(bflo(v[e]) * bflo(w[e]), bfhi(v[e]) * bfhi(w[e]));
    *reinterpret_cast<u32x4*>(y + i * 8) = o;
  }
}

// y = tanh(x);  dx = dy * (1 - y^2)
__global__ void tanh_fwd_kernel(const bf16_t* __restrict__ x, long long n8, bf16_t* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(tanhf(bflo(v[e])), tanhf(bfhi(v[e])));
    *reinterpret_cast<u32x4*>(y + i * 8) = o;
  }
}
__global__ void tanh_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, long long n8,
                                bf16_t* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 g = *reinterpret_cast<const u32x4*>(dy + i * 8);
    const u32x4 v = *reinterpret_cast<const u32x4*>(y + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bflo(v[e]), b = bfhi(v[e]);
      o[e] = pack2bf(bflo(g[e]) * (1.f - a * a), bfhi(g[e]) * (1.f - b * b));
    }
    *reinterpret_cast<u32x4*>(dx + i * 8) = o;
  }
}

// out[b, c] (+)= sum_t x[b, t, c]   (x rows ld apart; fp32 out)
__global__ __launch_bounds__(256) void sum_time_kernel(const bf16_t* __restrict__ x, long long ld,
                                                       int T, int C, float* __restrict__ out,
                                                       int accumulate) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += bf2f(x[((long long)b * T + t) * ld + c]);
  float* o = out + (long long)b * C + c;
  *o = accumulate ? *o + s : s;
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_tts_loss_num_parts(int B, int T) {
  return ceil_div((long long)B * T, kTtsRowsPerBlock);
}

extern "C" int os2s_tts_loss_padded(os2s_stream_t stream_, const uint16_t* pred, long long ld_pred, int T_pred,
                                    const float* target, long long ld_target, int T_target, float target_pad,
                                    const int32_t* lens, int B, int F, int mode, float weight,
                                    const float* grad_scale_dev, float* partial, float* loss, uint16_t* dpred) {
  OS2S_REQUIRE(pred && target && partial && loss && B >= 1 && T_pred >= 1 && T_target >= 1 && F >= 1 &&
               mode >= 0 && mode <= 2);
  OS2S_REQUIRE(ld_pred >= F && ld_target >= F);
  hipStream_t stream = (hipStream_t)stream_;
  const int T = std::max(T_pred, T_target);
  const int nparts = os2s_tts_loss_num_parts(B, T);
  OS2S_LAUNCH(tts_loss_kernel, dim3(nparts), dim3(256), 0, stream, (const bf16_t*)pred, ld_pred, T_pred, target,
              ld_target, T_target, target_pad, lens, B, F, mode, weight, grad_scale_dev, partial, (bf16_t*)dpred);
  OS2S_LAUNCH(tts_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, nparts, lens, B, T, F,
              weight, loss);
  return OS2S_OK;
}

extern "C" int os2s_tts_loss(os2s_stream_t stream_, const uint16_t* pred, long long ld_pred,
                             const float* target, long long ld_target, const int32_t* lens, int B,
                             int T, int F, int mode, float weight, const float* grad_scale_dev,
                             float* partial, float* loss, uint16_t* dpred) {
  return os2s_tts_loss_padded(stream_, pred, ld_pred, T, target, ld_target, T, 0.f, lens, B, F, mode, weight,
                              grad_scale_dev, partial, loss, dpred);
}

static int ew_grid(long long n8) { return (int)std::min<long long>((n8 + 255) / 256, 4096); }

extern "C" int os2s_exp_fwd(os2s_stream_t stream, const uint16_t* x, long long n, uint16_t* y) {
  OS2S_REQUIRE(x && y && n >= 0 && n % 8 == 0);
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(exp_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
              n / 8, (bf16_t*)y);
  return OS2S_OK;
}

extern "C" int os2s_tanh_fwd(os2s_stream_t stream, const uint16_t* x, long long n, uint16_t* y) {
  OS2S_REQUIRE(x && y && n >= 0 && n % 8 == 0);
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(tanh_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
              n / 8, (bf16_t*)y);
  return OS2S_OK;
}

extern "C" int os2s_tanh_bwd(os2s_stream_t stream, const uint16_t* dy, const uint16_t* y, long long n,
                             uint16_t* dx) {
  OS2S_REQUIRE(dy && y && dx && n >= 0 && n % 8 == 0);
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(tanh_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
              (const bf16_t*)y, n / 8, (bf16_t*)dx);
  return OS2S_OK;
}

extern "C" int os2s_mul_bf16(os2s_stream_t stream, const uint16_t* a, const uint16_t* b, long long n,
                             uint16_t* y) {
  OS2S_REQUIRE(a && b && y && n >= 0 && n % 8 == 0);
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(mul_bf16_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
              (const bf16_t*)b, n / 8, (bf16_t*)y);
  return OS2S_OK;
}

extern "C" int os2s_sum_time(os2s_stream_t stream, const uint16_t* x, long long ld, int B, int T, int C,
                             float* out, int accumulate) {
  OS2S_REQUIRE(x && out && B >= 1 && T >= 1 && C >= 1 && ld >= C);
  OS2S_LAUNCH(sum_time_kernel, dim3(ceil_div(C, 256), B), dim3(256), 0, (hipStream_t)stream,
              (const bf16_t*)x, ld, T, C, out, accumulate);
  return OS2S_OK;
}
