// Mixed-precision optimizer step as three multi-tensor kernels over FLAT fp32
// buffers (master weights, gradients, optimizer state), HBM-bound, no host sync.
//
// Reference (open_seq2seq/optimizers):
//   mp_wrapper.py:44-122   MixedPrecisionOptimizerWrapper: loss*scale -> grads ->
//       fp32 (+ scale*grad(reg) on the master copy) -> *1/scale -> [all-reduce] ->
//       post-process (clip | LARC) -> NaN/Inf + amax check -> scaler update ->
//       cond(skip) -> inner optimizer on fp32 masters -> saturate_cast to fp16.
//   optimizers.py:333-377  LARC;  :408-482 clip by global norm.
//   automatic_loss_scaler.py:50-110 Backoff, :113-203 LogMax.
//   novograd.py:93-126     NovoGrad (per-tensor 2nd moment, eps INSIDE the sqrt,
//       weight decay after normalisation, then TF Momentum m = b1*m + g; w -= lr*m).
//   lr_policies.py         fixed / poly / exp / cosine / transformer policies.
// Layout: every tensor starts at a multiple of kChunk elements in the flat
// buffers (zero padded), so chunk c belongs to exactly one tensor
// (chunk_tensor[c]); per-tensor scalars live in small device arrays.
// bf16 here plays the role of the reference's fp16 compute copy.
#include "os2s_common.hpp"

namespace os2s {

constexpr int kChunk = 4096;

struct OptDeviceState {   // lives in device memory (os2s_opt_state_bytes())
  long long global_step;
  long long scaler_iteration;
  long long last_overflow_iteration;
  long long num_skipped;
  float loss_scale;
  float lr;
  float global_grad_norm;
  float grad_amax;
  int has_nan;
  int skip;
  // LogMax scaler state
  float x_hat, slow_x_hat, xsquared_hat, b1_correction, b2_correction;
  float grad_scale_latched;  // scale the gradients in the buffer were produced with
};

}  // namespace os2s

using namespace os2s;

namespace os2s {

__device__ float lr_policy_value(const os2s_opt_config_t& c, long long step) {
  float lr = c.learning_rate;
  const float fs = (float)step;
  switch (c.lr_policy) {
    case 1: {  // poly_decay (lr_policies.py:83-128 + tf.train.polynomial_decay)
      if (c.warmup_steps > 0 && step < c.warmup_steps) lr = lr * fs / (float)c.warmup_steps;
      if (step < c.begin_decay_at) return lr;
      long long s = step - c.begin_decay_at;
      if (s > c.decay_steps) s = c.decay_steps;
      const float frac = 1.f - (float)s / (float)c.decay_steps;
      return (lr - c.min_lr) * powf(frac, c.power) + c.min_lr;
    }
    case 2: {  // exp_decay
      if (step >= c.begin_decay_at) {
        float p = (float)(step - c.begin_decay_at) / (float)c.decay_steps;
        if (c.use_staircase_decay) p = floorf(p);
        lr = lr * powf(c.decay_rate, p);
      }
      return fmaxf(c.min_lr, lr);
    }
    case 3: {  // transformer_policy
      const float ws = (float)c.warmup_steps;
      const float decay = c.coefficient * powf((float)c.d_model, -0.5f) *
                          fminf((fs + 1.f) * powf(ws, -1.5f), powf(fs + 1.f, -0.5f));
      const float nl = decay * lr;
      return c.has_max_lr ? fminf(c.max_lr, nl) : nl;
    }
    case 4: {  // cosine_decay (tf.train.cosine_decay, alpha = min_lr)
      if (c.warmup_steps > 0 && step < c.warmup_steps) lr = lr * fs / (float)c.warmup_steps;
      if (step < c.begin_decay_at) return lr;
      long long s = step - c.begin_decay_at;
      if (s > c.decay_steps) s = c.decay_steps;
      const float cosd = 0.5f * (1.f + cosf(3.14159265358979323846f * (float)s / (float)c.decay_steps));
      return lr * ((1.f - c.min_lr) * cosd + c.min_lr);
    }
    case 5: {  // piecewise_constant (lr_policies.py:30-57 + tf.train.piecewise_constant: x <= b[0] -> v[0],
               // b[i-1] < x <= b[i] -> v[i], x > b[-1] -> v[-1])
      int i = 0;
      while (i < c.pw_count && step > c.pw_boundaries[i]) ++i;
      return lr * c.pw_rates[i];
    }
    case 6: {  // inv_poly_decay (lr_policies.py:203-245): lr / (1 + scale * step)^power,
               // scale = ((lr / min_lr)^(1/power) - 1) / decay_steps, min_lr clamped to [1e-8, lr]
      const float mn = fminf(fmaxf(c.min_lr, 1e-8f), lr);
      const float scale = (powf(lr / mn, 1.f / c.power) - 1.f) / (float)c.decay_steps;
      return lr / powf(1.f + scale * fs, c.power);
    }
    default:
      return lr;
  }
}

// ---- pass 1: per-chunk statistics of the effective gradient ----------------
// g_eff = g * inv_scale + l2[tensor] * w      (mp_wrapper.py:79-95)
// partial[c] = {sum g_eff^2, sum w^2, max |g_eff|, nan flag}
// NEED_W = false: no tensor has an l2 term and LARC is off — the weights are not read at all (a third of
// this pass's 2.7 GB for Transformer-big; round 5)
template <bool NEED_W>
__global__ __launch_bounds__(256) void mt_grad_stats_kernel(
    const float* __restrict__ grads, const float* __restrict__ weights,
    const int32_t* __restrict__ chunk_tensor, const float* __restrict__ tensor_l2,
    const OptDeviceState* __restrict__ st, int world_size, float* __restrict__ partial) {
  __shared__ float red[4][4];
  const int c = blockIdx.x;
  const int ti = chunk_tensor[c];
  const float inv = 1.f / (st->grad_scale_latched * (float)world_size);
  const float l2 = tensor_l2 ? tensor_l2[ti] : 0.f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(grads + (long long)c * kChunk);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(weights + (long long)c * kChunk);
  float sg = 0.f, sw = 0.f, mx = 0.f, nan = 0.f;
#pragma unroll
  for (int i = 0; i < kChunk / 4 / 256; ++i) {
    const f32x4 g = g4[i * 256 + threadIdx.x];
    f32x4 w = {0.f, 0.f, 0.f, 0.f};
    if (NEED_W) w = w4[i * 256 + threadIdx.x];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = NEED_W ? g[e] * inv + l2 * w[e] : g[e] * inv;
      sg += ge * ge;
      sw += w[e] * w[e];
      mx = fmaxf(mx, fabsf(ge));   // fmaxf drops NaN -> tracked separately
      nan = (ge != ge) ? 1.f : nan;
    }
  }
  sg = wave_sum(sg); sw = wave_sum(sw); mx = wave_max(mx); nan = wave_max(nan);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[wid][0] = sg; red[wid][1] = sw; red[wid][2] = mx; red[wid][3] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, m = 0.f, n = 0.f;
    for (int w = 0; w < 4; ++w) { a += red[w][0]; b += red[w][1]; m = fmaxf(m, red[w][2]); n = fmaxf(n, red[w][3]); }
    partial[c * 4 + 0] = a; partial[c * 4 + 1] = b; partial[c * 4 + 2] = m; partial[c * 4 + 3] = n;
  }
}

// ---- pass 2a: one workgroup per tensor: reduce its chunks' partials ------------
__global__ __launch_bounds__(256) void mt_tensor_reduce_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ tensor_chunk_begin,
    float* __restrict__ tensor_gnorm2, float* __restrict__ tensor_wnorm2,
    float* __restrict__ tensor_amax, float* __restrict__ tensor_nan) {
  __shared__ double sh_g[4], sh_w[4];
  __shared__ float sh_m[4], sh_n[4];
  const int t = blockIdx.x;
  const int c0 = tensor_chunk_begin[t], c1 = tensor_chunk_begin[t + 1];
  double g2 = 0.0, w2 = 0.0;
  float mx = 0.f, nn = 0.f;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (long long)c * 4);
    g2 += (double)v[0];
    w2 += (double)v[1];
    mx = fmaxf(mx, v[2]);
    nn = fmaxf(nn, v[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    g2 += __shfl_xor(g2, o, 64);
    w2 += __shfl_xor(w2, o, 64);
  }
  mx = wave_max(mx);
  nn = wave_max(nn);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sh_g[wid] = g2; sh_w[wid] = w2; sh_m[wid] = mx; sh_n[wid] = nn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    g2 = (sh_g[0] + sh_g[1]) + (sh_g[2] + sh_g[3]);
    w2 = (sh_w[0] + sh_w[1]) + (sh_w[2] + sh_w[3]);
    mx = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    nn = fmaxf(fmaxf(sh_n[0], sh_n[1]), fmaxf(sh_n[2], sh_n[3]));
    tensor_gnorm2[t] = (float)g2;
    tensor_wnorm2[t] = (float)w2;
    tensor_amax[t] = mx;
    tensor_nan[t] = (nn > 0.f || g2 != g2) ? 1.f : 0.f;
  }
}

// ---- pass 2b: one workgroup: global norm, post-processing factors, NaN/Inf
//      decision, loss-scaler update, lr, NovoGrad second moments ----------------
__global__ __launch_bounds__(256) void mt_finalize_kernel(
    int ntensors, os2s_opt_config_t cfg, OptDeviceState* __restrict__ st,
    float* __restrict__ tensor_gnorm2, float* __restrict__ tensor_wnorm2,
    float* __restrict__ tensor_amax, float* __restrict__ tensor_mult /* in: nan flags */,
    float* __restrict__ tensor_v /* NovoGrad 2nd moments */) {
  __shared__ float sh_f[256];
  __shared__ float sh_m[256];
  __shared__ int sh_nan;
  __shared__ float s_gn, s_amax;
  if (threadIdx.x == 0) sh_nan = 0;
  __syncthreads();
  float tot = 0.f;
  int anynan = 0;
  for (int t = threadIdx.x; t < ntensors; t += 256) {
    tot += tensor_gnorm2[t];
    if (tensor_mult[t] > 0.f) anynan = 1;
  }
  if (anynan) atomicOr(&sh_nan, 1);
  sh_f[threadIdx.x] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += (double)sh_f[i];
    s_gn = (float)sqrt(s);
  }
  __syncthreads();
  const float gnorm = s_gn;
  if (threadIdx.x == 0 && cfg.clip_global_norm > 0.f && !(gnorm == gnorm)) sh_nan = 1;
  const long long step = st->global_step;
  const float lr = lr_policy_value(cfg, step);
  // clip by global norm: scale = clip * min(1/norm, 1/clip)  (optimizers.py:447-449)
  float clip_scale = 1.f;
  if (cfg.clip_global_norm > 0.f)
    clip_scale = cfg.clip_global_norm * fminf(1.f / gnorm, 1.f / cfg.clip_global_norm);
  // per-tensor post-processing factor and the amax of the post-processed grads
  float amax = 0.f;
  for (int t = threadIdx.x; t < ntensors; t += 256) {
    float f = clip_scale;
    if (cfg.use_larc) {
      const float vn = sqrtf(tensor_wnorm2[t]);
      const float gn = sqrtf(tensor_gnorm2[t]) * clip_scale;
      float u;
      if (!cfg.larc_mode_scale) {
        u = fmaxf(cfg.larc_eta * vn / (lr * (gn + cfg.larc_epsilon)), cfg.larc_min_update);
        u = fminf(u, 1.f);
      } else {
        u = fmaxf(cfg.larc_eta * vn / (gn + cfg.larc_epsilon), cfg.larc_min_update);
      }
      f *= u;
    }
    tensor_mult[t] = f;          // provisional: post-processing factor only
    // an Inf gradient under global-norm clipping: norm = inf -> clip scale 0 -> the clipped
    // gradient is inf * 0 = NaN in the reference (_clip_by_global_norm), check_grads sees has_nan
    // and the step is skipped. fmaxf would silently drop that NaN, so test the product itself.
    const float am = tensor_amax[t] * f;
    amax = fmaxf(amax, am);
    if (!(f == f) || !(am == am)) atomicOr(&sh_nan, 1);
  }
  sh_m[threadIdx.x] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int i = 0; i < 256; ++i) m = fmaxf(m, sh_m[i]);
    s_amax = m;
    const int has_nan = sh_nan;
    const bool overflow = has_nan || isinf(m);
    st->has_nan = has_nan;
    st->grad_amax = m;
    st->global_grad_norm = gnorm;
    st->lr = lr;
    int skip = 0;
    if (cfg.scaler == 1) {          // BackoffScaler.update_op
      skip = overflow;
      float sc = st->loss_scale;
      if (overflow) {
        sc = fminf(fmaxf(sc / cfg.step_factor, cfg.scale_min), cfg.scale_max);
        st->last_overflow_iteration = st->scaler_iteration;
      } else {
        const long long since = st->scaler_iteration - st->last_overflow_iteration;
        if (since % cfg.step_window == 0)
          sc = fminf(fmaxf(sc * cfg.step_factor, cfg.scale_min), cfg.scale_max);
      }
      st->loss_scale = sc;
      st->scaler_iteration += 1;
    } else if (cfg.scaler == 2) {   // LogMaxScaler.update_op
      skip = overflow;
      const float x = overflow ? powf(2.f, cfg.log_max) : (logf(m) / logf(2.f));
      st->x_hat = cfg.lm_beta1 * st->x_hat + (1.f - cfg.lm_beta1) * x;
      st->b1_correction *= cfg.lm_beta1;
      const float mu = st->x_hat / (1.f - st->b1_correction);
      st->slow_x_hat = cfg.lm_beta2 * st->slow_x_hat + (1.f - cfg.lm_beta2) * x;
      st->xsquared_hat = cfg.lm_beta2 * st->xsquared_hat + (1.f - cfg.lm_beta2) * (x * x);
      st->b2_correction *= cfg.lm_beta2;
      const float e_x2 = st->xsquared_hat / (1.f - st->b2_correction);
      const float slow_mu = st->slow_x_hat / (1.f - st->b2_correction);
      const float sigma = sqrtf(fmaxf(e_x2 - slow_mu * slow_mu, 0.f));
      const float log_cutoff = sigma * cfg.overflow_std_dev + mu;
      const float proposed = powf(2.f, 16.f - log_cutoff);
      st->loss_scale = fminf(fmaxf(proposed, cfg.scale_min), cfg.scale_max);
      st->scaler_iteration += 1;
    }
    // without a loss scaler the reference applies the update unconditionally
    st->skip = skip;
    if (skip) st->num_skipped += 1;
  }
  __syncthreads();
  if (st->skip) return;
  // NovoGrad second moment on the post-processed grad; 1/sqrt(v + eps) (and grad_averaging) is
  // folded into the per-tensor multiplier. The reference graph AS WRITTEN (novograd.py:100-115)
  // reads the nvgrad2_ema variables but never assigns them — the tf.cond result only replaces
  // the Python list entry — so they stay 0 and every step takes the `g_2` branch: v_t = |g_t|^2,
  // beta2 unused. That is the default here (novograd_ema = 0: results identical to the
  // reference); novograd_ema = 1 keeps the moving average of the published algorithm.
  if (cfg.optimizer == 2) {
    for (int t = threadIdx.x; t < ntensors; t += 256) {
      const float f = tensor_mult[t];
      const float g2 = tensor_gnorm2[t] * f * f;
      float v = tensor_v[t];
      v = (v == 0.f || !cfg.novograd_ema) ? g2 : (v * cfg.beta2 + g2 * (1.f - cfg.beta2));
      tensor_v[t] = v;
      tensor_mult[t] = f / sqrtf(v + cfg.epsilon);
    }
  }
  if (threadIdx.x == 0) st->global_step = step + 1;
}

// ---- pass 3: apply -----------------------------------------------------------
// One chunk per workgroup, 16 elements per thread. The optimizer is a template parameter and every
// load of the thread (4 x {grad, weight, moments}) is issued before the first use: with the
// optimizer as a run-time switch hipcc emitted load - s_waitcnt vmcnt(0) - switch - store per 4
// elements, i.e. 3 loads in flight per thread and a branch per element.
// chunk0: first chunk of the range this launch covers (os2s_opt_apply_range: the apply pass cut into ranges, an event
// after each, so that the next step's forward pass can start on the variables already updated). zero_grads: the
// gradient chunk is written back as zeros once it has been read — the next step needs no separate fill of the
// gradient buffer, which would have to wait for the whole apply pass (a skipped step zeroes without applying).
template <int OPT>
__global__ __launch_bounds__(256) void mt_apply_kernel(
    float* __restrict__ grads, float* __restrict__ weights, float* __restrict__ m1,
    float* __restrict__ m2, bf16_t* __restrict__ w16, const int32_t* __restrict__ chunk_tensor,
    const float* __restrict__ tensor_l2, const float* __restrict__ tensor_mult,
    const float* __restrict__ tensor_wd_mask, os2s_opt_config_t cfg,
    const OptDeviceState* __restrict__ st, int chunk0, int zero_grads) {
  constexpr int N = kChunk / 4 / 256;
  const int c = blockIdx.x + chunk0;
  if (st->skip) {
    if (zero_grads) {
      const long long b0 = (long long)c * kChunk + (long long)threadIdx.x * 4;
#pragma unroll
      for (int i = 0; i < N; ++i) *reinterpret_cast<f32x4*>(grads + b0 + (long long)i * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const int ti = chunk_tensor[c];
  // st->loss_scale was already updated by the finalize pass; the gradients in
  // the buffer were produced with the PREVIOUS scale, saved in st->grad_scale_latched.
  const float inv = 1.f / (st->grad_scale_latched * (float)cfg.world_size);
  const float l2 = tensor_l2 ? tensor_l2[ti] : 0.f;
  const float mult = tensor_mult[ti];
  const float lr = st->lr;
  const float wd = cfg.weight_decay * (tensor_wd_mask ? tensor_wd_mask[ti] : 1.f);
  const float ga = cfg.grad_averaging ? (1.f - cfg.beta1) : 1.f;
  float adam_lr = lr;
  if (OPT == 3) {
    const float t = (float)(st->global_step);  // already incremented: t = step+1
    adam_lr = lr * sqrtf(1.f - powf(cfg.beta2, t)) / (1.f - powf(cfg.beta1, t));
  }
  const long long base = (long long)c * kChunk + (long long)threadIdx.x * 4;
  f32x4 g[N], w[N], m[N], v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const long long off = base + (long long)i * 1024;
    // the gradient is dead after this pass and the bf16 copy is next read a step later: nontemporal (the
    // pattern alone, tools/probe_streams.hip: 5.89 -> 6.14 TB/s)
    g[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grads + off));
    w[i] = *reinterpret_cast<const f32x4*>(weights + off);
    if (OPT != 0) m[i] = *reinterpret_cast<const f32x4*>(m1 + off);
    if (OPT == 3) v[i] = *reinterpret_cast<const f32x4*>(m2 + off);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const long long off = base + (long long)i * 1024;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = (g[i][e] * inv + l2 * w[i][e]) * mult;
      if (OPT == 0) {
        w[i][e] -= lr * ge;
      } else if (OPT == 1) {
        m[i][e] = cfg.beta1 * m[i][e] + ge;
        w[i][e] -= lr * m[i][e];
      } else if (OPT == 2) {
        ge += wd * w[i][e];
        ge *= ga;
        m[i][e] = cfg.beta1 * m[i][e] + ge;
        w[i][e] -= lr * m[i][e];
      } else {
        m[i][e] = cfg.beta1 * m[i][e] + (1.f - cfg.beta1) * ge;
        v[i][e] = cfg.beta2 * v[i][e] + (1.f - cfg.beta2) * ge * ge;
        w[i][e] -= adam_lr * m[i][e] / (sqrtf(v[i][e]) + cfg.epsilon);
      }
    }
    *reinterpret_cast<f32x4*>(weights + off) = w[i];
    if (zero_grads) *reinterpret_cast<f32x4*>(grads + off) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (OPT != 0) *reinterpret_cast<f32x4*>(m1 + off) = m[i];
    if (OPT == 3) *reinterpret_cast<f32x4*>(m2 + off) = v[i];
    if (w16) {
      u32x2 o;
      o[0] = pack2bf(w[i][0], w[i][1]);
      o[1] = pack2bf(w[i][2], w[i][3]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(w16 + off));
    }
  }
}

// remember the scale the current gradients were produced with
__global__ void opt_latch_scale_kernel(OptDeviceState* st) { st->grad_scale_latched = st->loss_scale; }

__global__ void opt_init_state_kernel(OptDeviceState* st, float loss_scale) {
  st->global_step = 0; st->scaler_iteration = 0; st->last_overflow_iteration = -1;
  st->num_skipped = 0; st->loss_scale = loss_scale; st->lr = 0.f; st->global_grad_norm = 0.f;
  st->grad_amax = 0.f; st->has_nan = 0; st->skip = 0; st->x_hat = 0.f; st->slow_x_hat = 0.f;
  st->xsquared_hat = 0.f; st->b1_correction = 1.f; st->b2_correction = 1.f; st->grad_scale_latched = loss_scale;
}

// fp32 -> bf16 copy of a flat buffer (initial compute copies / broadcast)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src,
                                                            bf16_t* __restrict__ dst,
                                                            long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
       i += (long long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    u32x2 o;
    o[0] = pack2bf(v[0], v[1]);
    o[1] = pack2bf(v[2], v[3]);
    reinterpret_cast<u32x2*>(dst)[i] = o;
  }
}

// Batched "dgrad copy" of conv weights: wT[k'][ci][co] = w[K-1-k'][co][ci].
struct WtDesc { long long src_off, dst_off; int K, Cout, Cin, tile_begin; };

__global__ __launch_bounds__(256) void conv_weight_dgrad_copy_kernel(
    const bf16_t* __restrict__ w16, bf16_t* __restrict__ wt16, const WtDesc* __restrict__ descs,
    int ndesc) {
  __shared__ bf16_t tile[64][66];
  // find descriptor by tile index (binary search over tile_begin)
  const int tileid = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= tileid) lo = mid; else hi = mid - 1;
  }
  const WtDesc d = descs[lo];
  const int local = tileid - d.tile_begin;
  const int tco = (d.Cout + 63) / 64, tci = (d.Cin + 63) / 64;
  const int k = local / (tco * tci);
  const int r = local - k * (tco * tci);
  const int co0 = (r / tci) * 64, ci0 = (r % tci) * 64;
  const bf16_t* src = w16 + d.src_off + (long long)k * d.Cout * d.Cin;
  bf16_t* dst = wt16 + d.dst_off + (long long)(d.K - 1 - k) * d.Cin * d.Cout;
  // 16-byte pieces on both sides when the rows allow it (every conv / dense layer of the models: channel
  // counts are multiples of 8): 2-byte accesses moved 2.6 TB/s, a quarter of the instructions were
  // address arithmetic per element
  const bool vec = (d.Cin % 8 == 0) && (d.Cout % 8 == 0) && ((((uintptr_t)src) | ((uintptr_t)dst)) % 16 == 0);
  if (vec) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int piece = it * 256 + threadIdx.x;
      const int a = piece >> 3, b8 = piece & 7;          // a: co, 8 consecutive ci
      const int co = co0 + a, ci = ci0 + b8 * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (co < d.Cout && ci < d.Cin) v = *reinterpret_cast<const u32x4*>(src + (long long)co * d.Cin + ci);
      uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[a][b8 * 8]);     // 132-byte rows: 4-byte aligned
      t32[0] = v[0]; t32[1] = v[1]; t32[2] = v[2]; t32[3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int piece = it * 256 + threadIdx.x;
      const int a = piece >> 3, b8 = piece & 7;          // a: ci, 8 consecutive co
      const int ci = ci0 + a, co = co0 + b8 * 8;
      if (ci < d.Cin && co < d.Cout) {
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = (uint32_t)tile[b8 * 8 + 2 * e][a] | ((uint32_t)tile[b8 * 8 + 2 * e + 1][a] << 16);
        *reinterpret_cast<u32x4*>(dst + (long long)ci * d.Cout + co) = v;
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int a = i >> 6, b = i & 63;   // a: co, b: ci (contiguous)
    const int co = co0 + a, ci = ci0 + b;
    tile[a][b] = (co < d.Cout && ci < d.Cin) ? src[(long long)co * d.Cin + ci] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int a = i >> 6, b = i & 63;   // a: ci, b: co (contiguous)
    const int ci = ci0 + a, co = co0 + b;
    if (co < d.Cout && ci < d.Cin) dst[(long long)ci * d.Cout + co] = tile[b][a];
  }
}

}  // namespace os2s

extern "C" int os2s_opt_chunk_elems(void) { return kChunk; }
extern "C" size_t os2s_opt_state_bytes(void) { return sizeof(OptDeviceState); }
extern "C" size_t os2s_opt_config_bytes(void) { return sizeof(os2s_opt_config_t); }

extern "C" int os2s_opt_init_state(os2s_stream_t stream, void* state, float loss_scale) {
  OS2S_REQUIRE(state);
  OS2S_LAUNCH(opt_init_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
              (OptDeviceState*)state, loss_scale);
  return OS2S_OK;
}

// One optimisation step over `nchunks` chunks of kChunk elements.
//   m1/m2: optimizer state (momentum / Adam m and v), w16: bf16 compute copy (may be NULL)
//   partial: scratch [nchunks*4]; tensor_* arrays: [ntensors] (tensor_chunk_begin: [ntensors+1])
// = os2s_opt_prepare (statistics of the gradient, overflow / skip decision, loss-scale update, learning rate, LARC /
// NovoGrad factors: everything that needs ALL gradients) + os2s_opt_apply_range over all chunks.
extern "C" int os2s_opt_prepare(os2s_stream_t stream_, const os2s_opt_config_t* cfg, void* state,
                                const float* grads, const float* weights, int nchunks, int ntensors,
                                const int32_t* chunk_tensor, const int32_t* tensor_chunk_begin,
                                const float* tensor_l2, float* partial, float* tensor_gnorm2, float* tensor_wnorm2,
                                float* tensor_amax, float* tensor_mult, float* tensor_v) {
  OS2S_REQUIRE(cfg && state && grads && weights && chunk_tensor && tensor_chunk_begin);
  OS2S_REQUIRE(partial && tensor_gnorm2 && tensor_wnorm2 && tensor_amax && tensor_mult);
  OS2S_REQUIRE(nchunks >= 1 && ntensors >= 1 && cfg->world_size >= 1);
  if (cfg->optimizer == 2) OS2S_REQUIRE(tensor_v);
  hipStream_t stream = (hipStream_t)stream_;
  OptDeviceState* st = (OptDeviceState*)state;
  OS2S_LAUNCH(opt_latch_scale_kernel, dim3(1), dim3(1), 0, stream, st);
  if (tensor_l2 != nullptr || cfg->use_larc) {
    OS2S_LAUNCH(mt_grad_stats_kernel<true>, dim3(nchunks), dim3(256), 0, stream, grads, weights,
                chunk_tensor, tensor_l2, st, cfg->world_size, partial);
  } else {
    OS2S_LAUNCH(mt_grad_stats_kernel<false>, dim3(nchunks), dim3(256), 0, stream, grads, weights,
                chunk_tensor, tensor_l2, st, cfg->world_size, partial);
  }
  OS2S_LAUNCH(mt_tensor_reduce_kernel, dim3(ntensors), dim3(256), 0, stream, partial,
              tensor_chunk_begin, tensor_gnorm2, tensor_wnorm2, tensor_amax, tensor_mult);
  OS2S_LAUNCH(mt_finalize_kernel, dim3(1), dim3(256), 0, stream, ntensors, *cfg, st,
              tensor_gnorm2, tensor_wnorm2, tensor_amax, tensor_mult, tensor_v);
  return OS2S_OK;
}

extern "C" int os2s_opt_apply_range(os2s_stream_t stream_, const os2s_opt_config_t* cfg, const void* state,
                                    float* grads, float* weights, float* m1, float* m2, uint16_t* w16,
                                    int chunk_begin, int chunk_end, const int32_t* chunk_tensor,
                                    const float* tensor_l2, const float* tensor_mult,
                                    const float* tensor_wd_mask, int zero_grads) {
  OS2S_REQUIRE(cfg && state && grads && weights && chunk_tensor && tensor_mult);
  OS2S_REQUIRE(chunk_begin >= 0 && chunk_end >= chunk_begin);
  if (cfg->optimizer != 0) OS2S_REQUIRE(m1);
  if (cfg->optimizer == 3) OS2S_REQUIRE(m2);
  if (chunk_end == chunk_begin) return OS2S_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const OptDeviceState* st = (const OptDeviceState*)state;
  const int n = chunk_end - chunk_begin;
#define OS2S_APPLY(OPT)                                                                    \
  OS2S_LAUNCH(mt_apply_kernel<OPT>, dim3(n), dim3(256), 0, stream, grads, weights, m1, m2, \
              w16, chunk_tensor, tensor_l2, tensor_mult, tensor_wd_mask, *cfg, st, chunk_begin, zero_grads ? 1 : 0)
  switch (cfg->optimizer) {
    case 0: OS2S_APPLY(0); break;
    case 1: OS2S_APPLY(1); break;
    case 2: OS2S_APPLY(2); break;
    default: OS2S_APPLY(3); break;
  }
#undef OS2S_APPLY
  return OS2S_OK;
}

extern "C" int os2s_opt_step(os2s_stream_t stream_, const os2s_opt_config_t* cfg, void* state,
                             const float* grads, float* weights, float* m1, float* m2,
                             uint16_t* w16, int nchunks, int ntensors,
                             const int32_t* chunk_tensor, const int32_t* tensor_chunk_begin,
                             const float* tensor_l2, const float* tensor_wd_mask,
                             float* partial, float* tensor_gnorm2, float* tensor_wnorm2,
                             float* tensor_amax, float* tensor_mult, float* tensor_v) {
  OS2S_REQUIRE(cfg);
  if (cfg->optimizer != 0) OS2S_REQUIRE(m1);
  if (cfg->optimizer == 3) OS2S_REQUIRE(m2);
  const int rc = os2s_opt_prepare(stream_, cfg, state, grads, weights, nchunks, ntensors, chunk_tensor,
                                  tensor_chunk_begin, tensor_l2, partial, tensor_gnorm2, tensor_wnorm2, tensor_amax,
                                  tensor_mult, tensor_v);
  if (rc != OS2S_OK) return rc;
  // (the gradient buffer is const for this entry point's callers: no zeroing)
  return os2s_opt_apply_range(stream_, cfg, state, const_cast<float*>(grads), weights, m1, m2, w16, 0, nchunks,
                              chunk_tensor, tensor_l2, tensor_mult, tensor_wd_mask, 0);
}

extern "C" int os2s_cast_f32_to_bf16(os2s_stream_t stream, const float* src, uint16_t* dst,
                                     long long n) {
  OS2S_REQUIRE(src && dst && n >= 0 && n % 4 == 0);
  if (n == 0) return OS2S_OK;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  OS2S_LAUNCH(cast_f32_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src,
              dst, n / 4);
  return OS2S_OK;
}

// descs: device array of ndesc records {src_off, dst_off, K, Cout, Cin, tile_begin}
// (int64, int64, int32 x4; 32 bytes each); total_tiles = sum K*ceil(Cout/64)*ceil(Cin/64).
extern "C" int os2s_conv_weight_dgrad_copy(os2s_stream_t stream, const uint16_t* w16,
                                           uint16_t* wt16, const void* descs, int ndesc,
                                           int total_tiles) {
  OS2S_REQUIRE(w16 && wt16 && descs && ndesc >= 1 && total_tiles >= 1);
  static_assert(sizeof(WtDesc) == 32, "descriptor layout");
  OS2S_LAUNCH(conv_weight_dgrad_copy_kernel, dim3(total_tiles), dim3(256), 0,
              (hipStream_t)stream, w16, wt16, (const WtDesc*)descs, ndesc);
  return OS2S_OK;
}
