// Token-wise kernels of the Transformer NMT path (HBM-bound, bf16 I/O, fp32 math),
// operating on PACKED token-major tensors [N_tokens, hidden]:
//   * embedding gather * sqrt(d) + sinusoidal position signal + dropout, and its
//     scatter-add gradient   (parts/transformer/embedding_layer.py:59-88,
//     parts/transformer/utils.py:28-54, encoders/transformer_encoder.py:150-161,
//     decoders/transformer_decoder.py:197-213 — the decoder's shift-right is an index
//     remap done by the caller: ids of the previous position, 0 for the first);
//   * LayerNormalization (fp32 statistics, eps 1e-6; parts/transformer/common.py:41-68)
//     forward / backward (the backward also adds the residual-branch gradient of the
//     pre-norm wrapper, common.py:99-106);
//   * dropout / relu-dropout backward helpers;
//   * PaddedCrossEntropyLossWithSmoothing (losses/sequence_loss.py:257-309) fused with
//     its gradient: one read of the [N, V] logits, one write of dlogits.
#include <cstdlib>
#include "os2s_common.hpp"

namespace os2s {

// ---------------------------------------------------------------------------
// embedding
// ---------------------------------------------------------------------------
// out[n, :] = (id valid ? E[id]*sqrt(D) : 0) + posenc(pos[n]) ; then dropout
__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ pos,
    const bf16_t* __restrict__ table, int V, int D, long long N, float emb_scale, float keep_prob,
    unsigned long long seed, bf16_t* __restrict__ out, int plain) {
  const int D8 = D >> 3;
  const int half = D >> 1;
  const float log_inc = logf(1.0e4f) / (float)(half - 1);
  const float ik = 1.f / keep_prob;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N * D8;
       i += (long long)gridDim.x * 256) {
    const long long n = i / D8;
    const int c0 = (int)(i - n * D8) * 8;
    int id = ids[n];
    if (id > V - 1) id = 0;   // out-of-bound ids map to the pad symbol (embedding_layer.py:71-73)
    if (id < 0) id = 0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (id != 0 || plain) {   // padding embeddings are zeroed (:79-87); plain = tf.nn.embedding_lookup
      const u32x4 t = *reinterpret_cast<const u32x4*>(table + (long long)id * D + c0);
      v[0] = bflo(t[0]); v[1] = bfhi(t[0]); v[2] = bflo(t[1]); v[3] = bfhi(t[1]);
      v[4] = bflo(t[2]); v[5] = bfhi(t[2]); v[6] = bflo(t[3]); v[7] = bfhi(t[3]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= emb_scale;
    }
    const float p = pos ? (float)pos[n] : 0.f;
    uint32_t keep = 0xffu;
    if (keep_prob < 1.f) keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      const int j = c < half ? c : c - half;
      const float ang = p * __expf(-(float)j * log_inc);
      float val = v[e] + (pos ? (c < half ? sinf(ang) : cosf(ang)) : 0.f);
      if (keep_prob < 1.f) val = ((keep >> e) & 1u) ? val * ik : 0.f;
      v[e] = val;
    }
    u32x4 o;
    o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
    o[2] = pack2bf(v[4], v[5]); o[3] = pack2bf(v[6], v[7]);
    *reinterpret_cast<u32x4*>(out + n * D + c0) = o;
  }
}

// dE[id] += emb_scale * dropout'(dout[n])   (fp32 atomics into the flat gradient)
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const int32_t* __restrict__ ids, const bf16_t* __restrict__ dout, int V, int D, long long N,
    float emb_scale, float keep_prob, unsigned long long seed, float* __restrict__ dtable,
    int plain) {
  const int D8 = D >> 3;
  const float ik = 1.f / keep_prob;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N * D8;
       i += (long long)gridDim.x * 256) {
    const long long n = i / D8;
    const int c0 = (int)(i - n * D8) * 8;
    int id = ids[n];
    if (id > V - 1 || id < 0) id = 0;
    if (id == 0 && !plain) continue;
    const u32x4 t = *reinterpret_cast<const u32x4*>(dout + n * D + c0);
    float g[8] = {bflo(t[0]), bfhi(t[0]), bflo(t[1]), bfhi(t[1]),
                  bflo(t[2]), bfhi(t[2]), bflo(t[3]), bfhi(t[3])};
    uint32_t keep = 0xffu;
    if (keep_prob < 1.f) keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = g[e] * emb_scale;
      if (keep_prob < 1.f) x = ((keep >> e) & 1u) ? x * ik : 0.f;
      if (x != 0.f)
        __hip_atomic_fetch_add(dtable + (long long)id * D + c0 + e, x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Deterministic mode: one thread per 8-column chunk walks the tokens in order — every element of dE gets
// its adds from ONE thread in token order (no atomics; ~1 us per token: a debugging aid).
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(
    const int32_t* __restrict__ ids, const bf16_t* __restrict__ dout, int V, int D, long long N,
    float emb_scale, float keep_prob, unsigned long long seed, float* __restrict__ dtable,
    int plain) {
  const int D8 = D >> 3;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= D8) return;
  const float ik = 1.f / keep_prob;
  const int c0 = col * 8;
  for (long long n = 0; n < N; ++n) {
    int id = ids[n];
    if (id > V - 1 || id < 0) id = 0;
    if (id == 0 && !plain) continue;
    const long long i = n * D8 + col;
    const u32x4 t = *reinterpret_cast<const u32x4*>(dout + n * D + c0);
    float g[8] = {bflo(t[0]), bfhi(t[0]), bflo(t[1]), bfhi(t[1]),
                  bflo(t[2]), bfhi(t[2]), bflo(t[3]), bfhi(t[3])};
    uint32_t keep = 0xffu;
    if (keep_prob < 1.f) keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
    float* const row = dtable + (long long)id * D + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = g[e] * emb_scale;
      if (keep_prob < 1.f) x = ((keep >> e) & 1u) ? x * ik : 0.f;
      if (x != 0.f) row[e] += x;
    }
  }
}

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row
// ---------------------------------------------------------------------------
template <int VPL>  // 16-byte vectors per lane: D = 64 * 8 * VPL
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, long long N, bf16_t* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  constexpr int D = 64 * 8 * VPL;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float v[VPL][8];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(x + row * D + (u * 64 + lane) * 8);
    v[u][0] = bflo(t[0]); v[u][1] = bfhi(t[0]); v[u][2] = bflo(t[1]); v[u][3] = bfhi(t[1]);
    v[u][4] = bflo(t[2]); v[u][5] = bfhi(t[2]); v[u][6] = bflo(t[3]); v[u][7] = bfhi(t[3]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[u][e];
  }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[u][e] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int c0 = (u * 64 + lane) * 8;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (v[u][e] - mean) * rstd * gamma[c0 + e] + beta[c0 + e];
    u32x4 o;
    o[0] = pack2bf(r[0], r[1]); o[1] = pack2bf(r[2], r[3]);
    o[2] = pack2bf(r[4], r[5]); o[3] = pack2bf(r[6], r[7]);
    *reinterpret_cast<u32x4*>(y + row * D + c0) = o;
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// dx = dres + rstd * (g*dy - mean_D(g*dy) - xhat * mean_D(g*dy*xhat));
// per-block partial sums of dbeta = sum dy, dgamma = sum dy*xhat -> partial[blk][2][D]
//
// One wave per row, kLnWaves rows of a workgroup's block at a time. A row is two dependent steps (two
// wave reductions between its loads and its stores), so the kernel is a latency chain per wave:
// the first version (4 waves x 8 rows, loads of a row issued after the stores of the previous one,
// the residual gradient loaded after the reductions) ran at 1.1 TB/s. Now every load of row i + kLnWaves
// (dy, x, dres, mean, rstd) is issued before row i is reduced — through buffer descriptors over the
// block's rows, so the prefetch past the last row needs no branch: it is out of range, returns zeros
// and costs no memory request — and a workgroup has 8 waves.
constexpr int kLnWaves = 8;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ln_rows_rsrc(const void* base, long long r0, int n, int D) {
  const unsigned long long a = (unsigned long long)base + (unsigned long long)r0 * (unsigned)D * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int bytes = __builtin_amdgcn_readfirstlane(base ? n * D * 2 : 0);   // null tensor: everything is out of range
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

template <int VPL>
struct LnRow {
  u32x4 a[VPL], t[VPL], r[VPL];
  float mu, rs;
};

template <int VPL>
__global__ __launch_bounds__(64 * kLnWaves) void layernorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const bf16_t* __restrict__ dres, long long N, int rows_per_block, bf16_t* __restrict__ dx,
    float* __restrict__ partial) {
  constexpr int D = 64 * 8 * VPL;
  constexpr int kOob = 0x7fffffff;
  __shared__ float red[kLnWaves][D];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float gb[VPL][8], gg[VPL][8], gm[VPL][8];
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + (u * 64 + lane) * 8);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + (u * 64 + lane) * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gm[u][e] = g0[e]; gm[u][4 + e] = g1[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { gb[u][e] = 0.f; gg[u][e] = 0.f; }
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long left = N - r0;
  const int n = (int)(left < rows_per_block ? left : rows_per_block);
  const __amdgpu_buffer_rsrc_t dyr = ln_rows_rsrc(dy, r0, n, D);
  const __amdgpu_buffer_rsrc_t xr = ln_rows_rsrc(x, r0, n, D);
  const __amdgpu_buffer_rsrc_t drr = ln_rows_rsrc(dres, r0, n, D);
  const __amdgpu_buffer_rsrc_t dxr = ln_rows_rsrc(dx, r0, n, D);
  auto fetch = [&](int i, LnRow<VPL>& R) {
    const bool in = i < n;
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
      const int off = in ? i * (D * 2) + (u * 64 + lane) * 16 : kOob;
      R.a[u] = __builtin_amdgcn_raw_buffer_load_b128(dyr, off, 0, 0);
      R.t[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
      R.r[u] = __builtin_amdgcn_raw_buffer_load_b128(drr, off, 0, 0);
    }
    const long long gr = r0 + (in ? i : 0);
    R.mu = mean[gr];
    R.rs = rstd[gr];
  };
  auto process = [&](int i, const LnRow<VPL>& cur) {
    const float mu = cur.mu, rs = cur.rs;
    float dyv[VPL][8], xh[VPL][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
      const u32x4 a = cur.a[u], t = cur.t[u];
      dyv[u][0] = bflo(a[0]); dyv[u][1] = bfhi(a[0]); dyv[u][2] = bflo(a[1]); dyv[u][3] = bfhi(a[1]);
      dyv[u][4] = bflo(a[2]); dyv[u][5] = bfhi(a[2]); dyv[u][6] = bflo(a[3]); dyv[u][7] = bfhi(a[3]);
      xh[u][0] = bflo(t[0]); xh[u][1] = bfhi(t[0]); xh[u][2] = bflo(t[1]); xh[u][3] = bfhi(t[1]);
      xh[u][4] = bflo(t[2]); xh[u][5] = bfhi(t[2]); xh[u][6] = bflo(t[3]); xh[u][7] = bfhi(t[3]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[u][e] = (xh[u][e] - mu) * rs;
        gb[u][e] += dyv[u][e];
        gg[u][e] += dyv[u][e] * xh[u][e];
        const float gd = gm[u][e] * dyv[u][e];
        s1 += gd;
        s2 += gd * xh[u][e];
      }
    }
    s1 = wave_sum_dpp(s1) * (1.f / D);
    s2 = wave_sum_dpp(s2) * (1.f / D);
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = rs * (gm[u][e] * dyv[u][e] - s1 - xh[u][e] * s2);
      const u32x4 t = cur.r[u];            // zeros without a residual gradient
      r[0] += bflo(t[0]); r[1] += bfhi(t[0]); r[2] += bflo(t[1]); r[3] += bfhi(t[1]);
      r[4] += bflo(t[2]); r[5] += bfhi(t[2]); r[6] += bflo(t[3]); r[7] += bfhi(t[3]);
      u32x4 o;
      o[0] = pack2bf(r[0], r[1]); o[1] = pack2bf(r[2], r[3]);
      o[2] = pack2bf(r[4], r[5]); o[3] = pack2bf(r[6], r[7]);
      __builtin_amdgcn_raw_buffer_store_b128(o, dxr, i * (D * 2) + (u * 64 + lane) * 16, 0, 0);
    }
  };
  // two row buffers, used alternately (a register copy `cur = next` would have to wait for the prefetch);
  // sched_barrier: the prefetch is issued where it stands, not where hipcc would sink it to
  LnRow<VPL> ra, rb;
  fetch(wid, ra);
  for (int i = wid; i < n; i += 2 * kLnWaves) {
    fetch(i + kLnWaves, rb);
    __builtin_amdgcn_sched_barrier(0);
    process(i, ra);
    if (i + kLnWaves >= n) break;
    fetch(i + 2 * kLnWaves, ra);
    __builtin_amdgcn_sched_barrier(0);
    process(i + kLnWaves, rb);
  }
  // block reduce the parameter-gradient partials over the waves (fixed order)
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < VPL; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        red[wid][(u * 64 + lane) * 8 + e] = pass == 0 ? gb[u][e] : gg[u][e];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * kLnWaves) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < kLnWaves; ++w) acc += red[w][c];
      partial[((long long)blockIdx.x * 2 + pass) * D + c] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// dropout backward helpers
// ---------------------------------------------------------------------------
// mode 0: d = dout * keepmask/keep (mask from the hash);  mode 1: d = dout * (out > 0)/keep;
// mode 2: d = dout * (0 < out < bf16(20/keep))/keep — out = dropout(min(relu(.), 20)), the clipped ReLU
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const bf16_t* __restrict__ dout,
                                                          const bf16_t* __restrict__ out, int mode,
                                                          float keep_prob, unsigned long long seed,
                                                          long long n8, bf16_t* __restrict__ d) {
  const float ik = 1.f / keep_prob;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8;
       i += (long long)gridDim.x * 256) {
    const u32x4 t = reinterpret_cast<const u32x4*>(dout)[i];
    float g[8] = {bflo(t[0]), bfhi(t[0]), bflo(t[1]), bfhi(t[1]),
                  bflo(t[2]), bfhi(t[2]), bflo(t[3]), bfhi(t[3])};
    if (mode == 0) {
      const uint32_t keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((keep >> e) & 1u) ? g[e] * ik : 0.f;
    } else {
      const u32x4 o = reinterpret_cast<const u32x4*>(out)[i];
      const float ov[8] = {bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1]),
                           bflo(o[2]), bfhi(o[2]), bflo(o[3]), bfhi(o[3])};
      const float cap = mode == 2 ? bflo(pack2bf(20.f * ik, 0.f)) : __builtin_inff();
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (ov[e] > 0.f && ov[e] < cap) ? g[e] * ik : 0.f;
    }
    u32x4 r;
    r[0] = pack2bf(g[0], g[1]); r[1] = pack2bf(g[2], g[3]);
    r[2] = pack2bf(g[4], g[5]); r[3] = pack2bf(g[6], g[7]);
    reinterpret_cast<u32x4*>(d)[i] = r;
  }
}

// The same on a [rows, C] matrix, plus the column sums of d (= the gradient of the layer's bias,
// tf.layers.Dense(use_bias=True) in ffn_layer.py:51-85) from the same pass: a workgroup owns
// kDropColRows rows x up to 2048 columns, a thread 8 columns of every (256 / G)-th row, and writes
// one partial row of sums per workgroup into partial[nparts][2][C] (plane 0; plane 1 is zero —
// the layout os2s_bn_bwd_finalize reduces). Before: a separate pass over d (os2s_bn_stats).
constexpr int kDropColRows = 64;
__global__ __launch_bounds__(256) void dropout_bwd_colsum_kernel(const bf16_t* __restrict__ dout,
                                                                 const bf16_t* __restrict__ out, int mode,
                                                                 float keep_prob, unsigned long long seed,
                                                                 long long rows, int C, int G,
                                                                 bf16_t* __restrict__ d,
                                                                 float* __restrict__ partial) {
  __shared__ float sh[256 * 8];
  const float ik = 1.f / keep_prob;
  const int c8n = C >> 3;
  const int cg = blockIdx.y * G + (int)(threadIdx.x % G), rp = threadIdx.x / G, nrp = 256 / G;
  const long long r0 = (long long)blockIdx.x * kDropColRows;
  const long long r1 = r0 + kDropColRows < rows ? r0 + kDropColRows : rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cg < c8n && rp < nrp) {
    for (long long r = r0 + rp; r < r1; r += nrp) {
      const long long i = r * c8n + cg;
      const u32x4 t = reinterpret_cast<const u32x4*>(dout)[i];
      float g[8] = {bflo(t[0]), bfhi(t[0]), bflo(t[1]), bfhi(t[1]),
                    bflo(t[2]), bfhi(t[2]), bflo(t[3]), bfhi(t[3])};
      if (mode == 0) {
        const uint32_t keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((keep >> e) & 1u) ? g[e] * ik : 0.f;
      } else {
        const u32x4 o = reinterpret_cast<const u32x4*>(out)[i];
        const float ov[8] = {bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1]),
                             bflo(o[2]), bfhi(o[2]), bflo(o[3]), bfhi(o[3])};
        const float cap = mode == 2 ? bflo(pack2bf(20.f * ik, 0.f)) : __builtin_inff();
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (ov[e] > 0.f && ov[e] < cap) ? g[e] * ik : 0.f;
      }
      u32x4 rr;
      rr[0] = pack2bf(g[0], g[1]); rr[1] = pack2bf(g[2], g[3]);
      rr[2] = pack2bf(g[4], g[5]); rr[3] = pack2bf(g[6], g[7]);
      reinterpret_cast<u32x4*>(d)[i] = rr;
      // the sums are those of the ROUNDED values the weight-gradient GEMM will read
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += bflo(rr[e]); acc[2 * e + 1] += bfhi(rr[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if (rp == 0 && cg < c8n) {
    float* const p0 = partial + ((long long)blockIdx.x * 2) * C + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
      for (int q = 0; q < nrp; ++q) s += sh[(q * G + (int)(threadIdx.x % G)) * 8 + e];
      p0[e] = s;
      p0[C + e] = 0.f;
    }
  }
}

// y = residual + dropout(act(y + bias)) in place on bf16 [rows, C]: the epilogue of a Dense
// layer whose matmul ran in the vendor GEMM (same dropout stream as the fused conv epilogue:
// element index / 8 -> dropout_bits8). One HBM pass; C % 8 == 0.
__global__ __launch_bounds__(256) void dense_epilogue_kernel(bf16_t* __restrict__ y,
                                                             const float* __restrict__ bias, int C,
                                                             int act, float keep_prob,
                                                             unsigned long long seed,
                                                             const bf16_t* __restrict__ residual,
                                                             long long n8) {
  const float ik = 1.f / keep_prob;
  const int c8 = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8;
       i += (long long)gridDim.x * 256) {
    const u32x4 t = reinterpret_cast<const u32x4*>(y)[i];
    float g[8] = {bflo(t[0]), bfhi(t[0]), bflo(t[1]), bfhi(t[1]),
                  bflo(t[2]), bfhi(t[2]), bflo(t[3]), bfhi(t[3])};
    if (bias) {
      const int c0 = (int)(i % c8) * 8;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c0);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + c0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] += b0[e]; g[4 + e] += b1[e]; }
    }
    if (act == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = fmaxf(g[e], 0.f);
    }
    if (keep_prob < 1.f) {
      const uint32_t keep = dropout_bits8(seed, (unsigned long long)i, keep_prob);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((keep >> e) & 1u) ? g[e] * ik : 0.f;
    }
    if (residual) {
      const u32x4 r = reinterpret_cast<const u32x4*>(residual)[i];
      g[0] += bflo(r[0]); g[1] += bfhi(r[0]); g[2] += bflo(r[1]); g[3] += bfhi(r[1]);
      g[4] += bflo(r[2]); g[5] += bfhi(r[2]); g[6] += bflo(r[3]); g[7] += bfhi(r[3]);
    }
    u32x4 o;
    o[0] = pack2bf(g[0], g[1]); o[1] = pack2bf(g[2], g[3]);
    o[2] = pack2bf(g[4], g[5]); o[3] = pack2bf(g[6], g[7]);
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}

// out = a + b (bf16), used to sum gradient contributions of multi-consumer activations
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a,
                                                       const bf16_t* __restrict__ b, long long n8,
                                                       bf16_t* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8;
       i += (long long)gridDim.x * 256) {
    const u32x4 x = reinterpret_cast<const u32x4*>(a)[i];
    const u32x4 y = reinterpret_cast<const u32x4*>(b)[i];
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack2bf(bflo(x[e]) + bflo(y[e]), bfhi(x[e]) + bfhi(y[e]));
    reinterpret_cast<u32x4*>(out)[i] = r;
  }
}

// ---------------------------------------------------------------------------
// smoothed cross entropy + gradient, one workgroup per row
// ---------------------------------------------------------------------------
constexpr int kXentMaxPerThread = 20;   // 16-B vectors per thread -> V <= 256*8*20 = 40960

__global__ __launch_bounds__(256) void xent_smooth_kernel(
    const bf16_t* __restrict__ logits, const int32_t* __restrict__ labels, int V, int v_valid,
    long long ld, float confidence, float low_confidence, float normalizing, float grad_scale_host,
    const float* __restrict__ grad_scale_dev, float* __restrict__ row_loss,
    bf16_t* __restrict__ dlogits) {
  __shared__ float red[4];
  __shared__ float bc;
  const long long row = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int V8 = V >> 3;
  const bf16_t* lr = logits + row * ld;
  const int label = labels[row];
  if (label < 0) {   // masked position (sequence_mask weight 0): no loss, no gradient
    if (threadIdx.x == 0 && row_loss) row_loss[row] = 0.f;
    if (dlogits) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      for (int i = threadIdx.x; i < V8; i += 256)
        *reinterpret_cast<u32x4*>(dlogits + row * ld + (long long)i * 8) = z;
    }
    return;
  }
  u32x4 buf[kXentMaxPerThread];
  float mx = -INFINITY, sx = 0.f;
  // fully unrolled with a compile-time bound: the row stays in registers (a runtime-indexed
  // array would be demoted to scratch)
#pragma unroll
  for (int k = 0; k < kXentMaxPerThread; ++k) {
    const int i = threadIdx.x + k * 256;
    u32x4 t = {0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};   // -inf pairs
    if (i < V8) {
      t = *reinterpret_cast<const u32x4*>(lr + (long long)i * 8);
      if (i * 8 + 8 > v_valid) {   // vocabulary padding columns: -inf logits, no probability mass
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = i * 8 + 2 * e;
          if (c >= v_valid) t[e] = (t[e] & 0xffff0000u) | 0xff80u;
          if (c + 1 >= v_valid) t[e] = (t[e] & 0x0000ffffu) | 0xff800000u;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bflo(t[e]), b = bfhi(t[e]);
        mx = fmaxf(mx, fmaxf(a, b));
        sx += (a > -INFINITY ? a : 0.f) + (b > -INFINITY ? b : 0.f);
      }
    }
    buf[k] = t;
  }
  auto block_reduce = [&](float v, bool is_max) -> float {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0)
      bc = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))
                  : (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return bc;
  };
  mx = block_reduce(mx, true);
  sx = block_reduce(sx, false);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < kXentMaxPerThread; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) se += __expf(bflo(buf[k][e]) - mx) + __expf(bfhi(buf[k][e]) - mx);
  se = block_reduce(se, false);
  const float lse = mx + __logf(se);
  const float x_label = bf2f(lr[label]);
  // xent = -sum_v t_v logp_v,  t = confidence at label, low elsewhere
  const float sum_logp = sx - (float)v_valid * lse;
  const float logp_label = x_label - lse;
  const float xent = -(confidence * logp_label + low_confidence * (sum_logp - logp_label)) - normalizing;
  if (threadIdx.x == 0 && row_loss) row_loss[row] = xent;
  if (dlogits) {
    const float gs = grad_scale_dev ? grad_scale_host * (*grad_scale_dev) : grad_scale_host;
    bf16_t* dr = dlogits + row * ld;
#pragma unroll
    for (int k = 0; k < kXentMaxPerThread; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i >= V8) continue;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = i * 8 + 2 * e;
        const float p0 = __expf(bflo(buf[k][e]) - lse), p1 = __expf(bfhi(buf[k][e]) - lse);
        const float t0 = (c == label) ? confidence : (c < v_valid ? low_confidence : 0.f);
        const float t1 = (c + 1 == label) ? confidence : (c + 1 < v_valid ? low_confidence : 0.f);
        o[e] = pack2bf((p0 - t0) * gs, (p1 - t1) * gs);
      }
      *reinterpret_cast<u32x4*>(dr + (long long)i * 8) = o;
    }
  }
}

// argmax over the first v_valid columns of each bf16 row (first maximum wins, as tf.argmax)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const bf16_t* __restrict__ x, long long N,
                                                          int v_valid, long long ld,
                                                          int32_t* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < v_valid; c += 64) {
    const float v = bf2f(x[row * ld + c]);
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

__global__ void sum_rows_kernel(const float* __restrict__ v, long long n, float mul,
                                float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)(red[0] * (double)mul);
}

static inline int ew_blocks(long long work) {
  long long b = (work + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_embed_fwd(os2s_stream_t stream, const int32_t* ids, const int32_t* pos,
                              const uint16_t* table, int V, int D, long long N, float emb_scale,
                              float keep_prob, unsigned long long seed, uint16_t* out,
                              int plain_lookup) {
  OS2S_REQUIRE(ids && table && out && V >= 1 && D >= 16 && D % 16 == 0 && N >= 0);
  OS2S_REQUIRE(pos || plain_lookup);
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  if (N == 0) return OS2S_OK;
  OS2S_LAUNCH(embed_fwd_kernel, dim3(ew_blocks(N * (D / 8))), dim3(256), 0, (hipStream_t)stream,
              ids, pos, table, V, D, N, emb_scale, keep_prob, seed, out, plain_lookup);
  return OS2S_OK;
}

extern "C" int os2s_embed_bwd(os2s_stream_t stream, const int32_t* ids, const uint16_t* dout, int V,
                              int D, long long N, float emb_scale, float keep_prob,
                              unsigned long long seed, float* dtable,
                              int plain_lookup) {
  OS2S_REQUIRE(ids && dout && dtable && D % 8 == 0 && N >= 0);
  if (N == 0) return OS2S_OK;
  if (os2s_deterministic()) {
    OS2S_LAUNCH(embed_bwd_det_kernel, dim3(ceil_div(D / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                ids, dout, V, D, N, emb_scale, keep_prob, seed, dtable, plain_lookup);
    return OS2S_OK;
  }
  OS2S_LAUNCH(embed_bwd_kernel, dim3(ew_blocks(N * (D / 8))), dim3(256), 0, (hipStream_t)stream,
              ids, dout, V, D, N, emb_scale, keep_prob, seed, dtable, plain_lookup);
  return OS2S_OK;
}

extern "C" int os2s_layernorm_fwd(os2s_stream_t stream, const uint16_t* x, const float* gamma,
                                  const float* beta, float eps, long long N, int D, uint16_t* y,
                                  float* mean, float* rstd) {
  OS2S_REQUIRE(x && gamma && beta && y && N >= 0);
  if (N == 0) return OS2S_OK;
  dim3 grid(ceil_div(N, 4));
  if (D == 1024) {
    OS2S_LAUNCH(layernorm_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps,
                N, y, mean, rstd);
  } else if (D == 512) {
    OS2S_LAUNCH(layernorm_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps,
                N, y, mean, rstd);
  } else {
    return OS2S_ERR_UNSUPPORTED;
  }
  return OS2S_OK;
}

// rows per workgroup of the LayerNorm backward (8 / 16 / 32 measured equal on a Transformer-big step:
// the step is bound by the sum of kernel times, not by this kernel's latency)
static const int kLnRowsPerBlock = 32;
extern "C" int os2s_layernorm_bwd_num_parts(long long N) { return ceil_div(N, kLnRowsPerBlock); }

// partial: [num_parts, 2, D] (row 0: sum dy = dbeta, row 1: sum dy*xhat = dgamma); reduce it
// with os2s_bn_bwd_finalize(partial, nparts, nq=2, q=1, C=D, ...).
extern "C" int os2s_layernorm_bwd(os2s_stream_t stream, const uint16_t* dy, const uint16_t* x,
                                  const float* gamma, const float* mean, const float* rstd,
                                  const uint16_t* dres, long long N, int D, uint16_t* dx,
                                  float* partial) {
  OS2S_REQUIRE(dy && x && gamma && mean && rstd && dx && partial && N >= 0);
  if (N == 0) return OS2S_OK;
  dim3 grid(ceil_div(N, kLnRowsPerBlock));
  if (D == 1024) {
    OS2S_LAUNCH(layernorm_bwd_kernel<2>, grid, dim3(64 * kLnWaves), 0, (hipStream_t)stream, dy, x, gamma, mean,
                rstd, dres, N, kLnRowsPerBlock, dx, partial);
  } else if (D == 512) {
    OS2S_LAUNCH(layernorm_bwd_kernel<1>, grid, dim3(64 * kLnWaves), 0, (hipStream_t)stream, dy, x, gamma, mean,
                rstd, dres, N, kLnRowsPerBlock, dx, partial);
  } else {
    return OS2S_ERR_UNSUPPORTED;
  }
  return OS2S_OK;
}

extern "C" int os2s_dropout_bwd(os2s_stream_t stream, const uint16_t* dout, const uint16_t* out,
                                int mode, float keep_prob, unsigned long long seed, long long n,
                                uint16_t* d) {
  OS2S_REQUIRE(dout && d && n >= 0 && n % 8 == 0 && (mode == 0 || ((mode == 1 || mode == 2) && out)));
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(dropout_bwd_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, dout,
              out, mode, keep_prob, seed, n / 8, d);
  return OS2S_OK;
}

extern "C" int os2s_dropout_bwd_colsum_num_parts(long long rows) {
  return (int)((rows + os2s::kDropColRows - 1) / os2s::kDropColRows);
}

extern "C" int os2s_dropout_bwd_colsum(os2s_stream_t stream, const uint16_t* dout, const uint16_t* out,
                                       int mode, float keep_prob, unsigned long long seed, long long rows,
                                       int C, uint16_t* d, float* partial) {
  OS2S_REQUIRE(dout && d && partial && rows >= 1 && C >= 8 && C % 8 == 0 && (mode == 0 || ((mode == 1 || mode == 2) && out)));
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  int G = C / 8 < 256 ? C / 8 : 256;
  while (256 % G) --G;                      // G divides 256: whole rows of threads
  dim3 grid((unsigned)os2s_dropout_bwd_colsum_num_parts(rows), (unsigned)ceil_div(C / 8, G));
  OS2S_LAUNCH(dropout_bwd_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, dout, out, mode, keep_prob,
              seed, rows, C, G, d, partial);
  return OS2S_OK;
}

extern "C" int os2s_dense_epilogue(os2s_stream_t stream, uint16_t* y, const float* bias, long long rows,
                                   int C, int act, float keep_prob, unsigned long long seed,
                                   const uint16_t* residual) {
  OS2S_REQUIRE(y && rows >= 0 && C >= 8 && C % 8 == 0 && (act == 0 || act == 1));
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  if (rows == 0) return OS2S_OK;
  const long long n8 = rows * (C / 8);
  OS2S_LAUNCH(dense_epilogue_kernel, dim3(ew_blocks(n8)), dim3(256), 0, (hipStream_t)stream, y, bias, C,
              act, keep_prob, seed, residual, n8);
  return OS2S_OK;
}

extern "C" int os2s_add_bf16(os2s_stream_t stream, const uint16_t* a, const uint16_t* b, long long n,
                             uint16_t* out) {
  OS2S_REQUIRE(a && b && out && n >= 0 && n % 8 == 0);
  if (n == 0) return OS2S_OK;
  OS2S_LAUNCH(add_bf16_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, n / 8,
              out);
  return OS2S_OK;
}

extern "C" int os2s_argmax_rows(os2s_stream_t stream, const uint16_t* x, long long N, int V_valid,
                                long long ld, int32_t* out) {
  OS2S_REQUIRE(x && out && N >= 0 && V_valid >= 1 && ld >= V_valid);
  if (N == 0) return OS2S_OK;
  OS2S_LAUNCH(argmax_rows_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, x, N,
              V_valid, ld, out);
  return OS2S_OK;
}

extern "C" int os2s_xent_smooth(os2s_stream_t stream, const uint16_t* logits, const int32_t* labels,
                                long long N, int V, int V_valid, long long ld,
                                float label_smoothing, float grad_scale,
                                const float* grad_scale_dev, float* row_loss, float* loss_mean,
                                uint16_t* dlogits) {
  OS2S_REQUIRE(logits && labels && N >= 1 && V >= 8 && V % 8 == 0 && ld % 8 == 0 && row_loss);
  OS2S_REQUIRE(V_valid >= 2 && V_valid <= V);
  if (V > 256 * 8 * kXentMaxPerThread) return OS2S_ERR_UNSUPPORTED;
  const float confidence = 1.f - label_smoothing;
  const float low = (1.f - confidence) / (float)(V_valid - 1);
  const float normalizing =
      -(confidence * logf(confidence) + (float)(V_valid - 1) * low * logf(low + 1e-20f));
  OS2S_LAUNCH(xent_smooth_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, logits, labels,
              V, V_valid, ld, confidence, low, normalizing, grad_scale, grad_scale_dev, row_loss, dlogits);
  if (loss_mean) {
    OS2S_LAUNCH(sum_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, N,
                grad_scale, loss_mean);
  }
  return OS2S_OK;
}
