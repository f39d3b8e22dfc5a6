// BatchNorm (+ residual sum + activation + dropout + sequence mask) for the
// conv blocks, channels-last, bf16 I/O with fp32 statistics. HBM-bound kernels:
// 16-byte vector loads/stores, one pass per tensor, wavefront/LDS reductions.
//
// Reference semantics (open_seq2seq/parts/cnns/conv_blocks.py):
//   conv_bn_actv            :170-232   y = act(BN(conv))
//   conv_bn_res_bn_actv     :61-168    y = act(BN(conv) + sum_i BN_i(conv1x1_i(res_i)))
//   followed by tf.nn.dropout (encoders/tdnn_encoder.py:255) and the sequence
//   mask the encoder multiplies onto the NEXT conv's input (:185-186,204-205),
//   which we fold into the producer's store.
// tf.layers.batch_normalization on a 4-D tensor uses the fused kernel
// (that is why the reference expands dims, conv_blocks.py:208-214):
//   training: normalise with the biased batch variance; moving statistics are
//   updated as moving = moving*momentum + batch*(1-momentum) where the batch
//   variance fed to the moving average carries Bessel's correction n/(n-1)
//   [TF fused_batch_norm convention — not in /root/reference];
//   statistics run over ALL B*T positions, padded frames included (Appendix B.2).
#include "os2s_common.hpp"

namespace os2s {

constexpr int kMaxBnInputs = 12;

// ---------------------------------------------------------------------------
// finalize: partial (sum, sumsq) -> mean / rstd / fused scale+shift, moving stats
// ---------------------------------------------------------------------------
constexpr int kFinLanes = 16;     // partial-row lanes per channel (x 64 channels = 1024 threads)

__device__ __forceinline__ void bn_finalize_body(
    const float* __restrict__ partial, int nparts, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, int training, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, float* __restrict__ scale_out,
    float* __restrict__ shift_out) {
  // 64 channels per block, kFinLanes lanes of partial rows per channel (coalesced over c): the
  // kernel is a latency chain over nparts / kFinLanes dependent-free loads, launched ~100x / step
  __shared__ double sh_s[kFinLanes][64], sh_q[kFinLanes][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, q = 0.0;
  if (training && c < C) {
    int i = pl;
    for (; i + 3 * kFinLanes < nparts; i += 4 * kFinLanes) {
      const float a0 = partial[((long long)i * 2) * C + c], b0 = partial[((long long)i * 2 + 1) * C + c];
      const float a1 = partial[((long long)(i + kFinLanes) * 2) * C + c], b1 = partial[((long long)(i + kFinLanes) * 2 + 1) * C + c];
      const float a2 = partial[((long long)(i + 2 * kFinLanes) * 2) * C + c], b2 = partial[((long long)(i + 2 * kFinLanes) * 2 + 1) * C + c];
      const float a3 = partial[((long long)(i + 3 * kFinLanes) * 2) * C + c], b3 = partial[((long long)(i + 3 * kFinLanes) * 2 + 1) * C + c];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      q += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; i < nparts; i += kFinLanes) {
      s += (double)partial[((long long)i * 2) * C + c];
      q += (double)partial[((long long)i * 2 + 1) * C + c];
    }
  }
  sh_s[pl][cl] = s;
  sh_q[pl][cl] = q;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  float mean, var;
  if (training) {
    s = 0.0; q = 0.0;
#pragma unroll
    for (int l = 0; l < kFinLanes; ++l) { s += sh_s[l][cl]; q += sh_q[l][cl]; }
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (moving_mean) {
      const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
      moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
      moving_var[c] = moving_var[c] * momentum + (float)unbiased * (1.f - momentum);
    }
  } else {
    mean = moving_mean[c];
    var = moving_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  if (mean_out) mean_out[c] = mean;
  if (rstd_out) rstd_out[c] = rstd;
  scale_out[c] = g * rstd;
  shift_out[c] = b - mean * g * rstd;
}

__global__ __launch_bounds__(64 * kFinLanes) void bn_finalize_kernel(
    const float* __restrict__ partial, int nparts, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, int training, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, float* __restrict__ scale_out,
    float* __restrict__ shift_out) {
  bn_finalize_body(partial, nparts, C, count, gamma, beta, eps, momentum, training, moving_mean, moving_var,
                   mean_out, rstd_out, scale_out, shift_out);
}

// The same for up to 16 BatchNorms of one geometry (nparts, C, count) in ONE launch (blockIdx.y = the
// BatchNorm): the 1x1 residual branches of a block end (up to 11 finalize launches of ~5.6 us each).
constexpr int kBnFinMaxJ = 16;
struct BnFinFwdMulti {
  const float* partial[kBnFinMaxJ]; const float* gamma[kBnFinMaxJ]; const float* beta[kBnFinMaxJ];
  float* moving_mean[kBnFinMaxJ]; float* moving_var[kBnFinMaxJ]; float* mean_out[kBnFinMaxJ];
  float* rstd_out[kBnFinMaxJ]; float* scale_out[kBnFinMaxJ]; float* shift_out[kBnFinMaxJ];
};
__global__ __launch_bounds__(64 * kFinLanes) void bn_finalize_fwd_multi_kernel(
    BnFinFwdMulti t, int nparts, int C, double count, float eps, float momentum, int training) {
  const int j = blockIdx.y;
  bn_finalize_body(t.partial[j], nparts, C, count, t.gamma[j], t.beta[j], eps, momentum, training,
                   t.moving_mean[j], t.moving_var[j], t.mean_out[j], t.rstd_out[j], t.scale_out[j], t.shift_out[j]);
}

// ---------------------------------------------------------------------------
// Stand-alone per-channel (sum, sumsq) partials of a [rows, C] bf16 tensor, for
// producers that do not emit them (same [nparts, 2, C] layout as the conv).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ y,
                                                       long long rows, int C,
                                                       int rows_per_block,
                                                       float* __restrict__ partial) {
  // thread -> 8-channel group g = tid % G, row lane rl = tid / G
  __shared__ float red[256 * 16];
  const int G = min(C / 8, 256);
  const int RL = 256 / G;
  const int cg0 = blockIdx.y * G;
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  const int c0 = (cg0 + g) * 8;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  if (c0 < C && rl < RL) {
    // four rows of loads in flight per thread (round 5: one load per iteration ran at 0.8 - 1 TB/s on the
    // [rows, 2400] bias-gradient sums of DeepSpeech2, 32 launches of 54 us per step)
    long long r = r0 + rl;
    for (; r + 3LL * RL < r1; r += 4LL * RL) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(y + (r + (long long)u * RL) * C + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bflo(v[u][e]), b = bfhi(v[u][e]);
          s[2 * e] += a; q[2 * e] += a * a;
          s[2 * e + 1] += b; q[2 * e + 1] += b * b;
        }
    }
    for (; r < r1; r += RL) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(y + r * C + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bflo(v[e]), b = bfhi(v[e]);
        s[2 * e] += a; q[2 * e] += a * a;
        s[2 * e + 1] += b; q[2 * e + 1] += b * b;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x * 16 + e] = s[e];
    red[threadIdx.x * 16 + 8 + e] = q[e];
  }
  __syncthreads();
  if (rl == 0 && c0 < C) {
    for (int e = 0; e < 8; ++e) {
      float ss = 0.f, qq = 0.f;
      for (int k = 0; k < RL; ++k) {
        ss += red[(k * G + g) * 16 + e];
        qq += red[(k * G + g) * 16 + 8 + e];
      }
      partial[((long long)blockIdx.x * 2 + 0) * C + c0 + e] = ss;
      partial[((long long)blockIdx.x * 2 + 1) * C + c0 + e] = qq;
    }
  }
}

// ---------------------------------------------------------------------------
// Forward apply: out = mask * dropout( act( sum_j y_j*scale_j + shift_j ) )
// ---------------------------------------------------------------------------
struct BnActArgs {
  const bf16_t* y[kMaxBnInputs];
  const float* scale[kMaxBnInputs];
  const float* shift[kMaxBnInputs];
  int J;
  bf16_t* out;
  const int32_t* out_len;  // [B] or null: rows t >= out_len[b] are zeroed
  int B, T, C;
  int act;                 // 0 none, 1 relu, 2 tanh, 3 relu capped at 20 (min(relu(x), 20): the clipped
                           // ReLU of the reference's DeepSpeech2 / Wave2Letter configs)
  float keep_prob;         // 1.0 = no dropout
  unsigned long long seed;
};

constexpr float kReluCap = 20.f;
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return tanhf(v);
  if (act == 3) return fminf(fmaxf(v, 0.f), kReluCap);
  return v;
}

// Thread layout of the row-walking kernels below: a workgroup covers trows consecutive rows of
// the flattened [B*T, C] tensor x G 8-channel groups (g_tiling below: 32 groups = 512 contiguous
// bytes per row and 32 / 64 rows per workgroup measured best; the kernels are bound by how many
// bytes are in flight, not by the segment width); thread -> group g = tid % G, fixed for
// the thread's whole walk so the per-channel constants live in registers, and row lane rl = tid / G.
//
// Round 4, the two backward kernels (bn_act_bwd_reduce, bn_bwd_apply): no branch stands around a
// memory instruction. The first version tracked (b, t) of a row with an incremental cursor and
// loaded `if (live) ...`; hipcc compiled that into one exec-masked basic block per row, each ending
// in s_waitcnt vmcnt(0) — a thread had two 16-byte loads in flight, not eight. Now (1) the workgroup
// classifies its rows once (one division per ROW, by the first trows threads, into LDS: BlockRows),
// (2) every tensor is read and written through a buffer descriptor that covers exactly the
// workgroup's rows, with the byte offset of a row that must not be touched replaced by an
// out-of-range one: such a load returns zeros without a memory request and such a store is
// dropped, (3) all loads of a thread's rows are issued back to back before the first use.
// tools/bench_bn_sweep.py, Jasper shapes, same box: apply 18.7 / 27.2 / 31.6 -> 17.8 / 24.1 / 29.5 us
// (512 / 768 / 1024 channels), reduce 28.4 / 38.0 / 45.5 -> 28.5 / 35.8 / 40.8 — 5-10 %, not the 2x
// the instruction stream suggested: at ~4 TB/s of mixed read / write streams over ~40 MB tensors
// these 20-40 us kernels are bound by the memory system, not by loads in flight per thread.
constexpr int kTileU = 4;
constexpr int kMaxTileRows = 1024;   // the tiling options' upper bound on rows per workgroup
constexpr int kOobOffset = 0x7fffffff;

struct TileMap {
  int G, RL, g, rl, cg;
  bool cvalid;
  __device__ __forceinline__ TileMap(int C8, int gmax) {
    G = min(C8, gmax);
    RL = 256 / G;
    g = threadIdx.x % G;
    rl = threadIdx.x / G;
    cg = blockIdx.y * G + g;
    cvalid = cg < C8 && rl < RL;
  }
};
static inline int tile_cblocks(int C8, int gmax) { return ceil_div(C8, C8 < gmax ? C8 : gmax); }

// {channel groups per workgroup, rows per workgroup} of bn_act_fwd / bn_act_bwd_reduce / bn_bwd_apply
// defaults from tools/bench_bn_sweep.py on MI355X (Jasper shapes, working set rotated out of the MALL)
static int g_tiling[3][2] = {{32, 32}, {32, 64}, {32, 64}};

// Row classes of a workgroup's block of rows [r0, r0 + n): bit 0 = t < len[b] + margin, bit 1 = t < len[b]
// (len = T without a length vector). Ends with a barrier: call it before any thread leaves.
struct BlockRows {
  long long r0;
  int n;
  __device__ __forceinline__ void init(unsigned char* flag, long long rows, int trows, int T,
                                       const int32_t* lens, int margin) {
    r0 = (long long)blockIdx.x * trows;
    const long long left = rows - r0;
    n = (int)(left < trows ? left : trows);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const long long r = r0 + i;
      const int b = (int)((unsigned long long)r / (unsigned)T);
      const int t = (int)(r - (long long)b * T);
      const int len = lens ? lens[b] : T;
      flag[i] = (unsigned char)((t < len + margin ? 1 : 0) | (t < len ? 2 : 0));
    }
    __syncthreads();
  }
};

// Descriptor over the n rows of C bf16 channels that start at row r0 of a [rows, C] tensor
__device__ __forceinline__ __amdgpu_buffer_rsrc_t block_rsrc(const void* base, long long r0, int n, int C) {
  const unsigned long long a = (unsigned long long)base + (unsigned long long)r0 * (unsigned)C * 2ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int bytes = __builtin_amdgcn_readfirstlane(n * C * 2);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 row_load(__amdgpu_buffer_rsrc_t rs, int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rs, int off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
}

__device__ __forceinline__ void unpack8(const u32x4& v, float (&o)[8]) {
  o[0] = bflo(v[0]); o[1] = bfhi(v[0]); o[2] = bflo(v[1]); o[3] = bfhi(v[1]);
  o[4] = bflo(v[2]); o[5] = bfhi(v[2]); o[6] = bflo(v[3]); o[7] = bfhi(v[3]);
}

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  u32x4 o;
  o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
  o[2] = pack2bf(v[4], v[5]); o[3] = pack2bf(v[6], v[7]);
  return o;
}

// The forward apply keeps the incremental (b, t) cursor: one tensor in, one out, four loads in flight
// were enough (the BlockRows version measured the same at 64 rows and slower at 32).
struct RowCursor {
  long long row;
  int b, t, T, len;
  const int32_t* lens;
  __device__ __forceinline__ void init(long long r, int T_, const int32_t* lens_) {
    row = r; T = T_; lens = lens_;
    b = (int)((unsigned long long)r / (unsigned)T_);
    t = (int)(r - (long long)b * T_);
    len = lens_ ? lens_[b] : T_;
  }
  __device__ __forceinline__ bool live() const { return t < len; }
  // step; only valid while row stays < B*T (callers check the row bound first)
  __device__ __forceinline__ void advance(int n, long long rows) {
    row += n; t += n;
    if (t >= T && row < rows) {
      do { t -= T; ++b; } while (t >= T);
      len = lens ? lens[b] : T;
    }
  }
};

template <bool SINGLE>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnActArgs p, int gmax, int trows) {
  const int C8 = p.C >> 3;
  const TileMap tm(C8, gmax);
  if (!tm.cvalid) return;
  const int rl = tm.rl, cg = tm.cg, RL = tm.RL;
  const int c0 = cg * 8;
  const long long rows = (long long)p.B * p.T;
  const long long r0 = (long long)blockIdx.x * trows;
  const long long r1 = min(rows, r0 + trows);
  const float inv_keep = 1.f / p.keep_prob;
  float sc[8], sh[8];
  if (SINGLE) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale[0] + c0);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.scale[0] + c0 + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.shift[0] + c0);
    const f32x4 h1 = *reinterpret_cast<const f32x4*>(p.shift[0] + c0 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
  }
  RowCursor cur;
  if (r0 + rl < r1) cur.init(r0 + rl, p.T, p.out_len);
  for (long long base = r0 + rl; base < r1; base += (long long)kTileU * RL) {
    long long rw[kTileU];
    bool ok[kTileU], lv[kTileU];
    float v[kTileU][8];
#pragma unroll
    for (int u = 0; u < kTileU; ++u) {
      rw[u] = base + (long long)u * RL;
      ok[u] = rw[u] < r1;
      lv[u] = ok[u] && cur.live();
      if (ok[u]) cur.advance(RL, rows);
    }
    if (SINGLE) {
      u32x4 y[kTileU];
#pragma unroll
      for (int u = 0; u < kTileU; ++u)
        if (lv[u]) y[u] = *reinterpret_cast<const u32x4*>(p.y[0] + rw[u] * p.C + c0);
#pragma unroll
      for (int u = 0; u < kTileU; ++u)
        if (lv[u]) {
          float yv[8];
          unpack8(y[u], yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[u][e] = yv[e] * sc[e] + sh[e];
        }
    } else {
#pragma unroll
      for (int u = 0; u < kTileU; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
      for (int j = 0; j < p.J; ++j) {
        u32x4 y[kTileU];
#pragma unroll
        for (int u = 0; u < kTileU; ++u)
          if (lv[u]) y[u] = *reinterpret_cast<const u32x4*>(p.y[j] + rw[u] * p.C + c0);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale[j] + c0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.scale[j] + c0 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.shift[j] + c0);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(p.shift[j] + c0 + 4);
#pragma unroll
        for (int u = 0; u < kTileU; ++u)
          if (lv[u]) {
            float yv[8];
            unpack8(y[u], yv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[u][e] += yv[e] * s0[e] + h0[e];
              v[u][4 + e] += yv[4 + e] * s1[e] + h1[e];
            }
          }
      }
    }
#pragma unroll
    for (int u = 0; u < kTileU; ++u) {
      if (!ok[u]) continue;
      u32x4 o = {0u, 0u, 0u, 0u};
      if (lv[u]) {
        uint32_t keep = 0xffu;
        if (p.keep_prob < 1.f)
          keep = dropout_bits8(p.seed, (unsigned long long)(rw[u] * C8 + cg), p.keep_prob);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = apply_act(v[u][e], p.act);
          if (p.keep_prob < 1.f) a = ((keep >> e) & 1u) ? a * inv_keep : 0.f;
          v[u][e] = a;
        }
        o = pack8(v[u]);
      }
      *reinterpret_cast<u32x4*>(p.out + rw[u] * p.C + c0) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// Backward pass 1: dz = dA * mask * dropout' * act'(out), written as bf16, plus
// per-channel partial sums  sum(dz)  and  sum(dz * xhat_j)  for every input j.
// ---------------------------------------------------------------------------
struct BnBwdReduceArgs {
  const bf16_t* dout;      // [B,T,C] grad wrt the block output
  const bf16_t* out;       // saved block output (for act')
  const bf16_t* y[kMaxBnInputs];
  const float* mean[kMaxBnInputs];
  const float* rstd[kMaxBnInputs];
  int J;
  bf16_t* dz;              // [B,T,C]
  float* partial;          // [nblocks_rows][1+J][C]
  const int32_t* out_len;
  int B, T, C, act;
  float keep_prob;
  unsigned long long seed;
  int rows_per_block;
};

// U = rows in flight per thread: 4 for the single-input instantiation, fewer for the residual block
// ends (their per-input accumulators already take 32 / 96 registers; with U = 4 the 12-input
// instantiation ran at one wave per SIMD). The inputs are read JB at a time so that every
// instantiation has 4 loads of y in flight, the first JB together with dout / out.
template <int J_MAX, int kTileU>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(BnBwdReduceArgs p, int gmax) {
  constexpr int U = kTileU;
  constexpr int JB = 4 / U;
  static_assert(JB >= 1 && J_MAX % JB == 0, "inputs are read in whole groups");
  __shared__ float red[256 * 8];
  __shared__ unsigned char flag[kMaxTileRows];
  const int C8 = p.C >> 3;
  const int trows = p.rows_per_block;
  const TileMap tm(C8, gmax);
  const int g = tm.g, rl = tm.rl, cg = tm.cg, RL = tm.RL, G = tm.G;
  const bool cvalid = tm.cvalid;
  const int c0 = cg * 8;
  const int rowb = p.C * 2;
  const float inv_keep = 1.f / p.keep_prob;
  // a capped output as stored: bf16(20 / keep) (values at or above it took no gradient)
  const float cap_out = bflo(pack2bf(kReluCap * inv_keep, 0.f));
  float sd[8];
  float sx[J_MAX][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sd[e] = 0.f;
#pragma unroll
  for (int j = 0; j < J_MAX; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) sx[j][e] = 0.f;
  BlockRows br;
  br.init(flag, (long long)p.B * p.T, trows, p.T, p.out_len, 0);
  if (cvalid) {
    const __amdgpu_buffer_rsrc_t dors = block_rsrc(p.dout, br.r0, br.n, p.C);
    const __amdgpu_buffer_rsrc_t outrs = block_rsrc(p.out, br.r0, br.n, p.C);
    const __amdgpu_buffer_rsrc_t dzrs = block_rsrc(p.dz, br.r0, br.n, p.C);
    for (int i0 = rl; i0 < br.n; i0 += U * RL) {
      int off[U], ld[U];
      bool lv[U];
      u32x4 d[U], o[U], y[JB][U];
      float dr[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * RL;
        const bool in = i < br.n;
        lv[u] = in && (flag[in ? i : 0] & 2);
        off[u] = in ? i * rowb + c0 * 2 : kOobOffset;
        ld[u] = lv[u] ? off[u] : kOobOffset;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        d[u] = row_load(dors, ld[u]);
        o[u] = row_load(outrs, ld[u]);
      }
      auto load_inputs = [&](int jc) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          // inputs past p.J: whatever the argument block holds there is never dereferenced
          const __amdgpu_buffer_rsrc_t yrs = block_rsrc(p.y[jc + jj], br.r0, br.n, p.C);
#pragma unroll
          for (int u = 0; u < U; ++u) y[jj][u] = row_load(yrs, jc + jj < p.J ? ld[u] : kOobOffset);
        }
      };
      auto sum_inputs = [&](int jc) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float yv[8];
            unpack8(y[jj][u], yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) sx[jc + jj][e] += dr[u][e] * yv[e];
          }
      };
      load_inputs(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float dv[8], ov[8], dzv[8];
        unpack8(d[u], dv);
        unpack8(o[u], ov);
        uint32_t keep = 0xffu;
        if (p.keep_prob < 1.f)
          keep = dropout_bits8(p.seed, (unsigned long long)((br.r0 + i0 + u * RL) * C8 + cg), p.keep_prob);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gsc = 1.f;
          if (p.keep_prob < 1.f) gsc = ((keep >> e) & 1u) ? inv_keep : 0.f;
          float dact = 1.f;
          if (p.act == 1) dact = ov[e] > 0.f ? 1.f : 0.f;
          else if (p.act == 3) dact = (ov[e] > 0.f && ov[e] < cap_out) ? 1.f : 0.f;
          else if (p.act == 2) {
            const float th = ov[e] * p.keep_prob;  // tanh(z) of a kept element
            dact = 1.f - th * th;
          }
          dzv[e] = dv[e] * gsc * dact;
        }
        // rows at or past out_len[b] read as zeros above, so dz is zero there
        const u32x4 w = pack8(dzv);
        row_store(dzrs, off[u], w);
        // the sums use the bf16-rounded dz, i.e. exactly what pass 2 re-reads
        unpack8(w, dr[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) sd[e] += dr[u][e];
      }
      // only sum(dz * y) is accumulated per row: sum(dz * xhat) = rstd * (sum(dz * y) -
      // mean * sum(dz)) is formed once per thread below (no per-row mean / rstd loads)
      sum_inputs(0);
#pragma unroll
      for (int jc = JB; jc < J_MAX; jc += JB)
        if (jc < p.J) {
          load_inputs(jc);
          sum_inputs(jc);
        }
    }
  }
  // block reduction over the row lanes, one quantity (dz sum, then each input's dz*xhat sum)
  // at a time through an 8 KB LDS buffer: the footprint does not grow with the number of
  // residual inputs, so the 12-input instantiation keeps its occupancy
  const bool writer = rl == 0 && cg < C8;
  auto reduce_q = [&](const float (&v)[8], int q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = v[e];
    __syncthreads();
    if (writer) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = 0.f;
        for (int k = 0; k < RL; ++k) s += red[(k * G + g) * 8 + e];
        p.partial[((long long)blockIdx.x * (1 + p.J) + q) * p.C + c0 + e] = s;
      }
    }
    __syncthreads();
  };
  reduce_q(sd, 0);
#pragma unroll
  for (int j = 0; j < J_MAX; ++j)
    if (j < p.J) {
      if (cvalid) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(p.mean[j] + c0);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(p.mean[j] + c0 + 4);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(p.rstd[j] + c0);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(p.rstd[j] + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sx[j][e] = q0[e] * (sx[j][e] - m0[e] * sd[e]);
          sx[j][4 + e] = q1[e] * (sx[j][4 + e] - m1[e] * sd[4 + e]);
        }
      }
      reduce_q(sx[j], 1 + j);
    }
}

// Backward finalize for input j: reduce the partials -> dgamma, dbeta and the two
// means pass 2 needs (c1 = mean(dz), c2 = mean(dz*xhat)).
struct BnFinMulti {
  float* dgamma[kMaxBnInputs];
  float* dbeta[kMaxBnInputs];
};

// Sum of rows q0 and q1 of the partials [nparts, nq, C] over the parts i = pl, pl + kFinLanes, ... for
// channel c, in fp64 and in a fixed order. Eight parts (16 loads) are in flight per thread and the
// tail is read with clamped indices and zeroed by a select: these kernels are a latency chain over
// nparts / kFinLanes loads (the two-at-a-time loop took 9 round trips for 263 parts, 11-13 us per
// launch, ~50 launches per step).
__device__ __forceinline__ void fin_sum_pair(const float* __restrict__ partial, int nparts, int nq, int q0,
                                             int q1, int C, int c, int pl, double& s0, double& s1) {
  const long long stride = (long long)nq * C;
  const float* const p0 = partial + (long long)q0 * C + c;
  const float* const p1 = partial + (long long)q1 * C + c;
  for (int i = pl; i < nparts; i += 8 * kFinLanes) {
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ik = i + k * kFinLanes;
      const long long o = (long long)(ik < nparts ? ik : nparts - 1) * stride;
      a[k] = p0[o];
      b[k] = p1[o];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i + k * kFinLanes >= nparts) { a[k] = 0.f; b[k] = 0.f; }
    s0 += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) +
          (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
    s1 += (((double)b[0] + (double)b[1]) + ((double)b[2] + (double)b[3])) +
          (((double)b[4] + (double)b[5]) + ((double)b[6] + (double)b[7]));
  }
}

// All inputs of a block end in ONE launch (blockIdx.y = input): c1 / c2 are [J, C].
__global__ __launch_bounds__(64 * kFinLanes) void bn_bwd_finalize_multi_kernel(
    const float* __restrict__ partial, int nparts, int nq, int C, double count, BnFinMulti ptrs,
    int accumulate, float* __restrict__ c1_all, float* __restrict__ c2_all);

__global__ __launch_bounds__(64 * kFinLanes) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nparts, int nq, int q, int C, double count,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double sh_d[kFinLanes][64], sh_x[kFinLanes][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double sd = 0.0, sx = 0.0;
  if (c < C) fin_sum_pair(partial, nparts, nq, 0, q, C, c, pl, sd, sx);
  sh_d[pl][cl] = sd;
  sh_x[pl][cl] = sx;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  sd = 0.0; sx = 0.0;
#pragma unroll
  for (int l = 0; l < kFinLanes; ++l) { sd += sh_d[l][cl]; sx += sh_x[l][cl]; }
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sd;
  c1[c] = (float)(sd / count);
  c2[c] = (float)(sx / count);
}

__global__ __launch_bounds__(64 * kFinLanes) void bn_bwd_finalize_multi_kernel(
    const float* __restrict__ partial, int nparts, int nq, int C, double count, BnFinMulti ptrs,
    int accumulate, float* __restrict__ c1_all, float* __restrict__ c2_all) {
  const int q = 1 + blockIdx.y;
  float* const dgamma = ptrs.dgamma[blockIdx.y];
  float* const dbeta = ptrs.dbeta[blockIdx.y];
  float* const c1 = c1_all + (long long)blockIdx.y * C;
  float* const c2 = c2_all + (long long)blockIdx.y * C;
  __shared__ double sh_d[kFinLanes][64], sh_x[kFinLanes][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double sd = 0.0, sx = 0.0;
  if (c < C) fin_sum_pair(partial, nparts, nq, 0, q, C, c, pl, sd, sx);
  sh_d[pl][cl] = sd;
  sh_x[pl][cl] = sx;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  sd = 0.0; sx = 0.0;
#pragma unroll
  for (int l = 0; l < kFinLanes; ++l) { sd += sh_d[l][cl]; sx += sh_x[l][cl]; }
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sd;
  c1[c] = (float)(sd / count);
  c2[c] = (float)(sx / count);
}


// Backward pass 2 for input j: dy = gamma*rstd*(dz - c1 - xhat*c2)
//   = A*dz + Bq*y + Cc with per-channel A, Bq, Cc held in registers; a thread owns
// 8 channels and 8 rows of the workgroup's block: 16 x 16-B loads in flight, all issued before the
// first use (64 rows x 32 groups per workgroup: the walk is one straight pass).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ c1,
    const float* __restrict__ c2, bf16_t* __restrict__ dy, const int32_t* __restrict__ out_len,
    int margin, int B, int T, int C, int gmax, int trows, int dz_to_len) {
  constexpr int U = 2 * kTileU;
  __shared__ unsigned char flag[kMaxTileRows];
  const int C8 = C >> 3;
  const TileMap tm(C8, gmax);
  BlockRows br;
  br.init(flag, (long long)B * T, trows, T, out_len, margin);
  if (!tm.cvalid) return;
  const int rl = tm.rl, cg = tm.cg, RL = tm.RL;
  const int c0 = cg * 8;
  const int rowb = C * 2;
  float A[8], Bq[8], Cc[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 rs4 = *reinterpret_cast<const f32x4*>(rstd + c0 + 4 * h);
    const f32x4 me4 = *reinterpret_cast<const f32x4*>(mean + c0 + 4 * h);
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(c1 + c0 + 4 * h);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(c2 + c0 + 4 * h);
    f32x4 gm4 = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm4 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float rs = rs4[e], gm = gm4[e];
      A[4 * h + e] = gm * rs;
      Bq[4 * h + e] = -gm * rs * rs * b4[e];
      Cc[4 * h + e] = gm * rs * (me4[e] * rs * b4[e] - a4[e]);
    }
  }
  const __amdgpu_buffer_rsrc_t dzrs = block_rsrc(dz, br.r0, br.n, C);
  const __amdgpu_buffer_rsrc_t yrs = block_rsrc(y, br.r0, br.n, C);
  const __amdgpu_buffer_rsrc_t dyrs = block_rsrc(dy, br.r0, br.n, C);
  // dz_to_len: dz is only DEFINED for rows before the sequence end (it was written by a data-
  // gradient launch that skips the rest, os2s_conv1d_dgrad_bnact_ws) and is zero beyond
  const unsigned dz_bit = dz_to_len ? 2u : 1u;
  for (int i0 = rl; i0 < br.n; i0 += U * RL) {
    int off[U];
    bool lv[U];
    u32x4 d[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RL;
      const bool in = i < br.n;
      const unsigned f = in ? flag[i] : 0u;
      lv[u] = (f & 1u) != 0u;
      off[u] = in ? i * rowb + c0 * 2 : kOobOffset;
      d[u] = row_load(dzrs, (f & dz_bit) ? off[u] : kOobOffset);
      yv[u] = row_load(yrs, lv[u] ? off[u] : kOobOffset);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // rows at or past out_len[b] + margin are written as zeros WITHOUT reading dz / y: the caller
      // states with `margin` how far past the sequence end its consumers look (the data- and
      // weight-gradient convolutions reach (K-1)*dilation rows into the padding; dy is NOT zero
      // there: -gamma*rstd*(c1 + xhat*c2) flows back through the batch statistics)
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = A[2 * e] * bflo(d[u][e]) + Bq[2 * e] * bflo(yv[u][e]) + Cc[2 * e];
        const float hi = A[2 * e + 1] * bfhi(d[u][e]) + Bq[2 * e + 1] * bfhi(yv[u][e]) + Cc[2 * e + 1];
        o[e] = pack2bf(lo, hi) & (lv[u] ? 0xffffffffu : 0u);   // a mask, not a branch
      }
      row_store(dyrs, off[u], o);
    }
  }
}

__global__ void dropout_mask_kernel(unsigned long long seed, long long n8, float keep,
                                    uint8_t* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = (uint8_t)dropout_bits8(seed, (unsigned long long)i, keep);
}

static inline int ew_grid(long long total_threads) {
  long long b = (total_threads + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_bn_finalize(os2s_stream_t stream, const float* partial, int nparts,
                                int C, long long count, const float* gamma,
                                const float* beta, float eps, float momentum,
                                int training, float* moving_mean, float* moving_var,
                                float* mean_out, float* rstd_out, float* scale_out,
                                float* shift_out) {
  OS2S_REQUIRE(C >= 1 && scale_out && shift_out);
  if (training) OS2S_REQUIRE(partial && nparts >= 1 && count >= 1);
  else OS2S_REQUIRE(moving_mean && moving_var);
  OS2S_LAUNCH(bn_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * kFinLanes), 0,
              (hipStream_t)stream, partial, nparts, C, (double)count, gamma, beta, eps,
              momentum, training, moving_mean, moving_var, mean_out, rstd_out, scale_out,
              shift_out);
  return OS2S_OK;
}

extern "C" int os2s_bn_finalize_multi(os2s_stream_t stream, int J, const float* const* partial, int nparts,
                                      int C, long long count, const float* const* gamma,
                                      const float* const* beta, float eps, float momentum, int training,
                                      float* const* moving_mean, float* const* moving_var,
                                      float* const* mean_out, float* const* rstd_out,
                                      float* const* scale_out, float* const* shift_out) {
  OS2S_REQUIRE(J >= 1 && J <= kBnFinMaxJ && C >= 1 && scale_out && shift_out);
  if (training) OS2S_REQUIRE(partial && nparts >= 1 && count >= 1);
  else OS2S_REQUIRE(moving_mean && moving_var);
  BnFinFwdMulti t = {};
  for (int j = 0; j < J; ++j) {
    OS2S_REQUIRE(scale_out[j] && shift_out[j]);
    if (training) OS2S_REQUIRE(partial[j]);
    else OS2S_REQUIRE(moving_mean[j] && moving_var[j]);
    t.partial[j] = partial ? partial[j] : nullptr;
    t.gamma[j] = gamma ? gamma[j] : nullptr;
    t.beta[j] = beta ? beta[j] : nullptr;
    t.moving_mean[j] = moving_mean ? moving_mean[j] : nullptr;
    t.moving_var[j] = moving_var ? moving_var[j] : nullptr;
    t.mean_out[j] = mean_out ? mean_out[j] : nullptr;
    t.rstd_out[j] = rstd_out ? rstd_out[j] : nullptr;
    t.scale_out[j] = scale_out[j];
    t.shift_out[j] = shift_out[j];
  }
  OS2S_LAUNCH(bn_finalize_fwd_multi_kernel, dim3(ceil_div(C, 64), J), dim3(64 * kFinLanes), 0,
              (hipStream_t)stream, t, nparts, C, (double)count, eps, momentum, training);
  return OS2S_OK;
}

static const int kStatsRowsPerBlock = 128;

extern "C" int os2s_bn_stats_num_parts(long long rows) {
  return ceil_div(rows, kStatsRowsPerBlock);
}

extern "C" int os2s_bn_stats(os2s_stream_t stream, const uint16_t* y, long long rows,
                             int C, float* partial) {
  OS2S_REQUIRE(y && partial && rows >= 1 && C >= 8 && C % 8 == 0);
  const int G = (C / 8) < 256 ? (C / 8) : 256;
  dim3 grid(ceil_div(rows, kStatsRowsPerBlock), ceil_div(C / 8, G));
  OS2S_LAUNCH(bn_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, rows, C,
              kStatsRowsPerBlock, partial);
  return OS2S_OK;
}

extern "C" int os2s_bn_act_fwd(os2s_stream_t stream, int J, const uint16_t* const* y,
                               const float* const* scale, const float* const* shift,
                               uint16_t* out, const int32_t* out_len, int B, int T,
                               int C, int act, float keep_prob,
                               unsigned long long seed) {
  OS2S_REQUIRE(J >= 1 && J <= kMaxBnInputs && out && C % 8 == 0 && C >= 8);
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f && act >= 0 && act <= 3);
  if ((long long)B * T == 0) return OS2S_OK;
  BnActArgs a;
  for (int j = 0; j < J; ++j) {
    OS2S_REQUIRE(y[j] && scale[j] && shift[j]);
    a.y[j] = y[j]; a.scale[j] = scale[j]; a.shift[j] = shift[j];
  }
  a.J = J; a.out = out; a.out_len = out_len; a.B = B; a.T = T; a.C = C; a.act = act;
  a.keep_prob = keep_prob; a.seed = seed;
  const int gmax = g_tiling[0][0], trows = g_tiling[0][1];
  dim3 grid(ceil_div((long long)B * T, trows), tile_cblocks(C / 8, gmax));
  if (J == 1) {
    OS2S_LAUNCH(bn_act_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, gmax, trows);
  } else {
    OS2S_LAUNCH(bn_act_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, gmax, trows);
  }
  return OS2S_OK;
}



// Benchmark options (os2s_set_option; tools/bench_bn_sweep.py): tiling of bn_act_fwd / bn_act_bwd_reduce /
// bn_bwd_apply — bn.<kernel>.groups (8-channel groups per block, 8 .. 256) and bn.<kernel>.rows (rows per
// block, 8 .. 1024; out-of-range values are ignored). Not thread-safe; the reduce tiling also sets what
// os2s_bn_act_bwd_num_parts returns.
static void bn_set_groups(int which, double v) { if (v >= 8 && v <= 256) g_tiling[which][0] = (int)v; }
static void bn_set_rows(int which, double v) { if (v >= 8 && v <= kMaxTileRows) g_tiling[which][1] = (int)v; }
static os2s::OptionReg r_bn0g("bn.act_fwd.groups", [](double v) { bn_set_groups(0, v); });
static os2s::OptionReg r_bn0r("bn.act_fwd.rows", [](double v) { bn_set_rows(0, v); });
static os2s::OptionReg r_bn1g("bn.act_bwd_reduce.groups", [](double v) { bn_set_groups(1, v); });
static os2s::OptionReg r_bn1r("bn.act_bwd_reduce.rows", [](double v) { bn_set_rows(1, v); });
static os2s::OptionReg r_bn2g("bn.bwd_apply.groups", [](double v) { bn_set_groups(2, v); });
static os2s::OptionReg r_bn2r("bn.bwd_apply.rows", [](double v) { bn_set_rows(2, v); });

extern "C" int os2s_bn_act_bwd_num_parts(long long rows) {
  return ceil_div(rows, g_tiling[1][1]);
}

extern "C" int os2s_bn_act_bwd_reduce(os2s_stream_t stream, int J, const uint16_t* dout,
                                      const uint16_t* out, const uint16_t* const* y,
                                      const float* const* mean, const float* const* rstd,
                                      uint16_t* dz, float* partial, const int32_t* out_len,
                                      int B, int T, int C, int act, float keep_prob,
                                      unsigned long long seed) {
  OS2S_REQUIRE(J >= 1 && J <= kMaxBnInputs && dout && out && dz && partial);
  OS2S_REQUIRE(C % 8 == 0 && C >= 8 && keep_prob > 0.f && keep_prob <= 1.f);
  if ((long long)B * T == 0) return OS2S_OK;
  BnBwdReduceArgs a;
  for (int j = 0; j < J; ++j) {
    OS2S_REQUIRE(y[j] && mean[j] && rstd[j]);
    a.y[j] = y[j]; a.mean[j] = mean[j]; a.rstd[j] = rstd[j];
  }
  a.dout = dout; a.out = out; a.J = J; a.dz = dz; a.partial = partial; a.out_len = out_len;
  a.B = B; a.T = T; a.C = C; a.act = act; a.keep_prob = keep_prob; a.seed = seed;
  a.rows_per_block = g_tiling[1][1];
  const int C8 = C / 8;
  const int gmax = g_tiling[1][0];
  dim3 grid(ceil_div((long long)B * T, a.rows_per_block), tile_cblocks(C8, gmax));
  const size_t smem = 0;
  if (J <= 1) {
    OS2S_LAUNCH((bn_act_bwd_reduce_kernel<1, 4>), grid, dim3(256), smem, (hipStream_t)stream, a, gmax);
  } else if (J <= 4) {
    OS2S_LAUNCH((bn_act_bwd_reduce_kernel<4, 2>), grid, dim3(256), smem, (hipStream_t)stream, a, gmax);
  } else {
    OS2S_LAUNCH((bn_act_bwd_reduce_kernel<kMaxBnInputs, 1>), grid, dim3(256), smem,
                (hipStream_t)stream, a, gmax);
  }
  return OS2S_OK;
}

extern "C" int os2s_bn_bwd_finalize(os2s_stream_t stream, const float* partial, int nparts,
                                    int nq, int q, int C, long long count, float* dgamma,
                                    float* dbeta, int accumulate, float* c1, float* c2) {
  OS2S_REQUIRE(partial && c1 && c2 && nparts >= 1 && q >= 1 && q < nq && count >= 1);
  OS2S_LAUNCH(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * kFinLanes), 0,
              (hipStream_t)stream, partial, nparts, nq, q, C, (double)count, dgamma, dbeta,
              accumulate, c1, c2);
  return OS2S_OK;
}

// The same from RAW partials [nparts, 2, C] = (sum dz, sum dz * y) per part (the data-gradient epilogue of
// os2s_conv1d_dgrad_bnact_ws): sum dz * xhat = rstd * (sum dz * y - mean * sum dz).
namespace os2s {
__global__ __launch_bounds__(64 * kFinLanes) void bn_bwd_finalize_raw_kernel(
    const float* __restrict__ partial, int nparts, int C, double count, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double sh_d[kFinLanes][64], sh_x[kFinLanes][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double sd = 0.0, sx = 0.0;
  if (c < C) fin_sum_pair(partial, nparts, 2, 0, 1, C, c, pl, sd, sx);
  sh_d[pl][cl] = sd;
  sh_x[pl][cl] = sx;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  sd = 0.0; sx = 0.0;
#pragma unroll
  for (int l = 0; l < kFinLanes; ++l) { sd += sh_d[l][cl]; sx += sh_x[l][cl]; }
  sx = (double)rstd[c] * (sx - (double)mean[c] * sd);
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sd;
  c1[c] = (float)(sd / count);
  c2[c] = (float)(sx / count);
}
}  // namespace os2s

extern "C" int os2s_bn_bwd_finalize_raw(os2s_stream_t stream, const float* partial, int nparts, int C,
                                        long long count, const float* mean, const float* rstd, float* dgamma,
                                        float* dbeta, int accumulate, float* c1, float* c2) {
  using namespace os2s;
  OS2S_REQUIRE(partial && mean && rstd && c1 && c2 && nparts >= 1 && C >= 1 && count >= 1);
  OS2S_LAUNCH(bn_bwd_finalize_raw_kernel, dim3(ceil_div(C, 64)), dim3(64 * kFinLanes), 0, (hipStream_t)stream,
              partial, nparts, C, (double)count, mean, rstd, dgamma, dbeta, accumulate, c1, c2);
  return OS2S_OK;
}

extern "C" int os2s_bn_bwd_finalize_multi(os2s_stream_t stream, const float* partial, int nparts,
                                          int J, int C, long long count, float* const* dgamma,
                                          float* const* dbeta, int accumulate, float* c1,
                                          float* c2) {
  OS2S_REQUIRE(partial && c1 && c2 && dgamma && dbeta && nparts >= 1 && J >= 1 && J <= kMaxBnInputs &&
               count >= 1);
  BnFinMulti ptrs;
  for (int j = 0; j < J; ++j) { ptrs.dgamma[j] = dgamma[j]; ptrs.dbeta[j] = dbeta[j]; }
  OS2S_LAUNCH(bn_bwd_finalize_multi_kernel, dim3(ceil_div(C, 64), J), dim3(64 * kFinLanes), 0,
              (hipStream_t)stream, partial, nparts, 1 + J, C, (double)count, ptrs, accumulate, c1, c2);
  return OS2S_OK;
}

static int bn_bwd_apply_launch(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                               const float* gamma, const float* mean, const float* rstd,
                               const float* c1, const float* c2, uint16_t* dy,
                               const int32_t* out_len, int margin, int B, int T, int C, int dz_to_len = 0) {
  const int gmax = g_tiling[2][0], trows = g_tiling[2][1];
  dim3 grid(ceil_div((long long)B * T, trows), tile_cblocks(C / 8, gmax));
  OS2S_LAUNCH(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, dz, y, gamma, mean,
              rstd, c1, c2, dy, out_len, margin, B, T, C, gmax, trows, dz_to_len);
  return OS2S_OK;
}

extern "C" int os2s_bn_bwd_apply_ragged(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                                        const float* gamma, const float* mean, const float* rstd,
                                        const float* c1, const float* c2, uint16_t* dy,
                                        const int32_t* out_len, int margin, int B, int T, int C) {
  OS2S_REQUIRE(dz && y && mean && rstd && c1 && c2 && dy && C % 8 == 0 && B >= 0 && T >= 0);
  OS2S_REQUIRE(margin >= 0);
  if ((long long)B * T == 0) return OS2S_OK;
  return bn_bwd_apply_launch(stream, dz, y, gamma, mean, rstd, c1, c2, dy, out_len, margin, B, T, C);
}

// the same for a dz that is defined only up to the sequence ends (zero beyond, never read there)
extern "C" int os2s_bn_bwd_apply_ragged_dz(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                                           const float* gamma, const float* mean, const float* rstd,
                                           const float* c1, const float* c2, uint16_t* dy,
                                           const int32_t* out_len, int margin, int B, int T, int C) {
  OS2S_REQUIRE(dz && y && mean && rstd && c1 && c2 && dy && out_len && C % 8 == 0 && B >= 0 && T >= 0);
  OS2S_REQUIRE(margin >= 0);
  if ((long long)B * T == 0) return OS2S_OK;
  return bn_bwd_apply_launch(stream, dz, y, gamma, mean, rstd, c1, c2, dy, out_len, margin, B, T, C, 1);
}

extern "C" int os2s_bn_bwd_apply(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                                 const float* gamma, const float* mean, const float* rstd,
                                 const float* c1, const float* c2, uint16_t* dy,
                                 long long rows, int C) {
  OS2S_REQUIRE(dz && y && mean && rstd && c1 && c2 && dy && C % 8 == 0);
  OS2S_REQUIRE(rows >= 0 && rows < (1LL << 31));
  if (rows == 0) return OS2S_OK;
  return bn_bwd_apply_launch(stream, dz, y, gamma, mean, rstd, c1, c2, dy, nullptr, 0, 1, (int)rows, C);
}

extern "C" int os2s_dropout_mask(os2s_stream_t stream, unsigned long long seed,
                                 long long n8, float keep_prob, uint8_t* out) {
  OS2S_REQUIRE(out && n8 >= 0);
  if (n8 == 0) return OS2S_OK;
  OS2S_LAUNCH(dropout_mask_kernel, dim3(ew_grid(n8)), dim3(256), 0, (hipStream_t)stream,
              seed, n8, keep_prob, out);
  return OS2S_OK;
}
