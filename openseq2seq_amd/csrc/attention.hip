// Multi-head scaled-dot-product attention (forward + backward) for short
// sequences (Lq, Lk <= 64: the training regime of the Transformer configs,
// max_length 56 in example_configs/text2text/en-de/transformer-big.py:104),
// on PACKED token-major tensors, gfx950.
//
// Reference: Attention.call (open_seq2seq/parts/transformer/attention_layer.py:104-220):
//   q *= depth**-0.5; logits = q k^T (fp32 softmax when activations are half precision)
//   + bias (-1e9 on padded keys, parts/transformer/utils.py:82-129; causal band for the
//   decoder self-attention, utils.py:57-79); softmax; dropout(keep = 1-attention_dropout);
//   weights @ v.  split_heads/combine_heads are pure indexing: head h owns channels
//   [h*dh, (h+1)*dh) of the [tokens, hidden] projections, so no transposes are needed.
// Packed layout: sequences are concatenated without padding; cu_q / cu_k [B+1] give the
// token offsets. Padded keys simply do not exist (== the reference's -1e9 bias, whose
// exp underflows to exactly 0 in fp32).
//
// One WAVE owns one (batch, head): all products are 64x64x64 tiles of
// v_mfma_f32_32x32x16_bf16, always in the "swapped" orientation (result rows = the
// contiguous output dimension) so each lane ends up with 4 consecutive output elements
// (8-byte stores) and the softmax row of a query is lane-local (16+16 registers + one
// cross-half exchange). Operands whose reduction index is the row index in memory
// (V in P.V, and dO/K/Q in the backward products) are fetched with ds_read_b64_tr_b16.
// The backward recomputes P from (q, k, lse) — no [B,H,L,L] tensor is ever stored.
#include "os2s_common.hpp"

namespace os2s {

constexpr int kL = 64;    // tile edge (max sequence length handled)
constexpr int kDh = 64;   // head dim

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // [Nq, ld], [Nk, ld], [Nk, ld]
  bf16_t* o;                                           // [Nq, ldo]
  float* lse;                                          // [Nq, H]
  const int32_t* cu_q; const int32_t* cu_k;            // [B+1]
  int B, H;
  long long ldq, ldk, ldv, ldo;
  int causal;
  float scale, keep_prob;
  unsigned long long seed;
  // backward only
  const bf16_t* d_o; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  long long lddo, lddq, lddk, lddv;
};

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
  return z;
}

// k-contiguous operand straight from global: row-major [rows][ld], 8 consecutive k
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* base, long long ld, int row, int nvalid,
                                              int kofs) {
  if (row >= nvalid) return zero8();
  return *reinterpret_cast<const bf16x8*>(base + (long long)row * ld + kofs);
}

// ---- LDS images (64 x 64 bf16 = 8 KB each, 128-byte rows) -----------------
// "kc": row-major, k contiguous, 16-B slot XOR swizzle (conflict-free ds_read_b128)
__device__ __forceinline__ int kc_off(int row, int k) {   // byte offset of element (row, k)
  return row * 128 + ((((k >> 3) ^ ((row >> 1) & 7))) << 4) + (k & 7) * 2;
}
__device__ __forceinline__ bf16x8 frag_kc(const char* buf, int row, int kslot) {
  return *reinterpret_cast<const bf16x8*>(buf + row * 128 + ((kslot ^ ((row >> 1) & 7)) << 4));
}
// "tr": rows = reduction index, 32-B unit XOR swizzle for ds_read_b64_tr_b16
__device__ __forceinline__ int tr_off(int row, int m) {   // byte offset of element (row=k, m)
  const int u = (m >> 4) ^ (((row >> 1) & 1) << 1);
  return row * 128 + (u << 5) + (m & 15) * 2;
}
__device__ __forceinline__ bf16x8 frag_tr(const char* buf, int mtile, int kk, int lane) {
  // operand[m][k] with m = mtile*32 + (lane&31), k = kk*16 + (lane>>5)*8 + 0..7, from [k][m]
  const int g16 = (lane >> 4) & 1, i16 = lane & 15;
  const int unit = mtile * 2 + g16;
  const int r0 = kk * 16 + (lane >> 5) * 8 + (i16 >> 2);
  const int r1 = r0 + 4;
  const int c = (i16 & 3) * 8;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(buf + r0 * 128 +
                                                   ((unit ^ (((r0 >> 1) & 1) << 1)) << 5) + c));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(buf + r1 * 128 +
                                                   ((unit ^ (((r1 >> 1) & 1) << 1)) << 5) + c));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// copy a [rows<=64][64] bf16 tile from global into a "tr" LDS image (zero fill)
__device__ __forceinline__ void stage_tr(char* buf, const bf16_t* base, long long ld, int nvalid,
                                         int lane) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int piece = it * 64 + lane;         // 512 pieces of 16 B
    const int row = piece >> 3, p8 = piece & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *reinterpret_cast<const u32x4*>(base + (long long)row * ld + p8 * 8);
    *reinterpret_cast<u32x4*>(buf + tr_off(row, p8 * 8)) = v;
  }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// S^T tile set: s[i][j][r] : key = i*32 + 4*hi + (r&3) + 8*(r>>2), query = j*32 + (lane&31)
__device__ __forceinline__ void scores(const AttnArgs& p, const bf16_t* qb, const bf16_t* kb, int Lq,
                                       int Lk, int lane, f32x16 (&s)[2][2]) {
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[i][j][e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = frag_global(kb, p.ldk, i * 32 + l31, Lk, kk * 16 + lhi * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = frag_global(qb, p.ldq, j * 32 + l31, Lq, kk * 16 + lhi * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        s[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], s[i][j], 0, 0, 0);
  }
}

__device__ __forceinline__ int key_of(int i, int r, int lhi) {
  return i * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
}

// dropout keep bits for the 4 consecutive keys key0..key0+3 of query row (bh, q)
__device__ __forceinline__ uint32_t attn_keep4(const AttnArgs& p, long long bh, int q, int key0) {
  const long long e0 = ((bh * kL + q) * kL + key0);
  return (dropout_bits8(p.seed, (unsigned long long)(e0 >> 3), p.keep_prob) >> (uint32_t)(e0 & 7)) & 0xfu;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
constexpr int kFwdWaves = 4;

__global__ __launch_bounds__(kFwdWaves * 64) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  long long bh = (long long)blockIdx.x * kFwdWaves + wid;
  const bool live = bh < (long long)p.B * p.H;
  if (!live) bh = 0;
  const int b = (int)(bh / p.H), h = (int)(bh - (long long)b * p.H);
  const int q0 = p.cu_q[b], k0 = p.cu_k[b];
  const int Lq = min(p.cu_q[b + 1] - q0, kL), Lk = min(p.cu_k[b + 1] - k0, kL);
  char* pm = smem + wid * 16384;        // PM[q][key]  (kc image)
  char* vt = pm + 8192;                 // V[key][d]   (tr image)
  const bf16_t* qb = p.q + (long long)q0 * p.ldq + h * kDh;
  const bf16_t* kb = p.k + (long long)k0 * p.ldk + h * kDh;
  const bf16_t* vb = p.v + (long long)k0 * p.ldv + h * kDh;

  f32x16 s[2][2];
  scores(p, qb, kb, Lq, Lk, lane, s);
  stage_tr(vt, vb, p.ldv, Lk, lane);
  // ---- softmax over keys (rows) for each query column ------------------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = j * 32 + l31;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key_of(i, r, lhi);
        const bool ok = key < Lk && !(p.causal && key > q);
        const float v = ok ? s[i][j][r] * p.scale : -INFINITY;
        s[i][j][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, xhalf(m));
    const float msafe = m == -INFINITY ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(s[i][j][r] - msafe);
        s[i][j][r] = e;
        l += e;
      }
    l += xhalf(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (live && lhi == 0 && q < Lq && p.lse) p.lse[(long long)(q0 + q) * p.H + h] = msafe + __logf(l);
    const float ik = 1.f / p.keep_prob;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = i * 32 + 8 * g + 4 * lhi;
        uint32_t keep = 0xfu;
        if (p.keep_prob < 1.f) keep = attn_keep4(p, bh, q, key0);
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pv = s[i][j][4 * g + e] * inv;
          if (p.keep_prob < 1.f) pv = ((keep >> e) & 1u) ? pv * ik : 0.f;
          w[e] = pv;
        }
        u32x2 pk;
        pk[0] = pack2bf(w[0], w[1]);
        pk[1] = pack2bf(w[2], w[3]);
        *reinterpret_cast<u32x2*>(pm + kc_off(q, key0)) = pk;
      }
  }
  __syncthreads();
  // ---- O^T[d][q] = sum_key V^T[d][key] * PM^T[key][q] -------------------------
  f32x16 o[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][j][e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[2], bq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = frag_tr(vt, i, kk, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) bq[j] = frag_kc(pm, j * 32 + l31, kk * 2 + lhi);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        o[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], o[i][j], 0, 0, 0);
  }
  if (live) {
    bf16_t* ob = p.o + (long long)q0 * p.ldo + h * kDh;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = j * 32 + l31;
      if (q < Lq) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = i * 32 + 8 * g + 4 * lhi;
            u32x2 pk;
            pk[0] = pack2bf(o[i][j][4 * g], o[i][j][4 * g + 1]);
            pk[1] = pack2bf(o[i][j][4 * g + 2], o[i][j][4 * g + 3]);
            *reinterpret_cast<u32x2*>(ob + (long long)q * p.ldo + d0) = pk;
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// forward for sequences longer than one tile (inference: eval / infer batches are not
// length-filtered like the training set). One wave per (batch, head, 64-query tile) walks the
// key tiles with an online softmax; per-lane state because a lane's MFMA column IS its query.
// No dropout on this path (keep_prob == 1).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kFwdWaves * 64) void attn_fwd_long_kernel(AttnArgs p, int q_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const long long work = (long long)blockIdx.x * kFwdWaves + wid;
  if (work >= (long long)p.B * p.H * q_tiles) return;       // waves are independent (own LDS slice)
  const long long bh = work / q_tiles;
  const int qt = (int)(work - bh * q_tiles);
  const int b = (int)(bh / p.H), h = (int)(bh - (long long)b * p.H);
  const int q0 = p.cu_q[b], k0 = p.cu_k[b];
  const int Lq_tot = p.cu_q[b + 1] - q0, Lk_tot = p.cu_k[b + 1] - k0;
  const int qs = qt * kL;
  if (qs >= Lq_tot) return;
  const int Lq = min(Lq_tot - qs, kL);
  char* pm = smem + wid * 16384;        // P[q][key]  (kc image)
  char* vt = pm + 8192;                 // V[key][d]  (tr image)
  const bf16_t* qb = p.q + (long long)(q0 + qs) * p.ldq + h * kDh;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  f32x16 o[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][j][e] = 0.f;
  const int nkt = p.causal ? min(qt + 1, (Lk_tot + kL - 1) / kL) : (Lk_tot + kL - 1) / kL;
  for (int kt = 0; kt < nkt; ++kt) {
    const int ks = kt * kL;
    const int Lk = min(Lk_tot - ks, kL);
    const bf16_t* kb = p.k + (long long)(k0 + ks) * p.ldk + h * kDh;
    const bf16_t* vb = p.v + (long long)(k0 + ks) * p.ldv + h * kDh;
    f32x16 s[2][2];
    scores(p, qb, kb, Lq, Lk, lane, s);
    stage_tr(vt, vb, p.ldv, Lk, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = j * 32 + l31;
      float mt = -INFINITY;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_of(i, r, lhi);
          const bool ok = key < Lk && !(p.causal && ks + key > qs + q);
          const float v = ok ? s[i][j][r] * p.scale : -INFINITY;
          s[i][j][r] = v;
          mt = fmaxf(mt, v);
        }
      mt = fmaxf(mt, xhalf(mt));
      const float mn = fmaxf(m_run[j], mt);
      const float msafe = mn == -INFINITY ? 0.f : mn;
      const float corr = m_run[j] == -INFINITY ? 0.f : __expf(m_run[j] - msafe);
      float lsum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __expf(s[i][j][r] - msafe);      // exp(-inf) = 0 for masked keys
          s[i][j][r] = e;
          lsum += e;
        }
      lsum += xhalf(lsum);
      l_run[j] = l_run[j] * corr + lsum;
      m_run[j] = mn;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][j][e] *= corr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = i * 32 + 8 * g + 4 * lhi;
          u32x2 pk;
          pk[0] = pack2bf(s[i][j][4 * g], s[i][j][4 * g + 1]);
          pk[1] = pack2bf(s[i][j][4 * g + 2], s[i][j][4 * g + 3]);
          *reinterpret_cast<u32x2*>(pm + kc_off(q, key0)) = pk;
        }
    }
    __builtin_amdgcn_wave_barrier();      // LDS ops of one wave complete in order
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 a[2], bq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag_tr(vt, i, kk, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[j] = frag_kc(pm, j * 32 + l31, kk * 2 + lhi);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          o[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], o[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  bf16_t* ob = p.o + (long long)(q0 + qs) * p.ldo + h * kDh;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = j * 32 + l31;
    if (q < Lq) {
      const float inv = l_run[j] > 0.f ? 1.f / l_run[j] : 0.f;
      if (lhi == 0 && p.lse)
        p.lse[(long long)(q0 + qs + q) * p.H + h] = (m_run[j] == -INFINITY ? 0.f : m_run[j]) + __logf(l_run[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = i * 32 + 8 * g + 4 * lhi;
          u32x2 pk;
          pk[0] = pack2bf(o[i][j][4 * g] * inv, o[i][j][4 * g + 1] * inv);
          pk[1] = pack2bf(o[i][j][4 * g + 2] * inv, o[i][j][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(ob + (long long)q * p.ldo + d0) = pk;
        }
    }
  }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// One WORKGROUP of four waves per (batch, head); wave (wi, wj) owns the 32 x 32 quadrant
// (wi, wj) of every 64 x 64 product. The kernel is a latency chain per (batch, head) — global
// loads, two products, the softmax algebra, three gradient products — so what matters is how
// short a wave's share of the chain is and how many waves a CU holds to hide it: a quarter of
// the loads, MFMAs and transcendental work per wave, ~100 registers (the one-wave version held
// three full 64 x 64 accumulator sets = 256 registers) and 40 KB of LDS per workgroup = 16 waves
// per CU instead of 6. dO, K and Q are staged ONCE, up front, as transpose-read images (the
// one-wave version re-staged one image between the gradient products, each a global round trip
// in the middle of the chain).
constexpr int kBwdThreads = 256;
constexpr int kBwdLds = 5 * 8192;       // dO, K, Q images + PM + dS (the delta exchange aliases dS)

// copy a [rows<=64][64] bf16 tile from global into a "tr" LDS image (zero fill), 256 threads
__device__ __forceinline__ void stage_tr4(char* buf, const bf16_t* base, long long ld, int nvalid, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int piece = it * kBwdThreads + tid;   // 512 pieces of 16 B
    const int row = piece >> 3, p8 = piece & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < nvalid) v = *reinterpret_cast<const u32x4*>(base + (long long)row * ld + p8 * 8);
    *reinterpret_cast<u32x4*>(buf + tr_off(row, p8 * 8)) = v;
  }
}

// store quadrant D[m = d block i][n block j] (lane col n, 4 consecutive d per reg group) to
// out[n][d], rows n < nvalid
__device__ __forceinline__ void store_dT_quad(const f32x16& d, int i, int j, bf16_t* out, long long ld,
                                              int nvalid, float mul, int lane) {
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n = j * 32 + l31;
  if (n < nvalid) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = i * 32 + 8 * g + 4 * lhi;
      u32x2 pk;
      pk[0] = pack2bf(d[4 * g] * mul, d[4 * g + 1] * mul);
      pk[1] = pack2bf(d[4 * g + 2] * mul, d[4 * g + 3] * mul);
      *reinterpret_cast<u32x2*>(out + (long long)n * ld + d0) = pk;
    }
  }
}

__global__ __launch_bounds__(kBwdThreads) void attn_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wj = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const long long bh = blockIdx.x;
  const int b = (int)(bh / p.H), h = (int)(bh - (long long)b * p.H);
  const int q0 = p.cu_q[b], k0 = p.cu_k[b];
  const int Lq = min(p.cu_q[b + 1] - q0, kL), Lk = min(p.cu_k[b + 1] - k0, kL);
  char* const do_img = smem;            // dO[q][d]   tr image (rows = q)
  char* const k_img = smem + 8192;      // K[key][d]  tr image (rows = key)
  char* const q_img = smem + 16384;     // Q[q][d]    tr image (rows = q)
  char* const pm = smem + 24576;        // PM[q][key] tr image (rows = q)
  char* const ds = smem + 32768;        // dS[q][key] tr image (rows = q); also read row-wise for dQ
  float* const dbuf = reinterpret_cast<float*>(ds);   // [2][64] partial deltas, before dS is written
  const bf16_t* qb = p.q + (long long)q0 * p.ldq + h * kDh;
  const bf16_t* kb = p.k + (long long)k0 * p.ldk + h * kDh;
  const bf16_t* vb = p.v + (long long)k0 * p.ldv + h * kDh;
  const bf16_t* dob = p.d_o + (long long)q0 * p.lddo + h * kDh;

  // ---- every global read of the workgroup is issued here -----------------------------------
  bf16x8 fk[4], fq[4], fv[4], fdo[4];
  const int key_row = wi * 32 + l31, q_row = wj * 32 + l31;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fk[kk] = frag_global(kb, p.ldk, key_row, Lk, kk * 16 + lhi * 8);
    fq[kk] = frag_global(qb, p.ldq, q_row, Lq, kk * 16 + lhi * 8);
    fv[kk] = frag_global(vb, p.ldv, key_row, Lk, kk * 16 + lhi * 8);
    fdo[kk] = frag_global(dob, p.lddo, q_row, Lq, kk * 16 + lhi * 8);
  }
  const int q = q_row;
  const float lse = (q < Lq) ? p.lse[(long long)(q0 + q) * p.H + h] : 0.f;
  stage_tr4(do_img, dob, p.lddo, Lq, tid);
  stage_tr4(k_img, kb, p.ldk, Lk, tid);
  stage_tr4(q_img, qb, p.ldq, Lq, tid);

  // ---- S^T[key][q] = K Q^T and dPM^T[key][q] = V dO^T, this wave's quadrant -------------------
  f32x16 s, dp;
#pragma unroll
  for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[kk], fq[kk], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[kk], fdo[kk], dp, 0, 0, 0);
  }
  // ---- P, M, partial delta = sum over this wave's 32 keys of P * M * dPM; PM[q][key] to LDS ------
  const float ik = 1.f / p.keep_prob;
  float delta = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int key0 = wi * 32 + 8 * g + 4 * lhi;
    uint32_t keep = 0xfu;
    if (p.keep_prob < 1.f) keep = attn_keep4(p, bh, q, key0);
    float pmv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e, key = key0 + e;
      const bool ok = key < Lk && q < Lq && !(p.causal && key > q);
      const float pv = ok ? __expf(s[r] * p.scale - lse) : 0.f;
      float mk = 1.f;
      if (p.keep_prob < 1.f) mk = ((keep >> e) & 1u) ? ik : 0.f;
      s[r] = pv;                  // P
      dp[r] *= mk;                // dP = dPM * M
      delta += pv * dp[r];
      pmv[e] = pv * mk;
    }
    u32x2 pk;
    pk[0] = pack2bf(pmv[0], pmv[1]);
    pk[1] = pack2bf(pmv[2], pmv[3]);
    *reinterpret_cast<u32x2*>(pm + tr_off(q, key0)) = pk;
  }
  delta += xhalf(delta);
  if (lhi == 0) dbuf[wi * 64 + q] = delta;
  __syncthreads();
  delta += dbuf[(wi ^ 1) * 64 + q];      // the other 32 keys of this query (wave (1 - wi, wj))
  __syncthreads();                        // dbuf is overwritten by the dS image next
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int key0 = wi * 32 + 8 * g + 4 * lhi;
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = s[4 * g + e] * (dp[4 * g + e] - delta);   // dS (w.r.t. the scaled logits)
    u32x2 pk;
    pk[0] = pack2bf(w[0], w[1]);
    pk[1] = pack2bf(w[2], w[3]);
    *reinterpret_cast<u32x2*>(ds + tr_off(q, key0)) = pk;
  }
  __syncthreads();
  // ---- the three gradient products, quadrant (wi = d block, wj = key / query block) ------------
  f32x16 dv, dq, dk;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dv[e] = 0.f; dq[e] = 0.f; dk[e] = 0.f; }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    // dV^T[d][key] = sum_q dO[q][d] PM[q][key]: both operands by transpose reads (rows = q)
    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(do_img, wi, kk, lane), frag_tr(pm, wj, kk, lane), dv, 0, 0, 0);
    // dQ^T[d][q] = scale * sum_key K[key][d] dS[q][key]: B operand = 8 consecutive keys of row q
    const bf16x8 dsrow = *reinterpret_cast<const bf16x8*>(ds + tr_off(wj * 32 + l31, kk * 16 + lhi * 8));
    dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(k_img, wi, kk, lane), dsrow, dq, 0, 0, 0);
    // dK^T[d][key] = scale * sum_q Q[q][d] dS[q][key]: both operands by transpose reads (rows = q)
    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(q_img, wi, kk, lane), frag_tr(ds, wj, kk, lane), dk, 0, 0, 0);
  }
  store_dT_quad(dv, wi, wj, p.dv + (long long)k0 * p.lddv + h * kDh, p.lddv, Lk, 1.f, lane);
  store_dT_quad(dq, wi, wj, p.dq + (long long)q0 * p.lddq + h * kDh, p.lddq, Lq, p.scale, lane);
  store_dT_quad(dk, wi, wj, p.dk + (long long)k0 * p.lddk + h * kDh, p.lddk, Lk, p.scale, lane);
}

}  // namespace os2s

using namespace os2s;

static int attn_check(const int32_t* cu_q, const int32_t* cu_k, int B, int H, int dh, int max_len) {
  if (!cu_q || !cu_k || B < 1 || H < 1) return OS2S_ERR_INVALID_ARG;
  if (dh != kDh || max_len > kL) return OS2S_ERR_UNSUPPORTED;
  return OS2S_OK;
}

extern "C" int os2s_attention_fwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                                  const uint16_t* v, uint16_t* o, float* lse,
                                  const int32_t* cu_q, const int32_t* cu_k, int B, int H, int dh,
                                  int max_len, long long ldq, long long ldk, long long ldv,
                                  long long ldo, int causal, float scale, float keep_prob,
                                  unsigned long long seed) {
  OS2S_REQUIRE(q && k && v && o);
  int rc = attn_check(cu_q, cu_k, B, H, dh, max_len > kL ? kL : max_len);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0);
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.cu_q = cu_q; a.cu_k = cu_k; a.B = B; a.H = H;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.causal = causal; a.scale = scale;
  a.keep_prob = keep_prob; a.seed = seed;
  const size_t smem = (size_t)kFwdWaves * 16384;
  if (max_len > kL) {      // multi-tile forward (inference); no attention dropout on this path
    if (keep_prob < 1.f) return OS2S_ERR_UNSUPPORTED;
    static bool attr_long = false;
    if (!attr_long) {
      if (hipFuncSetAttribute((const void*)attn_fwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess) return OS2S_ERR_LAUNCH;
      attr_long = true;
    }
    const int q_tiles = (max_len + kL - 1) / kL;
    OS2S_LAUNCH(attn_fwd_long_kernel, dim3(ceil_div((long long)B * H * q_tiles, kFwdWaves)),
                dim3(kFwdWaves * 64), smem, (hipStream_t)stream, a, q_tiles);
    return OS2S_OK;
  }
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess) return OS2S_ERR_LAUNCH;
    attr = true;
  }
  OS2S_LAUNCH(attn_fwd_kernel, dim3(ceil_div((long long)B * H, kFwdWaves)), dim3(kFwdWaves * 64),
              smem, (hipStream_t)stream, a);
  return OS2S_OK;
}

extern "C" int os2s_attention_bwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                                  const uint16_t* v, const uint16_t* d_o, const float* lse,
                                  uint16_t* dq, uint16_t* dk, uint16_t* dv, const int32_t* cu_q,
                                  const int32_t* cu_k, int B, int H, int dh, int max_len,
                                  long long ldq, long long ldk, long long ldv, long long lddo,
                                  long long lddq, long long lddk, long long lddv, int causal,
                                  float scale, float keep_prob, unsigned long long seed) {
  OS2S_REQUIRE(q && k && v && d_o && lse && dq && dk && dv);
  int rc = attn_check(cu_q, cu_k, B, H, dh, max_len);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0);
  OS2S_REQUIRE(lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.lse = const_cast<float*>(lse); a.cu_q = cu_q; a.cu_k = cu_k;
  a.B = B; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.causal = causal; a.scale = scale;
  a.keep_prob = keep_prob; a.seed = seed; a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
  a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  static bool attr = false;
  const size_t smem = (size_t)kBwdLds;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess) return OS2S_ERR_LAUNCH;
    attr = true;
  }
  OS2S_LAUNCH(attn_bwd_kernel, dim3((unsigned)((long long)B * H)), dim3(kBwdThreads), smem, (hipStream_t)stream, a);
  return OS2S_OK;
}
