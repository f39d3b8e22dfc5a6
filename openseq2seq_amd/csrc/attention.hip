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
constexpr long long kMaxLd = 1ll << 22;   // row strides (elements): 64 rows of it stay inside 31 bits of bytes

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

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
constexpr int kFwdWaves = 4;

// ---- helpers shared by the single-tile forward and the backward ------------------------------
// Descriptor over rows [0, rows) x 64 channels of one head of a row-major [*, ld] tensor: a load of a
// row >= rows is out of range and returns zeros — no branch, no address clamp (the branchy version
// spent 600 instructions of the backward's prologue on guards and 64-bit address arithmetic).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t head_rsrc(const bf16_t* base, long long ld, int rows) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int bytes = __builtin_amdgcn_readfirstlane(rows > 0 ? (int)((long long)(rows - 1) * ld * 2 + kDh * 2) : 0);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 load8(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// the two 32-lane halves of a wave exchange one register each (v_permlane32_swap: a VALU move, not a
// trip through the LDS crossbar): r[0] = {a.lo, b.lo}, r[1] = {a.hi, b.hi} as (lanes 0-31, lanes 32-63)
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// A 32 x 32 MFMA result holds, per lane (column n = lane & 31), rows 8g + 4*(lane >> 5) + 0..3 in
// registers 4g .. 4g+3. For the register groups g0, g0 + 1 return 8 CONSECUTIVE rows as bf16: lanes
// 0-31 get rows 8*g0 .. +7, lanes 32-63 rows 8*(g0+1) .. +7 (one 16-byte store instead of two 8-byte ones).
__device__ __forceinline__ u32x4 rows8(const f32x16& d, int g0, float mul) {
  const uint32_t a0 = pack2bf(d[4 * g0] * mul, d[4 * g0 + 1] * mul);
  const uint32_t a1 = pack2bf(d[4 * g0 + 2] * mul, d[4 * g0 + 3] * mul);
  const uint32_t b0 = pack2bf(d[4 * g0 + 4] * mul, d[4 * g0 + 5] * mul);
  const uint32_t b1 = pack2bf(d[4 * g0 + 6] * mul, d[4 * g0 + 7] * mul);
  const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
  u32x4 o;
  o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
  return o;
}
// store block D[m = d block i][n block j] (lane column n, rows = d) to out[n][d], rows n < nvalid
__device__ __forceinline__ void store_dT_quad(const f32x16& d, int i, int j, bf16_t* out, long long ld,
                                              int nvalid, float mul, int lane) {
  const int n = j * 32 + (lane & 31);
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) {
    const u32x4 v = rows8(d, 2 * gp, mul);
    if (n < nvalid)
      *reinterpret_cast<u32x4*>(out + (long long)n * ld + i * 32 + 8 * (2 * gp + (lane >> 5))) = v;
  }
}
// frag_tr with the 16 reduction rows of step kk taken in the order the score accumulators hold them:
// slot (lane >> 5, e) <-> row kk*16 + 4*(lane >> 5) + (e & 3) + 8*(e >> 2), so a lane's own P values
// (register groups 2*(kk&1), 2*(kk&1)+1 of key block kk >> 1) ARE its B operand — P never visits LDS
__device__ __forceinline__ bf16x8 frag_tr_acc(const char* buf, int mtile, int kk, int lane) {
  const int g16 = (lane >> 4) & 1, i16 = lane & 15;
  const int unit = mtile * 2 + g16;
  const int r0 = kk * 16 + (lane >> 5) * 4 + (i16 >> 2);
  const int r1 = r0 + 8;
  const int c = (i16 & 3) * 8;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(buf + r0 * 128 +
                                                   ((unit ^ (((r0 >> 1) & 1) << 1)) << 5) + c));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(buf + r1 * 128 +
                                                   ((unit ^ (((r1 >> 1) & 1) << 1)) << 5) + c));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// The dropout decisions of 4 consecutive keys = ONE of the two mixes of dropout_bits8 (the element
// index ((bh*64 + q)*64 + key0) / 4 selects it); successive register groups advance that index by a
// compile-time constant, so the 64-bit multiply by the golden ratio happens once per lane.
constexpr unsigned long long attn_drop_step(int idx4_delta) { return (unsigned long long)idx4_delta * kDropGolden; }

// One WAVE per (batch, head); waves of a workgroup are independent (no workgroup barrier). Blocks of
// 32 keys / 32 queries past the sequence's length — and, with the causal band, the block above the
// diagonal — are skipped outright (wave-uniform branches): with max_length 56 half the sequences fit
// one 32-block, a quarter of the tile work. Only V^T goes through LDS (8 KB per wave).
template <bool kDrop>
__global__ __launch_bounds__(kFwdWaves * 64, 3) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bh = blockIdx.x * kFwdWaves + wid;      // B * H < 2^30 (checked by the host side)
  if (bh >= p.B * p.H) return;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = p.cu_q[b], k0 = p.cu_k[b];
  const int Lq = min(p.cu_q[b + 1] - q0, kL), Lk = min(p.cu_k[b + 1] - k0, kL);
  if (Lq <= 0) return;
  char* const vt = smem + wid * 8192;   // V[key][d]   (tr image)
  const int nj = Lq > 32 ? 2 : 1, ni = Lk > 32 ? 2 : 1;
  const bool causal = p.causal != 0;
  const __amdgpu_buffer_rsrc_t qrs = head_rsrc(p.q + (long long)q0 * p.ldq + h * kDh, p.ldq, Lq);
  const __amdgpu_buffer_rsrc_t krs = head_rsrc(p.k + (long long)k0 * p.ldk + h * kDh, p.ldk, Lk);
  const __amdgpu_buffer_rsrc_t vrs = head_rsrc(p.v + (long long)k0 * p.ldv + h * kDh, p.ldv, Lk);

  // ---- every global read is issued here -----------------------------------------------------
  bf16x8 ka[2][4], qb[2][4];
  {
    const int kv = l31 * (int)p.ldk * 2 + lhi * 16, qv = l31 * (int)p.ldq * 2 + lhi * 16;
    const int ks = 32 * (int)p.ldk * 2, qs = 32 * (int)p.ldq * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      ka[0][kk] = load8(krs, kv + kk * 32, 0);
      qb[0][kk] = load8(qrs, qv + kk * 32, 0);
      ka[1][kk] = load8(krs, kv + kk * 32, ks);      // past the length: zeros, no memory access
      qb[1][kk] = load8(qrs, qv + kk * 32, qs);
    }
  }
  const int Lk16 = (Lk + 15) & ~15;      // V rows the P.V reduction touches (zero-filled past Lk)
  u32x4 vst[8];
  {
    const int vv = (lane >> 3) * (int)p.ldv * 2 + (lane & 7) * 16;
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (it * 8 < Lk16) vst[it] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vv, it * 8 * (int)p.ldv * 2, 0);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it)
    if (it * 8 < Lk16)
      *reinterpret_cast<u32x4*>(vt + tr_off(it * 8 + (lane >> 3), (lane & 7) * 8)) = vst[it];
  // ---- S^T[key][q] per live 32 x 32 block -----------------------------------------------------
  f32x16 s[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[i][j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (i < ni && j < nj && !(causal && i > j)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          s[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[i][kk], qb[j][kk], s[i][j], 0, 0, 0);
      }
  // ---- softmax over keys (rows) for each query column; P stays in registers as the B operand ------
  const float sc2 = p.scale * 1.4426950408889634f;     // logits in the log2 domain: exp2 is one instruction
  const uint32_t thr = (uint32_t)(p.keep_prob * 65536.0f);
  const unsigned long long zlane =
      (unsigned long long)((long long)bh * (kL * kL / 4) + l31 * (kL / 4) + lhi) * kDropGolden + p.seed;
  bf16x8 pmb[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pmb[j][kk] = zero8();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j >= nj) continue;
    const int q = j * 32 + l31;
    // register r of block i is key i*32 + (r&3) + 8*(r>>2) + 4*lhi: live <=> that constant part < lim
    const int lim = min(Lk, causal ? q + 1 : kL) - 4 * lhi;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= ni || (causal && i > j)) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (i * 32 + (r & 3) + 8 * (r >> 2)) < lim ? s[i][j][r] * sc2 : -INFINITY;
        s[i][j][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = half_max(m);
    const float msafe = m == -INFINITY ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= ni || (causal && i > j)) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[i][j][r] - msafe);
        s[i][j][r] = e;
        l += e;
      }
    }
    l = half_sum(l);
    if (lhi == 0 && q < Lq && p.lse)
      p.lse[(long long)(q0 + q) * p.H + h] = (msafe + __log2f(l)) * 0.6931471805599453f;
    float mul = l > 0.f ? 1.f / l : 0.f;
    if (kDrop) mul *= 1.f / p.keep_prob;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= ni || (causal && i > j)) continue;
      uint32_t pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t keep = 0xfu;
        if (kDrop) keep = dropout_bits4_z(zlane + attn_drop_step(j * (32 * kL / 4) + i * 8 + 2 * g), thr);
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = ((keep >> e) & 1u) ? s[i][j][4 * g + e] * mul : 0.f;
        pk[g][0] = pack2bf(w[0], w[1]);
        pk[g][1] = pack2bf(w[2], w[3]);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        u32x4 t;
        t[0] = pk[2 * k2][0]; t[1] = pk[2 * k2][1]; t[2] = pk[2 * k2 + 1][0]; t[3] = pk[2 * k2 + 1][1];
        pmb[j][2 * i + k2] = __builtin_bit_cast(bf16x8, t);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();        // the V^T image is this wave's own: LDS ops of a wave are in order
  // ---- O^T[d][q] = sum_key V^T[d][key] * PM^T[key][q] ---------------------------------------
  f32x16 o[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][j][e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (kk * 16 >= Lk) continue;
    const bf16x8 a0 = frag_tr_acc(vt, 0, kk, lane), a1 = frag_tr_acc(vt, 1, kk, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= nj || (causal && (kk >> 1) > j)) continue;
      o[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pmb[j][kk], o[0][j], 0, 0, 0);
      o[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pmb[j][kk], o[1][j], 0, 0, 0);
    }
  }
  bf16_t* ob = p.o + (long long)q0 * p.ldo + h * kDh;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j >= nj) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) store_dT_quad(o[i][j], i, j, ob, p.ldo, Lq, 1.f, lane);
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

template <bool kDrop>
__global__ __launch_bounds__(kBwdThreads) void attn_bwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wj = wid & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bh = blockIdx.x;             // B * H < 2^31 (checked by the host side)
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = p.cu_q[b], k0 = p.cu_k[b];
  const int Lq = min(p.cu_q[b + 1] - q0, kL), Lk = min(p.cu_k[b + 1] - k0, kL);
  const bool causal = p.causal != 0;
  char* const do_img = smem;            // dO[q][d]   tr image (rows = q)
  char* const k_img = smem + 8192;      // K[key][d]  tr image (rows = key)
  char* const q_img = smem + 16384;     // Q[q][d]    tr image (rows = q)
  char* const pm = smem + 24576;        // PM[q][key] tr image (rows = q)
  char* const ds = smem + 32768;        // dS[q][key] tr image (rows = q); also read row-wise for dQ
  float* const dbuf = reinterpret_cast<float*>(ds);   // [2][64] partial deltas, before dS is written
  const int ldq = (int)p.ldq, ldk = (int)p.ldk, ldv = (int)p.ldv, lddo = (int)p.lddo;
  const __amdgpu_buffer_rsrc_t qrs = head_rsrc(p.q + (long long)q0 * p.ldq + h * kDh, p.ldq, Lq);
  const __amdgpu_buffer_rsrc_t krs = head_rsrc(p.k + (long long)k0 * p.ldk + h * kDh, p.ldk, Lk);
  const __amdgpu_buffer_rsrc_t vrs = head_rsrc(p.v + (long long)k0 * p.ldv + h * kDh, p.ldv, Lk);
  const __amdgpu_buffer_rsrc_t dors = head_rsrc(p.d_o + (long long)q0 * p.lddo + h * kDh, p.lddo, Lq);
  // The 32 x 32 block (key block kb, query block qb) of S / P / dS exists iff it holds a live key and
  // a live query and is not wholly above the causal diagonal. Dead blocks are never computed, never
  // written to the PM / dS images and never read by the gradient products (wave-uniform branches).
  const int nbq = Lq > 32 ? 2 : 1, nbk = Lk > 32 ? 2 : 1;
  const bool live1 = wi * 32 < Lk && wj * 32 < Lq && !(causal && wi > wj);

  // ---- every global read of the workgroup is issued here -----------------------------------
  bf16x8 fk[4], fq[4], fv[4], fdo[4];
  const int q = wj * 32 + l31;
  float lse2 = 0.f;
  if (live1) {
    const int kv = l31 * ldk * 2 + lhi * 16, qv = l31 * ldq * 2 + lhi * 16;
    const int vv = l31 * ldv * 2 + lhi * 16, dv_ = l31 * lddo * 2 + lhi * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      fk[kk] = load8(krs, kv + kk * 32, wi * 32 * ldk * 2);
      fq[kk] = load8(qrs, qv + kk * 32, wj * 32 * ldq * 2);
      fv[kk] = load8(vrs, vv + kk * 32, wi * 32 * ldv * 2);
      fdo[kk] = load8(dors, dv_ + kk * 32, wj * 32 * lddo * 2);
    }
    if (q < Lq) lse2 = p.lse[(long long)(q0 + q) * p.H + h] * 1.4426950408889634f;
  }
  // the transpose-read images: 32-row blocks that hold a live row (zero fill past the length)
  u32x4 sdo[2], sk[2], sq[2];
  const int srow = tid >> 3, scol = (tid & 7) * 16;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (it < nbq) {
      sdo[it] = __builtin_amdgcn_raw_buffer_load_b128(dors, srow * lddo * 2 + scol, it * 32 * lddo * 2, 0);
      sq[it] = __builtin_amdgcn_raw_buffer_load_b128(qrs, srow * ldq * 2 + scol, it * 32 * ldq * 2, 0);
    }
    if (it < nbk) sk[it] = __builtin_amdgcn_raw_buffer_load_b128(krs, srow * ldk * 2 + scol, it * 32 * ldk * 2, 0);
  }

  // ---- S^T[key][q] = K Q^T and dPM^T[key][q] = V dO^T, this wave's block; P, M, the partial
  //      delta = sum over this wave's 32 keys of P * M * dPM; PM[q][key] to LDS -----------------------
  f32x16 s, dp;
  float delta = 0.f;
  if (live1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[kk], fq[kk], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[kk], fdo[kk], dp, 0, 0, 0);
    }
    const float ik = 1.f / p.keep_prob, sc2 = p.scale * 1.4426950408889634f;
    const uint32_t thr = (uint32_t)(p.keep_prob * 65536.0f);
    const unsigned long long zlane =
        (unsigned long long)((long long)bh * (kL * kL / 4) + q * (kL / 4) + wi * 8 + lhi) * kDropGolden + p.seed;
    // register 4g + e is key wi*32 + 8g + e + 4*lhi: live <=> 8g + e < lim
    const int lim = q < Lq ? min(Lk, causal ? q + 1 : kL) - 4 * lhi - wi * 32 : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t keep = 0xfu;
      if (kDrop) keep = dropout_bits4_z(zlane + attn_drop_step(2 * g), thr);
      float pmv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const float pv = (8 * g + e) < lim ? __builtin_amdgcn_exp2f(s[r] * sc2 - lse2) : 0.f;
        const float mk = kDrop ? (((keep >> e) & 1u) ? ik : 0.f) : 1.f;
        s[r] = pv;                  // P
        dp[r] *= mk;                // dP = dPM * M
        delta += pv * dp[r];
        pmv[e] = pv * mk;
      }
      u32x2 pk;
      pk[0] = pack2bf(pmv[0], pmv[1]);
      pk[1] = pack2bf(pmv[2], pmv[3]);
      *reinterpret_cast<u32x2*>(pm + tr_off(q, wi * 32 + 8 * g + 4 * lhi)) = pk;
    }
    delta = half_sum(delta);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int o = tr_off(it * 32 + srow, (tid & 7) * 8);
    if (it < nbq) {
      *reinterpret_cast<u32x4*>(do_img + o) = sdo[it];
      *reinterpret_cast<u32x4*>(q_img + o) = sq[it];
    }
    if (it < nbk) *reinterpret_cast<u32x4*>(k_img + o) = sk[it];
  }
  if (lhi == 0) dbuf[wi * 64 + q] = delta;       // 0 from a dead block
  __syncthreads();
  delta += dbuf[(wi ^ 1) * 64 + q];      // the other 32 keys of this query (wave (1 - wi, wj))
  __syncthreads();                        // dbuf is overwritten by the dS image next
  if (live1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = s[4 * g + e] * (dp[4 * g + e] - delta);   // dS (w.r.t. the scaled logits)
      u32x2 pk;
      pk[0] = pack2bf(w[0], w[1]);
      pk[1] = pack2bf(w[2], w[3]);
      *reinterpret_cast<u32x2*>(ds + tr_off(q, wi * 32 + 8 * g + 4 * lhi)) = pk;
    }
  }
  __syncthreads();
  // ---- the three gradient products: wave (wi = d block, wj = key / query block); reduction steps of
  //      16 rows that lie past the length or in a dead block are skipped -------------------------------
  f32x16 dv, dq, dk;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dv[e] = 0.f; dq[e] = 0.f; dk[e] = 0.f; }
  const bool krows = wj * 32 < Lk, qrows = wj * 32 < Lq;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (krows && kk * 16 < Lq && !(causal && wj > (kk >> 1))) {       // block (key wj, query kk/2)
      // dV^T[d][key] = sum_q dO[q][d] PM[q][key]: both operands by transpose reads (rows = q)
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(do_img, wi, kk, lane), frag_tr(pm, wj, kk, lane), dv, 0, 0, 0);
      // dK^T[d][key] = scale * sum_q Q[q][d] dS[q][key]: both operands by transpose reads (rows = q)
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(q_img, wi, kk, lane), frag_tr(ds, wj, kk, lane), dk, 0, 0, 0);
    }
    if (qrows && kk * 16 < Lk && !(causal && (kk >> 1) > wj)) {       // block (key kk/2, query wj)
      // dQ^T[d][q] = scale * sum_key K[key][d] dS[q][key]: B operand = 8 consecutive keys of row q
      const bf16x8 dsrow = *reinterpret_cast<const bf16x8*>(ds + tr_off(wj * 32 + l31, kk * 16 + lhi * 8));
      dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(k_img, wi, kk, lane), dsrow, dq, 0, 0, 0);
    }
  }
  if (krows) {
    store_dT_quad(dv, wi, wj, p.dv + (long long)k0 * p.lddv + h * kDh, p.lddv, Lk, 1.f, lane);
    store_dT_quad(dk, wi, wj, p.dk + (long long)k0 * p.lddk + h * kDh, p.lddk, Lk, p.scale, lane);
  }
  if (qrows) store_dT_quad(dq, wi, wj, p.dq + (long long)q0 * p.lddq + h * kDh, p.lddq, Lq, p.scale, lane);
}

}  // namespace os2s

using namespace os2s;

static int attn_check(const int32_t* cu_q, const int32_t* cu_k, int B, int H, int dh, int max_len) {
  if (!cu_q || !cu_k || B < 1 || H < 1) return OS2S_ERR_INVALID_ARG;
  if ((long long)B * H >= (1ll << 30)) return OS2S_ERR_INVALID_ARG;
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
  // 16-byte rows pieces (pointers 16-byte aligned too); row offsets are 32-bit inside the kernels
  OS2S_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  OS2S_REQUIRE(ldq < kMaxLd && ldk < kMaxLd && ldv < kMaxLd && ldo < kMaxLd);
  OS2S_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0);
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
  const size_t smem1 = (size_t)kFwdWaves * 8192;     // one V^T image per wave
  if (keep_prob < 1.f)
    OS2S_LAUNCH(attn_fwd_kernel<true>, dim3(ceil_div((long long)B * H, kFwdWaves)), dim3(kFwdWaves * 64),
                smem1, (hipStream_t)stream, a);
  else
    OS2S_LAUNCH(attn_fwd_kernel<false>, dim3(ceil_div((long long)B * H, kFwdWaves)), dim3(kFwdWaves * 64),
                smem1, (hipStream_t)stream, a);
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
  OS2S_REQUIRE(lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0);
  OS2S_REQUIRE(ldq < kMaxLd && ldk < kMaxLd && ldv < kMaxLd && lddo < kMaxLd);
  OS2S_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk |
                (uintptr_t)dv) % 16 == 0);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.lse = const_cast<float*>(lse); a.cu_q = cu_q; a.cu_k = cu_k;
  a.B = B; a.H = H; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.causal = causal; a.scale = scale;
  a.keep_prob = keep_prob; a.seed = seed; a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
  a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  const size_t smem = (size_t)kBwdLds;      // 40 KB: under the default dynamic limit
  if (keep_prob < 1.f)
    OS2S_LAUNCH(attn_bwd_kernel<true>, dim3((unsigned)((long long)B * H)), dim3(kBwdThreads), smem, (hipStream_t)stream, a);
  else
    OS2S_LAUNCH(attn_bwd_kernel<false>, dim3((unsigned)((long long)B * H)), dim3(kBwdThreads), smem, (hipStream_t)stream, a);
  return OS2S_OK;
}
