// Attention-RNN decoders (teacher-forced training pass and step-wise inference), gfx950.
//
// Reference semantics (tf.contrib.seq2seq.dynamic_decode over an AttentionWrapper cell):
//   * RNNDecoderWithAttention, attention_type gnmt / gnmt_v2
//     (open_seq2seq/decoders/rnn_decoders.py:147-321, parts/rnns/gnmt.py:32-79): the bottom
//     LSTM layer is the attention cell, cell_inputs = concat(inputs, previous attention)
//     (parts/rnns/attention_wrapper.py:1720-1760), query = its output, normalised Bahdanau
//     score (:482-539), attention = context (attention_layer_size=None, :1388-1415).
//   * Tacotron2Decoder (decoders/tacotron2_decoder.py:257-420): AttentionWrapper around a
//     MultiRNNCell of L LSTMCells (output dropout), LocationSensitiveAttention with
//     cumulative alignments (attention_wrapper.py:641-715, 749-878), output = concat(h, ctx).
//
// Structure: everything that does not depend on the previous step is hoisted out of the
// loop by the caller (input projection of all T steps, memory keys, layers above the
// attention cell, output projections, all weight gradients) and runs as large MFMA GEMMs.
// The loop itself is 2-3 launches per step forward and 3-4 backward, all state lives in
// per-step sequence buffers (slot t = input of step t), so a call can run any range of
// steps [t_begin, t_end) — the same entry point serves training and incremental decoding:
//   cell kernel   : gates = gx[t] + cat[t] . Wcat^T (split-K MFMA tile, rnn_tile.hpp),
//                   LSTM cell in the epilogue, h -> cat[t+1] (recurrence), dropped h -> y
//   attention     : one workgroup per sample: query projection, (location features),
//                   score, masked softmax, context; context -> ctx[t] and cat0[t+1]
//   backward      : d(attention input) GEMM, attention backward per sample (recomputes the
//                   tanh terms; accumulates dkeys, per-sample parameter partials), cell
//                   backward (two fused transposed GEMMs + gate derivatives).
#include <algorithm>
#include <cstdlib>
#include "os2s_common.hpp"
#include "rnn_tile.hpp"

namespace os2s {

constexpr int kAdWaves = 8;
constexpr int kAttnThreads = 512;

__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f / (1.f + __expf(2.f * x)); }

// ------------------------------------------------------------------ cell forward
struct AdCellFwd {
  int B, T, H, Kc, t;
  float forget_bias;
  const int32_t* lens;
  const bf16_t* cat;   // [B, T+1, Kc]
  const bf16_t* w;     // [4H, Kc]
  const uint8_t* w8;   // e4m3 copy of w [4H, Kc] (or null) ...
  const float* w8_scale;  // ... and its per-row scales [4H]
  const float* bias;   // [4H] or null
  const bf16_t* gx;    // [B, T, 4H] or null
  float* c_seq;        // [B, T, H]
  bf16_t* gates;       // [B, T, 4H] or null
  bf16_t* h_next;      // cat + column offset of the recurrent part
  bf16_t* y;           // dropped output, row (b,t) at y + b*y_bs + t*y_ts
  long long y_bs, y_ts;
  float out_keep;
  unsigned long long out_seed;
};

// One workgroup = 8 hidden units x 32 samples: the 32 tile rows are (gate, unit) pairs
// (row = 8*gate + unit), so H/8 workgroups stream disjoint 32-row slices of the weights and
// after the split-K reduction a lane owns all four gates of one (unit, sample).
// (20 slices per wave so that the Tacotron2 decoder LSTM, Kc = 2560, needs one round of loads instead
// of a second, mostly empty one: 46.1 vs 46.4 us per forward step — not kept.)
__global__ __launch_bounds__(64 * kAdWaves) void ad_cell_fwd_kernel(AdCellFwd p) {
  __shared__ float red[kAdWaves * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
  const int H = p.H;
  // the epilogue operands of this lane's (unit, sample) are fetched BEFORE the weight stream so that
  // their round trip overlaps it instead of following the reduction (as rnn.hip does)
  const int eb = b0 + l31, ej = j0 + wave + 4 * lhi;
  const bool elive = wave < 4 && eb < p.B && ej < H && !(p.lens && p.t >= p.lens[eb]);
  float e_gx[4] = {0.f, 0.f, 0.f, 0.f}, e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_sc[4] = {1.f, 1.f, 1.f, 1.f};
  float e_cprev = 0.f;
  if (elive) {
    const long long erow = (long long)eb * p.T + p.t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (p.w8) e_sc[g] = p.w8_scale[g * H + ej];
      if (p.gx) e_gx[g] = bf2f(p.gx[erow * (4 * H) + (long long)g * H + ej]);
      if (p.bias) e_bias[g] = p.bias[g * H + ej];
    }
    if (p.t > 0) e_cprev = p.c_seq[(erow - 1) * H + ej];
  }
  f32x16 accw;
#pragma unroll
  for (int e = 0; e < 16; ++e) accw[e] = 0.f;
  {
    const int g = l31 >> 3, uu = min(j0 + (l31 & 7), H - 1);
    const int brow = b0 + l31;
    const bf16_t* irow = brow < p.B ? p.cat + ((long long)brow * (p.T + 1) + p.t) * p.Kc : nullptr;
    if (p.w8) {
      tile_gemm_prefetch_w8<kAdWaves, 16>(p.w8 + ((long long)g * H + uu) * p.Kc, irow, p.Kc, accw, p.w8, p.cat);
    } else {
      const bf16_t* wrow = p.w + ((long long)g * H + uu) * p.Kc;
      tile_gemm_prefetch<kAdWaves, 16>(wrow, irow, p.Kc, accw, p.w);
    }
  }
  float pre[4];
  tile_reduce_units<kAdWaves>(accw, red, pre);
  if (wave >= 4) return;
  const int b = b0 + l31;
  if (b >= p.B) return;
  if (p.lens && p.t >= p.lens[b]) return;   // finished sample: buffers stay zero
  const int j = j0 + wave + 4 * lhi;
  if (j >= H) return;
  const long long row = (long long)b * p.T + p.t;
#pragma unroll
  for (int g = 0; g < 4; ++g) pre[g] = pre[g] * e_sc[g] + e_gx[g] + e_bias[g];   // (scale: e4m3 weight rows)
  const float cprev = e_cprev;
  const float ig = sigmoidf_(pre[0]), gg = tanhf(pre[1]);
  const float fg = sigmoidf_(pre[2] + p.forget_bias), og = sigmoidf_(pre[3]);
  const float cn = cprev * fg + ig * gg;
  float hn = tanhf(cn) * og;
  p.c_seq[row * H + j] = cn;
  if (p.gates) {
    bf16_t* gp = p.gates + row * (4 * H) + j;
    gp[0] = f2bf(ig); gp[H] = f2bf(fg); gp[2 * H] = f2bf(gg); gp[3 * H] = f2bf(og);
  }
  p.h_next[((long long)b * (p.T + 1) + p.t + 1) * p.Kc + j] = f2bf(hn);
  if (p.out_keep < 1.f) {
    const unsigned long long idx = (unsigned long long)row * H + j;
    const uint32_t bits = dropout_bits8(p.out_seed, idx >> 3, p.out_keep);
    hn = ((bits >> (j & 7)) & 1u) ? hn / p.out_keep : 0.f;
  }
  p.y[(long long)b * p.y_bs + (long long)p.t * p.y_ts + j] = f2bf(hn);
}

// ------------------------------------------------------------------ attention (shared)
// Location-sensitive attention: Conv1D(K taps -> F filters, bias) followed by the bias-free
// dense F -> U has no non-linearity in between, so per call the two are folded into ONE
// filter  Wck[k,u] = sum_f conv_w[k,f] dense_w[f,u],  bd[u] = sum_f conv_b[f] dense_w[f,u]:
//   location[s,u] = sum_k cum[s + k - padl] Wck[k,u] + bd[u]
// (F x fewer multiply-adds in the loop, Wck lives in registers: lane owns 2 units). The
// gradient w.r.t. conv_w / conv_b / dense_w is recovered from dWck, d(bd) after the loop.
struct AdAttn {
  int B, T, S, H, M, U, t, mode, use_bias, loc_k, Kc0, last;
  const int32_t* src_len;
  const int32_t* tgt_len;
  const bf16_t* yq;      // query input rows: yq + b*yq_bs + t*yq_ts
  long long yq_bs, yq_ts;
  const bf16_t* wq;      // [U, H]
  const bf16_t* keys;    // [B, S, U]
  const bf16_t* values;  // [B, S, M]
  const float* v; const float* g; const float* bias;
  const float* wck;      // [K, U] folded location filter followed by bd [U]   (mode 2)
  float* cum_seq;        // [B, T+1, S]
  float* align_seq;      // [B, T, S]
  float* q_seq;          // [B, T, U]
  bf16_t* ctx;           // raw context rows: ctx + b*ctx_bs + t*ctx_ts
  long long ctx_bs, ctx_ts;
  bf16_t* cat0;          // [B, T+1, Kc0]
  float attn_in_keep;
  unsigned long long attn_in_seed;
  // backward only
  const bf16_t* dctx_ext; long long dctx_bs, dctx_ts;   // external gradient of ctx rows or null
  const float* dattn;    // [B, M] gradient w.r.t. the attention part of cat0[t+1] (null when last)
  bf16_t* dctx_seq;      // [B, T, M] total context gradient (for the dvalues pass)
  float* dcum;           // [B, S] carry (mode 2)
  bf16_t* dpre_seq;      // [B, T, S, U] score pre-activation gradients (dkeys = sum over T)
  bf16_t* dq_seq;        // [B, T, U]
  float* dhq;            // [B, H]
  float* dnv_acc;        // [B, U]
  float* dbd_acc;        // [B, U]      (mode 2)
  float* dwck_acc;       // [B, K, U]   (mode 2)
};

constexpr int kAttnWaves = kAttnThreads / 64;

// development aid (build with -DOS2S_ATTN_PHASE_TIMERS): phase timestamps (s_memtime) of
// workgroup 0, read by os2s_debug_attn_phases / tools/bench_attn_decoder.py
#ifdef OS2S_ATTN_PHASE_TIMERS
__device__ long long g_attn_dbg[2][16];
#define AD_TICK(k, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_attn_dbg[k][i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define AD_TICK(k, i) do { } while (0)
#endif

__device__ __forceinline__ float block_sum(float x, float* red) {
  x = wave_sum_dpp(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kAttnWaves; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ float block_max(float x, float* red) {
  x = wave_max_dpp(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int w = 1; w < kAttnWaves; ++w) s = fmaxf(s, red[w]);
  return s;
}

// LDS carve-up (floats) shared by the forward and backward attention kernels
struct AttnLds {
  float *hq, *q, *nv, *bs, *e, *red, *cum, *scratch;
  uint32_t* keys;   // [S][U/2] bf16 pairs of this sample's keys (staged once per launch)
};
constexpr int kCtxSplitMax = 8;
constexpr int kLocKMax = 32;   // location filter taps held in registers (U == 128: 2 units / lane)
__host__ __device__ inline size_t attn_lds_floats(int H, int M, int U, int S, int mode, int K, bool bwd) {
  size_t n = (size_t)H + 3 * U + S + 64 + (size_t)S * U / 2;
  if (mode == 2) n += (size_t)(S + kLocKMax);
  size_t scratch = (size_t)kCtxSplitMax * M;     // forward: context partials
  if (bwd)   // dctx, dalign, de, dcum, dq/dnv partials, dWck
    scratch = (size_t)M + 3 * S + 2 * (size_t)kAttnWaves * U + (mode == 2 ? (size_t)K * U : 0);
  return n + scratch + 64;
}
__device__ __forceinline__ AttnLds attn_lds_carve(float* base, int H, int U, int S, int mode, int K) {
  AttnLds l;
  l.hq = base; base += H;
  l.q = base; base += U;
  l.nv = base; base += U;
  l.bs = base; base += U;
  l.e = base; base += S;
  l.red = base; base += 64;
  l.cum = nullptr;
  if (mode == 2) { l.cum = base; base += S + kLocKMax; }
  l.keys = reinterpret_cast<uint32_t*>(base); base += (size_t)S * U / 2;
  l.scratch = base;
  return l;
}

// score parameters -> LDS: nv (normalised v for mode 1), bs (bias (+ folded location bias))
__device__ __forceinline__ void attn_load_score_params(const AdAttn& p, const AttnLds& l) {
  const int tid = threadIdx.x, U = p.U;
  float ss = 0.f;
  for (int u = tid; u < U; u += kAttnThreads) {
    const float v = p.v[u];
    l.nv[u] = v;
    ss += v * v;
    float b = ((p.mode == 1 || (p.mode == 2 && p.use_bias)) && p.bias) ? p.bias[u] : 0.f;
    if (p.mode == 2) b += p.wck[p.loc_k * U + u];
    l.bs[u] = b;
  }
  if (p.mode == 1) {
    const float tot = block_sum(ss, l.red);
    const float sc = p.g[0] * rsqrtf(tot);
    for (int u = tid; u < U; u += kAttnThreads) l.nv[u] *= sc;
  }
  __syncthreads();
}

// this sample's keys -> LDS with 16-byte coalesced loads (all in flight at once; the latency
// hides behind the query projection)
__device__ __forceinline__ void attn_stage_keys(const AdAttn& p, const AttnLds& l, int b, int slen) {
  const int n16 = slen * p.U / 8;
  const u32x4* src = reinterpret_cast<const u32x4*>(p.keys + (long long)b * p.S * p.U);
  u32x4* dst = reinterpret_cast<u32x4*>(l.keys);
  for (int i = threadIdx.x; i < n16; i += kAttnThreads) dst[i] = src[i];
}

// cumulative alignments (state BEFORE step t), zero-padded for the SAME convolution
__device__ __forceinline__ void attn_load_cum(const AdAttn& p, const AttnLds& l, int b) {
  const int S = p.S, K = p.loc_k, padl = (K - 1) / 2;
  const float* cum = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
  for (int i = threadIdx.x; i < S + kLocKMax; i += kAttnThreads) {
    const int s = i - padl;
    l.cum[i] = (s >= 0 && s < S) ? cum[s] : 0.f;
  }
}

// pre-activation of the score for (s, unit pair u, u+1) without the location term
__device__ __forceinline__ void attn_pre2(const AdAttn& p, const AttnLds& l, int b, int s, int u,
                                          float& x0, float& x1) {
  const uint32_t kv = l.keys[(s * p.U + u) >> 1];
  x0 = bflo(kv) + l.q[u] + l.bs[u];
  x1 = bfhi(kv) + l.q[u + 1] + l.bs[u + 1];
}

// ------------------------------------------------------------------ attention forward
__global__ __launch_bounds__(kAttnThreads) void ad_attn_fwd_kernel(AdAttn p) {
  extern __shared__ float lds_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int H = p.H, M = p.M, U = p.U, S = p.S, K = p.loc_k;
  const AttnLds l = attn_lds_carve(lds_raw, H, U, S, p.mode, K);
  const int slen = min(max(p.src_len[b], 0), S);
  AD_TICK(0, 0);
  // query input
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  for (int h8 = tid; h8 < H / 8; h8 += kAttnThreads) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(yq + h8 * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { l.hq[h8 * 8 + 2 * e] = bflo(v[e]); l.hq[h8 * 8 + 2 * e + 1] = bfhi(v[e]); }
  }
  attn_stage_keys(p, l, b, slen);
  if (p.mode == 2) attn_load_cum(p, l, b);
  attn_load_score_params(p, l);   // ends with a barrier
  AD_TICK(0, 1);
  // q[u] = hq . Wq[u, :]  — 8 units per wave per batch, all their loads in flight together
  constexpr int UB = 8;
  for (int u0 = wave * UB; u0 < U; u0 += UB * kAttnWaves) {
    float part[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) part[i] = 0.f;
    for (int h = lane * 8; h < H; h += 1024) {
      u32x4 wv[UB][2];
#pragma unroll
      for (int i = 0; i < UB; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          // unconditional (clamped) loads: a load under a branch is waited for at the join,
          // which would serialise the round trips
          wv[i][hh] = *reinterpret_cast<const u32x4*>(p.wq + (long long)min(u0 + i, U - 1) * H +
                                                      min(h + 512 * hh, H - 8));
#pragma unroll
      for (int i = 0; i < UB; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          if (h + 512 * hh < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              part[i] += bflo(wv[i][hh][e]) * l.hq[h + 512 * hh + 2 * e] +
                         bfhi(wv[i][hh][e]) * l.hq[h + 512 * hh + 2 * e + 1];
          }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const float s = wave_sum_dpp(part[i]);
      if (lane == 0 && u0 + i < U) {
        l.q[u0 + i] = s;
        p.q_seq[((long long)b * p.T + p.t) * U + u0 + i] = s;
      }
    }
  }
  __syncthreads();
  AD_TICK(0, 2);
  // scores: one wave per source position, a lane owns unit pairs
  if (p.mode == 2) {
    float wk[kLocKMax][2];
#pragma unroll
    for (int k = 0; k < kLocKMax; ++k) {
      wk[k][0] = wk[k][1] = 0.f;
      if (k < K) {
        const f32x2 w = *reinterpret_cast<const f32x2*>(p.wck + (long long)k * U + 2 * lane);
        wk[k][0] = w[0]; wk[k][1] = w[1];
      }
    }
    // a wave owns a contiguous range of source positions and slides a 32-entry register
    // window over the padded cumulative alignments: slot (j + k) & 31 holds cum[s + k] at the
    // j-th step of a 32-step block, so each step costs ONE new LDS read
    const int chunk = (slen + kAttnWaves - 1) / kAttnWaves;
    const int c0 = wave * chunk, c1 = min(c0 + chunk, slen);
    const float qb0 = l.q[2 * lane] + l.bs[2 * lane], qb1 = l.q[2 * lane + 1] + l.bs[2 * lane + 1];
    const float nv0 = l.nv[2 * lane], nv1 = l.nv[2 * lane + 1];
    if (c0 < c1) {
      float cw[kLocKMax];
#pragma unroll
      for (int k = 0; k < kLocKMax; ++k) cw[k] = l.cum[c0 + k];
      for (int s0 = c0; s0 < c1; s0 += kLocKMax) {
#pragma unroll
        for (int j = 0; j < kLocKMax; ++j) {
          const int s = s0 + j;
          if (s < c1) {
            const uint32_t kv = l.keys[(s * U) / 2 + lane];
            float x0 = bflo(kv) + qb0, x1 = bfhi(kv) + qb1;
#pragma unroll
            for (int k = 0; k < kLocKMax; ++k) {
              x0 += cw[(j + k) & 31] * wk[k][0];
              x1 += cw[(j + k) & 31] * wk[k][1];
            }
            float part = nv0 * tanh_fast(x0) + nv1 * tanh_fast(x1);
            part = wave_sum_dpp(part);
            if (lane == 0) l.e[s] = part;
            cw[j & 31] = s < S ? l.cum[s + kLocKMax] : 0.f;
          }
        }
      }
    }
  } else {
    for (int s = wave; s < slen; s += kAttnWaves) {
      float part = 0.f;
      for (int u = 2 * lane; u < U; u += 128) {
        if (p.mode == 3) {        // Luong: score = keys . query (query = cell output, w_q = I)
          const uint32_t kv = l.keys[(s * U + u) >> 1];
          part += bflo(kv) * l.q[u] + bfhi(kv) * l.q[u + 1];
        } else {
          float x0, x1;
          attn_pre2(p, l, b, s, u, x0, x1);
          part += l.nv[u] * tanh_fast(x0) + l.nv[u + 1] * tanh_fast(x1);
        }
      }
      part = wave_sum_dpp(part);
      if (lane == 0) l.e[s] = part;
    }
  }
  __syncthreads();
  AD_TICK(0, 3);
  // masked softmax over s < slen
  float mx = -INFINITY;
  for (int s = tid; s < slen; s += kAttnThreads) mx = fmaxf(mx, l.e[s]);
  mx = block_max(mx, l.red);
  float sum = 0.f;
  for (int s = tid; s < slen; s += kAttnThreads) {
    const float ex = __expf(l.e[s] - mx);
    l.e[s] = ex;
    sum += ex;
  }
  sum = block_sum(sum, l.red);
  const float inv = slen > 0 ? 1.f / sum : 0.f;
  float* al_out = p.align_seq + ((long long)b * p.T + p.t) * S;
  for (int s = tid; s < S; s += kAttnThreads) {
    const float a = s < slen ? l.e[s] * inv : 0.f;
    l.e[s] = a;
    al_out[s] = a;
    if (p.mode == 2) {
      const long long ci = ((long long)b * (p.T + 1) + p.t) * S + s;
      p.cum_seq[ci + S] = p.cum_seq[ci] + a;
    }
  }
  __syncthreads();
  AD_TICK(0, 4);
  // context = sum_s align[s] * values[b, s, :]: 8-column groups x nsplit slices of s
  const int CG = min(M / 8, kAttnThreads);
  const int nsplit = min(kAttnThreads / CG, kCtxSplitMax);
  float* part = l.scratch;   // [nsplit][M]
  for (int c0 = 0; c0 < M / 8; c0 += CG) {
    const int cg = c0 + tid % CG, sp = tid / CG;
    if (sp < nsplit && cg < M / 8) {
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bf16_t* vp = p.values + (long long)b * S * M + cg * 8;
#pragma unroll 16
      for (int s = sp; s < slen; s += nsplit) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(vp + (long long)s * M);
        const float a = l.e[s];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a8[2 * e] += a * bflo(v[e]); a8[2 * e + 1] += a * bfhi(v[e]); }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) part[sp * M + cg * 8 + e] = a8[e];
    }
  }
  __syncthreads();
  AD_TICK(0, 5);
  bf16_t* ctx = p.ctx + (long long)b * p.ctx_bs + (long long)p.t * p.ctx_ts;
  bf16_t* cat = p.cat0 + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc0;
  for (int m8 = tid; m8 < M / 8; m8 += kAttnThreads) {
    float c8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c8[e] = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) c8[e] += part[sp * M + m8 * 8 + e];
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(c8[2 * e], c8[2 * e + 1]);
    *reinterpret_cast<u32x4*>(ctx + m8 * 8) = o;
    if (p.attn_in_keep < 1.f) {
      const unsigned long long idx8 = (((unsigned long long)b * (p.T + 1) + p.t + 1) * M) / 8 + m8;
      const uint32_t bits = dropout_bits8(p.attn_in_seed, idx8, p.attn_in_keep);
      const float ik = 1.f / p.attn_in_keep;
#pragma unroll
      for (int e = 0; e < 8; ++e) c8[e] = ((bits >> e) & 1u) ? c8[e] * ik : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(c8[2 * e], c8[2 * e + 1]);
    }
    *reinterpret_cast<u32x4*>(cat + m8 * 8) = o;
  }
  AD_TICK(0, 6);
}

// ------------------------------------------------------------------ d(attention input) GEMM
struct AdDattn {
  int B, M, K;
  const bf16_t* wT;     // [M rows, K] (first M rows of Wcat0^T)
  const bf16_t* dg;     // rows: dg + b*ld
  long long ld;
  float* out;           // [B, M]
};
// Backward GEMMs: 32-row tiles, the reduction split over 16 waves so that every load of a
// wave (one 16-wide k-slice per 256 of K) is in flight at once.
constexpr int kBwdWaves = 16;

// sum the partial tiles of the 16 waves; wave q < 4 receives rows 8q + 4*(lane>>5) + e
// (e < 4) of the tile: out[e] += sum_w acc_w[4q + e]. Two passes of 8 registers keep the
// LDS buffer at 32 KB. `red`: kBwdWaves*8*64 floats.
__device__ __forceinline__ void tile_reduce_quarters16(const f32x16& acc, float* red, float (&out)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int r = 0; r < 8; ++r) red[(wave * 8 + r) * 64 + lane] = acc[ps * 8 + r];
    __syncthreads();
    if ((wave >> 1) == ps && wave < 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kBwdWaves; ++w2) s += red[(w2 * 8 + 4 * (wave & 1) + e) * 64 + lane];
        out[e] += s;
      }
    }
    __syncthreads();
  }
}

// 8 output rows per workgroup (rows 8..31 of the MFMA tile are padding): M = 512 attention
// inputs are 64 workgroups instead of 16.
__global__ __launch_bounds__(64 * kBwdWaves) void ad_dattn_kernel(AdDattn p) {
  __shared__ float red[kBwdWaves * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
  f32x16 accw;
#pragma unroll
  for (int e = 0; e < 16; ++e) accw[e] = 0.f;
  const bf16_t* wrow = (l31 < 8 && j0 + l31 < p.M) ? p.wT + (long long)(j0 + l31) * p.K : nullptr;
  const int brow = b0 + l31;
  const bf16_t* irow = brow < p.B ? p.dg + (long long)brow * p.ld : nullptr;
  tile_gemm_prefetch<kBwdWaves, 8>(wrow, irow, p.K, accw, p.wT);
  float acc[4];
  tile_reduce_rows8<kBwdWaves>(accw, red, acc);
  if (wave >= 1) return;
  const int b = b0 + l31, j = j0 + 4 * lhi;
  if (b >= p.B || j >= p.M) return;
  f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<f32x4*>(p.out + (long long)b * p.M + j) = o;
}

// The same product cut kDaSplit ways along the reduction, one FULL 32-row tile per workgroup and ONE
// round of loads per wave; the pieces of a tile meet through a ticket exactly as in
// ad_cell_bwd_split_kernel (deterministic: the last arriver sums the slabs in piece order).
// M = 512, K = 4096: 16 x 8 = 128 workgroups of one round instead of 64 of two.
constexpr int kDaSplit = 8;
constexpr int kDaWaves = 8;
constexpr int kDaSlices = 4;
constexpr int kDaSlabFloats = 16 * 64;

struct AdDattnSplit {
  AdDattn g;
  float* part;   // [M/32][B/32][kDaSplit][kDaSlabFloats]
  int* ticket;   // [M/32][B/32], zero before and after every launch
};

__global__ __launch_bounds__(64 * kDaWaves) void ad_dattn_split_kernel(AdDattnSplit ps) {
  const AdDattn& p = ps.g;
  __shared__ float red[kDaWaves * 16 * 64];       // 32 KB
  __shared__ int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = blockIdx.x, kq = blockIdx.y, bb = blockIdx.z;
  const int j0 = mb * 32, b0 = bb * 32;
  const bool vrow = j0 + l31 < p.M;
  const int brow = b0 + l31;
  const bool vcol = brow < p.B;
  const bf16_t* const wr = p.wT + (long long)(vrow ? j0 + l31 : 0) * p.K;
  const bf16_t* const gr = p.dg + (long long)(vcol ? brow : 0) * p.ld;
  const int nit = (p.K + 15) >> 4, nper = (nit + kDaSplit - 1) / kDaSplit;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int base = 0; base < nper; base += kDaWaves * kDaSlices) {
    u32x4 wa[kDaSlices], ga[kDaSlices];
#pragma unroll
    for (int i = 0; i < kDaSlices; ++i) {
      const int s = kq * nper + base + wave + kDaWaves * i;
      const int ko = min(s * 16 + lhi * 8, p.K - 8);
      wa[i] = *reinterpret_cast<const u32x4*>(wr + ko);
      ga[i] = *reinterpret_cast<const u32x4*>(gr + ko);
    }
#pragma unroll
    for (int i = 0; i < kDaSlices; ++i) {
      const int sl = base + wave + kDaWaves * i, s = kq * nper + sl;
      const bool kv = sl < nper && s < nit && s * 16 + lhi * 8 < p.K;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (kv && vrow) ? wa[i] : zero),
                                                    __builtin_bit_cast(bf16x8, (kv && vcol) ? ga[i] : zero), acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float* const slab0 = ps.part + ((size_t)(mb * gridDim.z + bb) * kDaSplit) * kDaSlabFloats;
#pragma unroll
  for (int q = 0; q < kDaSlabFloats / (64 * kDaWaves); ++q) {
    const int o = tid + 64 * kDaWaves * q;
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kDaWaves; ++w2) s += red[w2 * 16 * 64 + o];
    slab0[(size_t)kq * kDaSlabFloats + o] = s;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const ticket = ps.ticket + mb * gridDim.z + bb;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != kDaSplit - 1) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // wave g < 4: rows 8g + 4*lhi + e of the tile (accumulator registers 4g + e), column = sample l31
  if (wave >= 4 || !vcol) return;
  const int j = j0 + 8 * wave + 4 * lhi;
  if (j >= p.M) return;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  for (int pc = 0; pc < kDaSplit; ++pc) {
    const float* const sl = slab0 + (size_t)pc * kDaSlabFloats;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += sl[(4 * wave + e) * 64 + lane];
  }
  *reinterpret_cast<f32x4*>(p.out + (long long)brow * p.M + j) = o;
}

// ------------------------------------------------------------------ attention backward
// sum over the 64 lanes of 32 per-lane values; lane L returns the total of value (L >> 1)
// same for 16 values: lane L returns the total of value (L >> 2)
__device__ __forceinline__ float wave_reduce16(float (&v)[16]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int off = 32 >> st, half = 8 >> st;
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float send = hi ? v[i] : v[i + half];
      const float keep = hi ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  float r = v[0] + __shfl_xor(v[0], 2, 64);
  return r + __shfl_xor(r, 1, 64);
}

__device__ __forceinline__ float wave_reduce32(float (&v)[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int st = 0; st < 5; ++st) {
    const int off = 32 >> st, half = 16 >> st;
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float send = hi ? v[i] : v[i + half];
      const float keep = hi ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 1, 64);
}

template <bool LOC>
__global__ __launch_bounds__(kAttnThreads) void ad_attn_bwd_kernel(AdAttn p) {
  extern __shared__ float lds_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int H = p.H, M = p.M, U = p.U, S = p.S, K = p.loc_k;
  const AttnLds l = attn_lds_carve(lds_raw, H, U, S, p.mode, K);
  float* dctx = l.scratch;            // [M]
  float* dal = dctx + M;              // [S]
  float* de = dal + S;                // [S]
  float* dcum_l = de + S;             // [S]   (mode 2)
  float* dqp = dcum_l + S;            // [waves][U]
  float* dnvp = dqp + kAttnWaves * U; // [waves][U]
  float* dwk_l = dnvp + kAttnWaves * U;   // [K][U]  (mode 2)
  const int slen = min(max(p.src_len[b], 0), S);
  const long long row = (long long)b * p.T + p.t;
  AD_TICK(1, 0);
  // total context gradient
  {
    const bf16_t* ext = p.dctx_ext ? p.dctx_ext + (long long)b * p.dctx_bs + (long long)p.t * p.dctx_ts : nullptr;
    for (int m8 = tid; m8 < M / 8; m8 += kAttnThreads) {
      float c8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (ext) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ext + m8 * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { c8[2 * e] = bflo(v[e]); c8[2 * e + 1] = bfhi(v[e]); }
      }
      if (!p.last) {
        const float* da = p.dattn + (long long)b * M + m8 * 8;
        uint32_t bits = 0xffu;
        float ik = 1.f;
        if (p.attn_in_keep < 1.f) {
          const unsigned long long idx8 = (((unsigned long long)b * (p.T + 1) + p.t + 1) * M) / 8 + m8;
          bits = dropout_bits8(p.attn_in_seed, idx8, p.attn_in_keep);
          ik = 1.f / p.attn_in_keep;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) if ((bits >> e) & 1u) c8[e] += da[e] * ik;
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(c8[2 * e], c8[2 * e + 1]);
      *reinterpret_cast<u32x4*>(p.dctx_seq + row * M + m8 * 8) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) dctx[m8 * 8 + e] = c8[e];
    }
  }
  AD_TICK(1, 1);
  attn_stage_keys(p, l, b, slen);
  for (int u = tid; u < U; u += kAttnThreads) l.q[u] = p.q_seq[row * U + u];
  for (int s = tid; s < S; s += kAttnThreads) l.e[s] = p.align_seq[row * S + s];
  if (p.mode == 2) {
    attn_load_cum(p, l, b);
    for (int s = tid; s < S; s += kAttnThreads) dcum_l[s] = p.dcum[(long long)b * S + s];
    for (int i = tid; i < K * U; i += kAttnThreads) dwk_l[i] = 0.f;
  }
  attn_load_score_params(p, l);   // barrier inside
  AD_TICK(1, 2);
  // dalign[s] = dctx . values[b,s,:] (+ carry of the cumulative-alignment state)
#pragma unroll 8
  for (int s = wave; s < slen; s += kAttnWaves) {
    float part = 0.f;
    const bf16_t* vp = p.values + ((long long)b * S + s) * M;
    for (int m = lane * 8; m < M; m += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(vp + m);
#pragma unroll
      for (int e = 0; e < 4; ++e) part += bflo(v[e]) * dctx[m + 2 * e] + bfhi(v[e]) * dctx[m + 2 * e + 1];
    }
    part = wave_sum_dpp(part);
    if (lane == 0) dal[s] = part + (p.mode == 2 ? dcum_l[s] : 0.f);
  }
  __syncthreads();
  AD_TICK(1, 3);
  float dot = 0.f;
  for (int s = tid; s < slen; s += kAttnThreads) dot += l.e[s] * dal[s];
  dot = block_sum(dot, l.red);
  for (int s = tid; s < S; s += kAttnThreads) de[s] = s < slen ? l.e[s] * (dal[s] - dot) : 0.f;
  __syncthreads();
  AD_TICK(1, 4);
  // through the score: dpre[s,u] = de[s] * nv[u] * (1 - tanh^2)
  if constexpr (LOC) {
    // U == 128: a lane owns ONE unit; a pair of waves covers the 128 units of a contiguous
    // range of source positions. Three 32-entry register windows slide with s (slot
    // (j + k) & 31 at the j-th step of a 32-step block): cw = padded cumulative alignments,
    // gw = partial state gradient c_u[s - padl + k] = sum_k' dpre[s',u] Wck[k',u] (a lane-local
    // convolution along s; the sum over units happens once per emitted position), so the
    // loop needs one LDS read and one wave reduction per position.
    const int padl = (K - 1) / 2;
    const int u = (wave & 1) * 64 + lane, grp = wave >> 1, ngrp = kAttnWaves / 2;
    const int chunk = (slen + ngrp - 1) / ngrp;
    const int c0 = grp * chunk, c1 = min(c0 + chunk, slen);
    float dq0 = 0.f, dn0 = 0.f;
    if (c0 < c1) {
      float wk[kLocKMax], dwk[kLocKMax], cw[kLocKMax], gw[kLocKMax];
#pragma unroll
      for (int k = 0; k < kLocKMax; ++k) {
        dwk[k] = 0.f;
        gw[k] = 0.f;
        wk[k] = k < K ? p.wck[(long long)k * U + u] : 0.f;
        cw[k] = l.cum[c0 + k];
      }
      const float nv0 = l.nv[u], qb = l.q[u] + l.bs[u];
      const bf16_t* kl = reinterpret_cast<const bf16_t*>(l.keys) + u;
      bf16_t* dps = p.dpre_seq + (row * S) * U + u;   // dpre of this step: dkeys = sum over steps
      // run kLocKMax steps past the end so that every window slot is emitted
      for (int s0 = c0; s0 < c1 + kLocKMax; s0 += kLocKMax) {
#pragma unroll
        for (int j = 0; j < kLocKMax; ++j) {
          const int s = s0 + j;
          if (s < c1) {
            float x0 = bf2f(kl[s * U]) + qb;
#pragma unroll
            for (int k = 0; k < kLocKMax; ++k) x0 += cw[(j + k) & 31] * wk[k];
            const float t0 = tanh_fast(x0);
            const float des = de[s];
            const float d0 = des * nv0 * (1.f - t0 * t0);
            dq0 += d0;
            dn0 += des * t0;
            dps[(long long)s * U] = f2bf(d0);
#pragma unroll
            for (int k = 0; k < kLocKMax; ++k) {
              dwk[k] += cw[(j + k) & 31] * d0;
              gw[(j + k) & 31] += d0 * wk[k];
            }
          }
          if (s < c1 + kLocKMax) {
            // slot j is complete: it is the state gradient at position s - padl
            const float gsum = wave_sum_dpp(gw[j & 31]);
            gw[j & 31] = 0.f;
            const int s2 = s - padl;
            if (lane == 0 && s2 >= 0 && s2 < S) atomicAdd(&dcum_l[s2], gsum);
            cw[j & 31] = s < S ? l.cum[s + kLocKMax] : 0.f;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kLocKMax; ++k)
        if (k < K) atomicAdd(&dwk_l[k * U + u], dwk[k]);
    }
    dqp[wave * U + u] = dq0;
    dnvp[wave * U + u] = dn0;
    dqp[wave * U + (u ^ 64)] = 0.f;
    dnvp[wave * U + (u ^ 64)] = 0.f;
  } else {
    constexpr int UP = 4;   // unit pairs per lane (U <= 512)
    float dq_acc[2 * UP], dnv_acc[2 * UP];
#pragma unroll
    for (int i = 0; i < 2 * UP; ++i) { dq_acc[i] = 0.f; dnv_acc[i] = 0.f; }
    for (int s = wave; s < slen; s += kAttnWaves) {
      const float des = de[s];
#pragma unroll
      for (int i = 0; i < UP; ++i) {
        const int u = 2 * lane + 128 * i;
        if (u < U) {
          if (p.mode == 3) {      // Luong: d score / d key = query, d score / d query = key
            const uint32_t kv = l.keys[(s * U + u) >> 1];
            dq_acc[2 * i] += des * bflo(kv);
            dq_acc[2 * i + 1] += des * bfhi(kv);
            *reinterpret_cast<uint32_t*>(p.dpre_seq + (row * S + s) * U + u) =
                pack2bf(des * l.q[u], des * l.q[u + 1]);
          } else {
            float x0, x1;
            attn_pre2(p, l, b, s, u, x0, x1);
            const float t0 = tanh_fast(x0), t1 = tanh_fast(x1);
            const float d0 = des * l.nv[u] * (1.f - t0 * t0), d1 = des * l.nv[u + 1] * (1.f - t1 * t1);
            dq_acc[2 * i] += d0;
            dq_acc[2 * i + 1] += d1;
            dnv_acc[2 * i] += des * t0;
            dnv_acc[2 * i + 1] += des * t1;
            *reinterpret_cast<uint32_t*>(p.dpre_seq + (row * S + s) * U + u) = pack2bf(d0, d1);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UP; ++i) {
      const int u = 2 * lane + 128 * i;
      if (u < U) {
        dqp[wave * U + u] = dq_acc[2 * i]; dqp[wave * U + u + 1] = dq_acc[2 * i + 1];
        dnvp[wave * U + u] = dnv_acc[2 * i]; dnvp[wave * U + u + 1] = dnv_acc[2 * i + 1];
      }
    }
  }
  __syncthreads();
  AD_TICK(1, 5);
  for (int u = tid; u < U; u += kAttnThreads) {
    float dq = 0.f, dn = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) { dq += dqp[w * U + u]; dn += dnvp[w * U + u]; }
    p.dq_seq[row * U + u] = f2bf(dq);
    p.dnv_acc[(long long)b * U + u] += dn;
    if (p.mode == 2) p.dbd_acc[(long long)b * U + u] += dq;
  }
  if (p.mode == 2) {
    for (int s = tid; s < S; s += kAttnThreads) p.dcum[(long long)b * S + s] = dcum_l[s];
    float* dwa = p.dwck_acc + (long long)b * K * U;
    for (int i = tid; i < K * U; i += kAttnThreads) dwa[i] += dwk_l[i];
  }
  __syncthreads();
  AD_TICK(1, 6);
  // d(query input) = dq . Wq runs on the matrix cores inside the top cell's backward kernel
}

// ------------------------------------------------------------------ location-sensitive attention, split
// The one-workgroup-per-sample kernels above spend their time in VALU work on ONE compute unit
// (scores: 200 positions x 128 units x (32 location taps + tanh); their gradient: three times that)
// and in memory round trips nothing overlaps (Wq: 256 KB per sample, values: 410 KB per sample, both
// L2-cold after the cell kernels streamed 32 MB of weights). For the location-sensitive mode
// (U == 128) a step is therefore cut over kLocParts x B workgroups by UNITS and over kLocCtxParts x B
// by context columns / source positions:
//   forward : ad_loc_scores_kernel   (part, b): q and the partial scores of 32 units, all positions
//             ad_loc_context_kernel  (cols, b): sum of the partial scores, masked softmax, alignments /
//                                               cumulative alignments, context columns
//   backward: ad_loc_dalign_kernel   (rows, b): total context gradient, d(alignment) of a range of positions
//             ad_loc_score_bwd_kernel(part, b): softmax backward, score gradient of 32 units
// A unit's quantities (q, dq, dnv, dWck column, dpre column) have ONE owner; what is summed over
// units crosses workgroups through small global buffers written by their owners and summed by the
// next kernel in a fixed order (partial scores [B, parts, S]; partial state gradients [B, parts, S]):
// no atomics on global memory, no spinning.
constexpr int kLocParts = 4;        // unit parts (32 units each)
constexpr int kLocUnits = 32;
constexpr int kLocStreams = 16;     // position streams per workgroup: a half wave = the 32 units of a stream
constexpr int kLocCtxParts = 8;

// sum over the 32 lanes of a half wave; valid in lanes 16..31 of each half
__device__ __forceinline__ float half_sum_dpp(float x) {
  x += dpp_mov<0xB1, 0xf>(0.f, x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E, 0xf>(0.f, x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x124, 0xf>(0.f, x);   // row_ror:4
  x += dpp_mov<0x128, 0xf>(0.f, x);   // row_ror:8
  x += dpp_mov<0x142, 0xa>(0.f, x);   // row_bcast:15 into rows 1, 3
  return x;
}

struct AdLoc {
  float* e_part;      // [B, kLocParts, S] partial scores (forward)
  float* dal;         // [B, S] d(alignment) incl. the carried state gradient (backward)
  float* dcum_part;   // [B, kLocParts, S] this step's state-gradient contributions per unit part
};

__host__ __device__ inline size_t loc_fwd_lds_floats(int H, int S) {
  return (size_t)H + 3 * kLocUnits + (S + kLocKMax) + (size_t)S * kLocUnits / 2 + 64;
}

__device__ __forceinline__ void ad_loc_scores_body(const AdAttn& p, const AdLoc& x) {
  extern __shared__ float lds_raw[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int H = p.H, U = p.U, S = p.S, K = p.loc_k;
  float* hq = lds_raw;
  float* q = hq + H;
  float* nv = q + kLocUnits;
  float* bs = nv + kLocUnits;
  float* cum = bs + kLocUnits;                                   // [S + kLocKMax], zero padded
  uint16_t* keys = reinterpret_cast<uint16_t*>(cum + S + kLocKMax);   // [S][32] bf16
  const int slen = min(max(p.src_len[b], 0), S);
  const int u0 = part * kLocUnits;
  const long long row = (long long)b * p.T + p.t;
  // this wave's rows of Wq do not depend on anything staged below: requested first, so that their round trip
  // overlaps the staging (and its barrier) instead of following it
  constexpr int UB = kLocUnits / kAttnWaves;
  u32x4 wv0[UB][2];
#pragma unroll
  for (int i = 0; i < UB; ++i)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      wv0[i][hh] = *reinterpret_cast<const u32x4*>(p.wq + (long long)(u0 + wave * UB + i) * H +
                                                   min(lane * 8 + 512 * hh, H - 8));
  __builtin_amdgcn_sched_barrier(0);
  // query input, this part's key columns (64 B per position), padded cumulative alignments
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  for (int h8 = tid; h8 < H / 8; h8 += kAttnThreads) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(yq + h8 * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { hq[h8 * 8 + 2 * e] = bflo(v[e]); hq[h8 * 8 + 2 * e + 1] = bfhi(v[e]); }
  }
  {
    const bf16_t* kb = p.keys + (long long)b * S * U + u0;
    for (int i = tid; i < slen * 4; i += kAttnThreads) {
      const int sp = i >> 2, c = i & 3;
      *reinterpret_cast<u32x4*>(keys + sp * kLocUnits + c * 8) =
          *reinterpret_cast<const u32x4*>(kb + (long long)sp * U + c * 8);
    }
    const int padl = (K - 1) / 2;
    const float* cs = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
    for (int i = tid; i < S + kLocKMax; i += kAttnThreads) {
      const int sp = i - padl;
      cum[i] = (sp >= 0 && sp < S) ? cs[sp] : 0.f;
    }
  }
  if (tid < kLocUnits) {
    const int u = u0 + tid;
    nv[tid] = p.v[u];
    bs[tid] = ((p.use_bias && p.bias) ? p.bias[u] : 0.f) + p.wck[(long long)K * U + u];
  }
  __syncthreads();
  // q[u] = hq . Wq[u, :] for the 32 units of the part: 4 units per wave
  {
    float acc[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) acc[i] = 0.f;
    for (int h = lane * 8; h < H; h += 1024) {
      u32x4 wv[UB][2];
#pragma unroll
      for (int i = 0; i < UB; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          wv[i][hh] = h == lane * 8 ? wv0[i][hh]
                                    : *reinterpret_cast<const u32x4*>(p.wq + (long long)(u0 + wave * UB + i) * H +
                                                                      min(h + 512 * hh, H - 8));
#pragma unroll
      for (int i = 0; i < UB; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          if (h + 512 * hh < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i] += bflo(wv[i][hh][e]) * hq[h + 512 * hh + 2 * e] + bfhi(wv[i][hh][e]) * hq[h + 512 * hh + 2 * e + 1];
          }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const float sq = wave_sum_dpp(acc[i]);
      if (lane == 0) {
        q[wave * UB + i] = sq;
        p.q_seq[row * U + u0 + wave * UB + i] = sq;
      }
    }
  }
  __syncthreads();
  // partial scores: a half wave = the 32 units of one stream of positions; 32-entry register window
  // over the padded cumulative alignments (slot (j + k) & 31 holds cum[s + k] at the j-th step of a
  // 32-step block: one new LDS read per position)
  const int ul = lane & 31, st = wave * 2 + (lane >> 5);
  float wk[kLocKMax];
#pragma unroll
  for (int k = 0; k < kLocKMax; ++k) wk[k] = k < K ? p.wck[(long long)k * U + u0 + ul] : 0.f;
  const int chunk = (slen + kLocStreams - 1) / kLocStreams;
  const int c0 = st * chunk, c1 = min(c0 + chunk, slen);
  const float qb = q[ul] + bs[ul], nv0 = nv[ul];
  float* eo = x.e_part + ((long long)b * kLocParts + part) * S;
  float cw[kLocKMax];
#pragma unroll
  for (int k = 0; k < kLocKMax; ++k) cw[k] = cum[min(c0, S) + k];
  for (int i0 = 0; i0 < chunk; i0 += kLocKMax) {        // uniform trip count: the streams stay in step
#pragma unroll
    for (int j = 0; j < kLocKMax; ++j) {
      const int sp = c0 + i0 + j;
      if (i0 + j < chunk) {
        float x0 = qb;
        if (sp < c1) x0 += bf2f(keys[sp * kLocUnits + ul]);
#pragma unroll
        for (int k = 0; k < kLocKMax; ++k) x0 += cw[(j + k) & 31] * wk[k];
        float pr = half_sum_dpp(nv0 * tanh_fast(x0));
        if (ul == 31 && sp < c1) eo[sp] = pr;
        cw[j & 31] = sp < S ? cum[sp + kLocKMax] : 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(kAttnThreads) void ad_loc_scores_kernel(AdAttn p, AdLoc x) { ad_loc_scores_body(p, x); }
// sum of the partial scores -> masked softmax -> alignments (+ cumulative) -> context columns
__global__ __launch_bounds__(256) void ad_loc_context_kernel(AdAttn p, AdLoc x, int ncg, int nsp) {
  extern __shared__ float lds_raw[];
  const int cpart = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int M = p.M, S = p.S;
  float* e = lds_raw;                  // [S]
  float* red = e + S;                  // [8]
  float* part = red + 8;               // [nsp][ncg * 8]
  const int slen = min(max(p.src_len[b], 0), S);
  const float* ep = x.e_part + (long long)b * kLocParts * S;
  float mx = -INFINITY;
  for (int sp = tid; sp < slen; sp += 256) {
    float v = ep[sp];
#pragma unroll
    for (int q = 1; q < kLocParts; ++q) v += ep[q * S + sp];
    e[sp] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max_dpp(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int sp = tid; sp < slen; sp += 256) {
    const float ex = __expf(e[sp] - mx);
    e[sp] = ex;
    sum += ex;
  }
  sum = wave_sum_dpp(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = red[4] + red[5] + red[6] + red[7];
  const float inv = slen > 0 ? 1.f / sum : 0.f;
  const long long row = (long long)b * p.T + p.t;
  for (int sp = tid; sp < S; sp += 256) {
    const float a = sp < slen ? e[sp] * inv : 0.f;
    e[sp] = a;
    if (cpart == 0) {
      p.align_seq[row * S + sp] = a;
      const long long ci = ((long long)b * (p.T + 1) + p.t) * S + sp;
      p.cum_seq[ci + S] = p.cum_seq[ci] + a;
    }
  }
  __syncthreads();
  // context columns [cpart * ncg * 8, + ncg * 8): thread = (8-column group, slice of the positions)
  const int MQ = ncg * 8, m0 = cpart * MQ;
  const int cg = tid % ncg, sq = tid / ncg;
  if (sq < nsp) {
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* vp = p.values + (long long)b * S * M + m0 + cg * 8;
#pragma unroll 8
    for (int sp = sq; sp < slen; sp += nsp) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(vp + (long long)sp * M);
      const float a = e[sp];
#pragma unroll
      for (int k = 0; k < 4; ++k) { a8[2 * k] += a * bflo(v[k]); a8[2 * k + 1] += a * bfhi(v[k]); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[sq * MQ + cg * 8 + k] = a8[k];
  }
  __syncthreads();
  if (tid < ncg) {
    float c8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c8[k] = 0.f;
      for (int q = 0; q < nsp; ++q) c8[k] += part[q * MQ + tid * 8 + k];
    }
    const int m8 = (m0 >> 3) + tid;
    bf16_t* ctx = p.ctx + (long long)b * p.ctx_bs + (long long)p.t * p.ctx_ts;
    bf16_t* cat = p.cat0 + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc0;
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(c8[2 * k], c8[2 * k + 1]);
    *reinterpret_cast<u32x4*>(ctx + m8 * 8) = o;
    if (p.attn_in_keep < 1.f) {
      const unsigned long long idx8 = (((unsigned long long)b * (p.T + 1) + p.t + 1) * M) / 8 + m8;
      const uint32_t bits = dropout_bits8(p.attn_in_seed, idx8, p.attn_in_keep);
      const float ik = 1.f / p.attn_in_keep;
#pragma unroll
      for (int k = 0; k < 8; ++k) c8[k] = ((bits >> k) & 1u) ? c8[k] * ik : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pack2bf(c8[2 * k], c8[2 * k + 1]);
    }
    *reinterpret_cast<u32x4*>(cat + m8 * 8) = o;
  }
}

// total context gradient of the step (every workgroup of a sample computes it; part 0 stores it) and
// d(alignment)[s] = dctx . values[b, s, :] + carried state gradient, for a range of positions. The
// state-gradient carry dcum[b, s] is owned here: dcum += the previous backward step's per-part
// contributions, summed in part order.
__global__ __launch_bounds__(256) void ad_loc_dalign_kernel(AdAttn p, AdLoc x) {
  extern __shared__ float lds_raw[];
  const int rp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int M = p.M, S = p.S;
  float* dctx = lds_raw;   // [M]
  const int slen = min(max(p.src_len[b], 0), S);
  const long long row = (long long)b * p.T + p.t;
  const bf16_t* ext = p.dctx_ext ? p.dctx_ext + (long long)b * p.dctx_bs + (long long)p.t * p.dctx_ts : nullptr;
  for (int m8 = tid; m8 < M / 8; m8 += 256) {
    float c8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ext) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(ext + m8 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { c8[2 * e] = bflo(v[e]); c8[2 * e + 1] = bfhi(v[e]); }
    }
    if (!p.last) {
      const float* da = p.dattn + (long long)b * M + m8 * 8;
      uint32_t bits = 0xffu;
      float ik = 1.f;
      if (p.attn_in_keep < 1.f) {
        const unsigned long long idx8 = (((unsigned long long)b * (p.T + 1) + p.t + 1) * M) / 8 + m8;
        bits = dropout_bits8(p.attn_in_seed, idx8, p.attn_in_keep);
        ik = 1.f / p.attn_in_keep;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) if ((bits >> e) & 1u) c8[e] += da[e] * ik;
    }
    if (rp == 0) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(c8[2 * e], c8[2 * e + 1]);
      *reinterpret_cast<u32x4*>(p.dctx_seq + row * M + m8 * 8) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dctx[m8 * 8 + e] = c8[e];
  }
  __syncthreads();
  const int chunk = (S + kLocCtxParts - 1) / kLocCtxParts;
  const int c0 = rp * chunk, c1 = min(c0 + chunk, S);
  // the state-gradient carry of this range (all S: positions past slen still receive contributions
  // through the filter's halo, they are never read back into a live alignment)
  float* carry = dctx + M;   // [chunk]
  for (int sp = c0 + tid; sp < c1; sp += 256) {
    float d = p.dcum[(long long)b * S + sp];
    const float* dp = x.dcum_part + (long long)b * kLocParts * S + sp;
#pragma unroll
    for (int q = 0; q < kLocParts; ++q) d += dp[q * S];
    p.dcum[(long long)b * S + sp] = d;
    carry[sp - c0] = d;
  }
  __syncthreads();
  const int e1 = min(c1, slen);
#pragma unroll 4
  for (int sp = c0 + wave; sp < e1; sp += 4) {
    float pr = 0.f;
    const bf16_t* vp = p.values + ((long long)b * S + sp) * M;
    for (int m = lane * 8; m < M; m += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(vp + m);
#pragma unroll
      for (int e = 0; e < 4; ++e) pr += bflo(v[e]) * dctx[m + 2 * e] + bfhi(v[e]) * dctx[m + 2 * e + 1];
    }
    pr = wave_sum_dpp(pr);
    if (lane == 0) x.dal[(long long)b * S + sp] = pr + carry[sp - c0];
  }
}

// state-gradient slots a stream emits: its chunk of positions + one filter width, in whole 32-step blocks
__host__ __device__ inline int loc_bwd_slab_pitch(int S) {
  const int chunk = (S + kLocStreams - 1) / kLocStreams;
  return ((chunk + kLocKMax + kLocKMax - 1) / kLocKMax) * kLocKMax;
}
__host__ __device__ inline size_t loc_bwd_lds_floats(int S, int K) {
  // q, nv, bs | e, dal, de | cum (padded) | dcum_l (padded) | dwk_l per stream | dqp, dnvp | red | keys | dslab
  return (size_t)3 * kLocUnits + 3 * S + (S + kLocKMax) + (S + 2 * kLocKMax) +
         (size_t)kLocStreams * K * kLocUnits + 2 * (size_t)kLocStreams * kLocUnits + 64 + (size_t)S * kLocUnits / 2 +
         (size_t)kLocStreams * loc_bwd_slab_pitch(S);
}

__global__ __launch_bounds__(kAttnThreads) void ad_loc_score_bwd_kernel(AdAttn p, AdLoc x) {
  extern __shared__ float lds_raw[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int U = p.U, S = p.S, K = p.loc_k;
  float* q = lds_raw;
  float* nv = q + kLocUnits;
  float* bs = nv + kLocUnits;
  float* e = bs + kLocUnits;                 // [S] alignments
  float* dal = e + S;                        // [S]
  float* de = dal + S;                       // [S]
  float* cum = de + S;                       // [S + kLocKMax]
  float* dcum_l = cum + S + kLocKMax;        // [S + 2 kLocKMax]: index = position + kLocKMax
  float* dwk_l = dcum_l + S + 2 * kLocKMax;  // [streams][K][32]: every stream's filter-gradient columns
  float* dqp = dwk_l + kLocStreams * K * kLocUnits;   // [streams][32]
  float* dnvp = dqp + kLocStreams * kLocUnits;
  float* red = dnvp + kLocStreams * kLocUnits;
  uint16_t* keys = reinterpret_cast<uint16_t*>(red + 64);
  // [streams][pitch]: the state-gradient slots of every stream, summed over the streams in order below
  // (an LDS float atomic per slot let up to four streams race on one position: last-bit differences
  // from run to run)
  float* dslab = red + 64 + (size_t)S * kLocUnits / 2;
  const int spitch = loc_bwd_slab_pitch(S);
  const int slen = min(max(p.src_len[b], 0), S);
  const int u0 = part * kLocUnits;
  const long long row = (long long)b * p.T + p.t;
  AD_TICK(1, 0);
  {
    const bf16_t* kb = p.keys + (long long)b * S * U + u0;
    for (int i = tid; i < slen * 4; i += kAttnThreads) {
      const int sp = i >> 2, c = i & 3;
      *reinterpret_cast<u32x4*>(keys + sp * kLocUnits + c * 8) =
          *reinterpret_cast<const u32x4*>(kb + (long long)sp * U + c * 8);
    }
    const int padl = (K - 1) / 2;
    const float* cs = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
    for (int i = tid; i < S + kLocKMax; i += kAttnThreads) {
      const int sp = i - padl;
      cum[i] = (sp >= 0 && sp < S) ? cs[sp] : 0.f;
    }
    for (int sp = tid; sp < S; sp += kAttnThreads) {
      e[sp] = p.align_seq[row * S + sp];
      dal[sp] = sp < slen ? x.dal[(long long)b * S + sp] : 0.f;
    }
  }
  if (tid < kLocUnits) {
    const int u = u0 + tid;
    q[tid] = p.q_seq[row * U + u];
    nv[tid] = p.v[u];
    bs[tid] = ((p.use_bias && p.bias) ? p.bias[u] : 0.f) + p.wck[(long long)K * U + u];
  }
  __syncthreads();
  AD_TICK(1, 1);
  float dot = 0.f;
  for (int sp = tid; sp < slen; sp += kAttnThreads) dot += e[sp] * dal[sp];
  dot = block_sum(dot, red);
  for (int sp = tid; sp < S; sp += kAttnThreads) de[sp] = sp < slen ? e[sp] * (dal[sp] - dot) : 0.f;
  __syncthreads();
  AD_TICK(1, 2);
  // score gradient of this part's 32 units: three 32-entry register windows slide with s (slot
  // (j + k) & 31 at the j-th step of a block): cw = padded cumulative alignments, gw = partial state
  // gradient sum_k' dpre[s', u] Wck[k', u] (lane-local along s; the sum over the 32 units once per
  // emitted position), dwk = the filter-gradient column of the unit
  const int padl = (K - 1) / 2;
  const int ul = lane & 31, st = wave * 2 + (lane >> 5), u = u0 + ul;
  const int chunk = (slen + kLocStreams - 1) / kLocStreams;
  const int c0 = st * chunk, c1 = min(c0 + chunk, slen);
  float dq0 = 0.f, dn0 = 0.f;
  {
    float wk[kLocKMax], dwk[kLocKMax], cw[kLocKMax], gw[kLocKMax];
#pragma unroll
    for (int k = 0; k < kLocKMax; ++k) {
      dwk[k] = 0.f;
      gw[k] = 0.f;
      wk[k] = k < K ? p.wck[(long long)k * U + u] : 0.f;
      cw[k] = cum[min(c0, S) + k];
    }
    const float nv0 = nv[ul], qb = q[ul] + bs[ul];
    bf16_t* dps = p.dpre_seq + (row * S) * U + u;
    // chunk + kLocKMax steps: the last kLocKMax only emit the remaining window slots. The branch
    // around the arithmetic is WAVE-uniform (the lower stream of the wave still has a position; the
    // upper stream runs with a zero score gradient once it is past its range): a per-lane branch here
    // made the compiler spill the four register windows (3.7 KB of scratch per lane)
    const int c0a = __builtin_amdgcn_readfirstlane(wave * 2 * chunk);
    for (int i0 = 0; i0 < chunk + kLocKMax; i0 += kLocKMax) {
#pragma unroll
      for (int j = 0; j < kLocKMax; ++j) {
        const int i = i0 + j, sp = c0 + i;
        {   // (whole 32-step blocks: the steps past chunk + kLocKMax emit zero slots — a second branch
            //  level around this body made the compiler spill the register windows: 3.6 KB of scratch)
          if (i < chunk && c0a + i < slen) {
            const bool on = sp < c1;
            float x0 = (on ? bf2f(keys[sp * kLocUnits + ul]) : 0.f) + qb;
#pragma unroll
            for (int k = 0; k < kLocKMax; ++k) x0 += cw[(j + k) & 31] * wk[k];
            const float t0 = tanh_fast(x0);
            const float des = on ? de[sp] : 0.f;
            const float d0 = des * nv0 * (1.f - t0 * t0);
            dq0 += d0;
            dn0 += des * t0;
            if (on) dps[(long long)sp * U] = f2bf(d0);
#pragma unroll
            for (int k = 0; k < kLocKMax; ++k) {
              dwk[k] += cw[(j + k) & 31] * d0;
              gw[(j + k) & 31] += d0 * wk[k];
            }
          }
          // slot j is complete: the state gradient at position sp - padl (of this stream's range)
          const float gsum = half_sum_dpp(gw[j & 31]);
          gw[j & 31] = 0.f;
          if (ul == 31 && i < spitch) dslab[st * spitch + i] = gsum;      // position c0 + i - padl
          cw[j & 31] = sp < S ? cum[min(sp, S) + kLocKMax] : 0.f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kLocKMax; ++k)
      if (k < K) dwk_l[(st * K + k) * kLocUnits + ul] = dwk[k];     // summed over the streams below, in order
  }
  AD_TICK(1, 3);
  dqp[st * kLocUnits + ul] = dq0;
  dnvp[st * kLocUnits + ul] = dn0;
  __syncthreads();
  AD_TICK(1, 4);
  if (tid < kLocUnits) {
    float dq = 0.f, dn = 0.f;
#pragma unroll
    for (int w = 0; w < kLocStreams; ++w) { dq += dqp[w * kLocUnits + tid]; dn += dnvp[w * kLocUnits + tid]; }
    p.dq_seq[row * U + u0 + tid] = f2bf(dq);
    p.dnv_acc[(long long)b * U + u0 + tid] += dn;
    p.dbd_acc[(long long)b * U + u0 + tid] += dq;
  }
  float* dpo = x.dcum_part + ((long long)b * kLocParts + part) * S;
  {
    // position sp was slot sp + padl - w * chunk of stream w (slots 0 .. nemit - 1 were written by every stream)
    const int nemit = ((chunk + 2 * kLocKMax - 1) / kLocKMax) * kLocKMax;
    for (int sp = tid; sp < S; sp += kAttnThreads) {
      float a = 0.f;
      for (int w = 0; w < kLocStreams; ++w) {
        const int i = sp + padl - w * chunk;
        if (i >= 0 && i < nemit) a += dslab[w * spitch + i];
      }
      dpo[sp] = a;
    }
  }
  float* dwa = p.dwck_acc + (long long)b * K * U;
  for (int i = tid; i < K * kLocUnits; i += kAttnThreads) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kLocStreams; ++w) a += dwk_l[w * K * kLocUnits + i];
    dwa[(i / kLocUnits) * U + u0 + (i % kLocUnits)] += a;
  }
  AD_TICK(1, 5);
  AD_TICK(1, 6);
}

// ------------------------------------------------------------------ cell backward
struct AdCellBwd {
  int B, T, H, t, last, KA, KB, KQ;
  float forget_bias;
  const int32_t* lens;
  const bf16_t* dy_ext; long long dy_bs, dy_ts;   // gradient of the (dropped) output or null
  const bf16_t* dgA; long long dgA_ld; const bf16_t* wAT; long long wAT_ld;   // upper layer, same step
  const bf16_t* dgB; long long dgB_ld; const bf16_t* wBT; long long wBT_ld;   // own layer, step t+1
  const bf16_t* dq; long long dq_ld; const bf16_t* wqT;                       // query layer: dq . Wq (top layer)
  const bf16_t* gates;   // [B,T,4H]
  const float* c_seq;    // [B,T,H]
  float* dc_carry;       // [B,H]
  bf16_t* dg_out;        // [B,T,4H]
  float out_keep;
  unsigned long long out_seed;
  float* part;           // split kernel: [H/32][B/32][kCbSplit] partial tiles
  int* ticket;           // split kernel: [H/32][B/32], zero before and after every launch
};

// dh = dropout'(dy_ext + dgA . WA^T + dq . Wq) + dgB . WB^T, then the LSTM gate derivatives.
// One workgroup = kCellBwdRows hidden units x 32 samples: with 32-row tiles a 1024-unit layer was
// 32 workgroups (an eighth of the chip), each streaming two 256 KB weight slices next to the two
// [32, 4H] gate-gradient blocks every workgroup reads anyway; 8-row tiles give 128 workgroups and
// 64 KB slices (rows 8..31 of the MFMA tile are padding). Both products are accumulated before
// ONE reduction through LDS.
constexpr int kCellBwdRows = 8;
constexpr int kCellBwdCols = 32;   // samples per workgroup. (Fetching wave 0's epilogue operands before the
                                   // products — in registers: spills at the 128-register budget of 16 waves; by
                                   // LDS-DMA: no spill — left the step at 95 us either way: not kept.)
                                   // (8 samples per workgroup — a quarter of the
                                   // gate-gradient bytes per CU, 4x the workgroups re-reading the weight
                                   // slices — was slower: 29.8 vs 26.4 us; the step is five dependent
                                   // rounds of loads, not bytes per CU.)
__global__ __launch_bounds__(64 * kBwdWaves) void ad_cell_bwd_kernel(AdCellBwd p) {
  __shared__ float red[kBwdWaves * 8 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * kCellBwdRows, b0 = blockIdx.y * kCellBwdCols;
  const int H = p.H;
  const bool vrow = l31 < kCellBwdRows && j0 + l31 < H;
  const int brow = b0 + l31;
  const bool vcol = l31 < kCellBwdCols && brow < p.B;
  f32x16 accwM, accwB;
#pragma unroll
  for (int e = 0; e < 16; ++e) { accwM[e] = 0.f; accwB[e] = 0.f; }
  if (p.KA > 0)
    tile_gemm_prefetch<kBwdWaves, 8>(vrow ? p.wAT + (long long)(j0 + l31) * p.wAT_ld : nullptr,
                                      vcol ? p.dgA + (long long)brow * p.dgA_ld : nullptr, p.KA, accwM, p.wAT);
  if (p.KQ > 0)
    tile_gemm_prefetch<kBwdWaves, 2>(vrow ? p.wqT + (long long)(j0 + l31) * p.KQ : nullptr,
                                     vcol ? p.dq + (long long)brow * p.dq_ld : nullptr, p.KQ, accwM, p.wqT);
  if (!p.last && p.KB > 0)
    tile_gemm_prefetch<kBwdWaves, 8>(vrow ? p.wBT + (long long)(j0 + l31) * p.wBT_ld : nullptr,
                                      vcol ? p.dgB + (long long)brow * p.dgB_ld : nullptr, p.KB, accwB, p.wBT);
  // rows 0..7 of the tile = accumulator elements 0..3 (row 4*(lane>>5) + e): 8 floats per lane and wave
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(wave * 8 + e) * 64 + lane] = accwM[e];
    red[(wave * 8 + 4 + e) * 64 + lane] = accwB[e];
  }
  __syncthreads();
  if (wave >= 1) return;
  float accM[4], accB[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float sm = 0.f, sb = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kBwdWaves; ++w2) {
      sm += red[(w2 * 8 + e) * 64 + lane];
      sb += red[(w2 * 8 + 4 + e) * 64 + lane];
    }
    accM[e] = sm; accB[e] = sb;
  }
  const int b = b0 + l31;
  if (l31 >= kCellBwdCols || b >= p.B) return;
  if (p.lens && p.t >= p.lens[b]) return;
  const int j = j0 + 4 * lhi;
  if (j >= H) return;
  const long long row = (long long)b * p.T + p.t;
  float dyv[4] = {accM[0], accM[1], accM[2], accM[3]};
  if (p.dy_ext) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p.dy_ext + (long long)b * p.dy_bs + (long long)p.t * p.dy_ts + j);
    dyv[0] += bflo(v[0]); dyv[1] += bfhi(v[0]); dyv[2] += bflo(v[1]); dyv[3] += bfhi(v[1]);
  }
  if (p.out_keep < 1.f) {
    const unsigned long long idx = (unsigned long long)row * H + j;
    const uint32_t bits = dropout_bits8(p.out_seed, idx >> 3, p.out_keep) >> (j & 4);
    const float inv = 1.f / p.out_keep;
#pragma unroll
    for (int e = 0; e < 4; ++e) dyv[e] = ((bits >> e) & 1u) ? dyv[e] * inv : 0.f;
  }
  float sv[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p.gates + row * (4 * H) + (long long)g * H + j);
    sv[g][0] = bflo(v[0]); sv[g][1] = bfhi(v[0]); sv[g][2] = bflo(v[1]); sv[g][3] = bfhi(v[1]);
  }
  f32x4 dcarry = {0.f, 0.f, 0.f, 0.f};
  if (!p.last) dcarry = *reinterpret_cast<const f32x4*>(p.dc_carry + (long long)b * H + j);
  const f32x4 cv = *reinterpret_cast<const f32x4*>(p.c_seq + row * H + j);
  f32x4 cprev = {0.f, 0.f, 0.f, 0.f};
  if (p.t > 0) cprev = *reinterpret_cast<const f32x4*>(p.c_seq + (row - 1) * H + j);
  f32x4 ndc;
  float dpre[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dh = dyv[e] + accB[e];
    const float ig = sv[0][e], fg = sv[1][e], gg = sv[2][e], og = sv[3][e];
    const float tc = tanhf(cv[e]);
    const float dc = dh * og * (1.f - tc * tc) + dcarry[e];
    dpre[0][e] = dc * gg * ig * (1.f - ig);         // i
    dpre[1][e] = dc * ig * (1.f - gg * gg);         // j
    dpre[2][e] = dc * cprev[e] * fg * (1.f - fg);   // f
    dpre[3][e] = dh * tc * og * (1.f - og);         // o
    ndc[e] = dc * fg;
  }
  *reinterpret_cast<f32x4*>(p.dc_carry + (long long)b * H + j) = ndc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x2 pk;
    pk[0] = pack2bf(dpre[g][0], dpre[g][1]);
    pk[1] = pack2bf(dpre[g][2], dpre[g][3]);
    *reinterpret_cast<u32x2*>(p.dg_out + row * (4 * H) + (long long)g * H + j) = pk;
  }
}

// ---- the same, cut kCbSplit ways along the reduction ----------------------------------------------
// The kernel above is five DEPENDENT rounds of loads per workgroup (two 4H-deep products of 2 rounds
// each + the query product) and every one of its 128 workgroups re-reads both [32, 4H] gate-gradient
// blocks for 8 useful rows of its MFMA tiles (512 KB of activations next to 128 KB of weights). Here a
// workgroup owns a FULL 32-unit tile and one quarter of the reduction of both products: one round of
// loads (8 + 8 slices per wave, all issued before the first MFMA), 128 KB of weights + 128 KB of
// activations per workgroup, the same 128 workgroups for H = 1024. The four partial tiles of a hidden
// block meet through a ticket (stores -> release -> one atomic per workgroup; the workgroup that draws
// the last ticket acquires, sums the four slabs in piece order — deterministic — and runs the gate
// derivatives; nobody spins). The epilogue operands are requested before the products by every piece
// (they are 30 registers; only the last arriver uses them), so the reducer's chain is slab reads only.
// Tacotron2 decoder shape (H = 1024, B = 32), backward time per decoder step: 95.8 us with the kernel
// above, 88.8 / 78.7 / 80.2 us cut 2 / 4 / 8 ways (-DOS2S_CB_SPLIT=n -DOS2S_CB_SLICES=m to rebuild one).
#ifndef OS2S_CB_SPLIT
#define OS2S_CB_SPLIT 4
#define OS2S_CB_SLICES 8
#endif
constexpr int kCbSplit = OS2S_CB_SPLIT;
constexpr int kCbWaves = 8;
constexpr int kCbSlices = OS2S_CB_SLICES;         // 16-deep slices per wave and product and batch
constexpr int kCbSlabFloats = 2 * 16 * 64;        // (M, B) x 16 accumulator registers x 64 lanes

__global__ __launch_bounds__(64 * kCbWaves) void ad_cell_bwd_split_kernel(AdCellBwd p) {
  __shared__ float red[kCbWaves * 32 * 64];       // 64 KB
  __shared__ int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = blockIdx.x, kq = blockIdx.y, bb = blockIdx.z;
  const int j0 = hb * 32, b0 = bb * 32;
  const int H = p.H;
  const bool vrow = j0 + l31 < H;
  const int brow = b0 + l31;
  const bool vcol = brow < p.B;
  // ---- epilogue operands of (sample l31, units j0 + 8*wave + 4*lhi .. +3), waves 0..3 ------------------
  const int ej = j0 + 8 * wave + 4 * lhi;
  const bool elive = wave < 4 && vcol && ej < H && !(p.lens && p.t >= p.lens[brow]);
  const long long erow = (long long)brow * p.T + p.t;
  u32x2 e_dy = {0u, 0u}, e_g[4];
  f32x4 e_dcarry = {0.f, 0.f, 0.f, 0.f}, e_cv = {0.f, 0.f, 0.f, 0.f}, e_cprev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) e_g[g] = u32x2{0u, 0u};
  if (elive) {
    if (p.dy_ext) e_dy = *reinterpret_cast<const u32x2*>(p.dy_ext + (long long)brow * p.dy_bs + (long long)p.t * p.dy_ts + ej);
#pragma unroll
    for (int g = 0; g < 4; ++g) e_g[g] = *reinterpret_cast<const u32x2*>(p.gates + erow * (4 * H) + (long long)g * H + ej);
    if (!p.last) e_dcarry = *reinterpret_cast<const f32x4*>(p.dc_carry + (long long)brow * H + ej);
    e_cv = *reinterpret_cast<const f32x4*>(p.c_seq + erow * H + ej);
    if (p.t > 0) e_cprev = *reinterpret_cast<const f32x4*>(p.c_seq + (erow - 1) * H + ej);
  }
  // ---- this piece of the two products (+ the whole query product in piece 0) ---------------------------
  f32x16 accM, accB;
#pragma unroll
  for (int e = 0; e < 16; ++e) { accM[e] = 0.f; accB[e] = 0.f; }
  const u32x4 zero = {0u, 0u, 0u, 0u};
  const bool doA = p.KA > 0, doB = !p.last && p.KB > 0;
  const bf16_t* const wA = p.wAT + (long long)(vrow ? j0 + l31 : 0) * p.wAT_ld;
  const bf16_t* const gA = p.dgA + (long long)(vcol ? brow : 0) * p.dgA_ld;
  const bf16_t* const wB = p.wBT + (long long)(vrow ? j0 + l31 : 0) * p.wBT_ld;
  const bf16_t* const gB = p.dgB + (long long)(vcol ? brow : 0) * p.dgB_ld;
  const int nitA = doA ? (p.KA + 15) >> 4 : 0, nitB = doB ? (p.KB + 15) >> 4 : 0;
  const int nperA = (nitA + kCbSplit - 1) / kCbSplit, nperB = (nitB + kCbSplit - 1) / kCbSplit;
  const int nper = max(nperA, nperB);
  u32x4 qa = zero, qb = zero;
  const bool doQ = p.KQ > 0 && kq == 0 && wave * 16 < p.KQ;       // KQ <= 16 * kCbWaves (checked by the host side)
  if (doQ) {
    const int ko = min(wave * 16 + lhi * 8, p.KQ - 8);
    qa = *reinterpret_cast<const u32x4*>(p.wqT + (long long)(vrow ? j0 + l31 : 0) * p.KQ + ko);
    qb = *reinterpret_cast<const u32x4*>(p.dq + (long long)(vcol ? brow : 0) * p.dq_ld + ko);
  }
  for (int base = 0; base < nper; base += kCbWaves * kCbSlices) {
    u32x4 wa[kCbSlices], ga[kCbSlices], wb[kCbSlices], gb[kCbSlices];
    if (doA) {
#pragma unroll
      for (int i = 0; i < kCbSlices; ++i) {
        const int s = kq * nperA + base + wave + kCbWaves * i;
        const int ko = min(s * 16 + lhi * 8, p.KA - 8);
        wa[i] = *reinterpret_cast<const u32x4*>(wA + ko);
        ga[i] = *reinterpret_cast<const u32x4*>(gA + ko);
      }
    }
    if (doB) {
#pragma unroll
      for (int i = 0; i < kCbSlices; ++i) {
        const int s = kq * nperB + base + wave + kCbWaves * i;
        const int ko = min(s * 16 + lhi * 8, p.KB - 8);
        wb[i] = *reinterpret_cast<const u32x4*>(wB + ko);
        gb[i] = *reinterpret_cast<const u32x4*>(gB + ko);
      }
    }
    if (doA) {
#pragma unroll
      for (int i = 0; i < kCbSlices; ++i) {
        const int sl = base + wave + kCbWaves * i, s = kq * nperA + sl;
        const bool kv = sl < nperA && s < nitA && s * 16 + lhi * 8 < p.KA;
        accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (kv && vrow) ? wa[i] : zero),
                                                       __builtin_bit_cast(bf16x8, (kv && vcol) ? ga[i] : zero), accM, 0, 0, 0);
      }
    }
    if (doB) {
#pragma unroll
      for (int i = 0; i < kCbSlices; ++i) {
        const int sl = base + wave + kCbWaves * i, s = kq * nperB + sl;
        const bool kv = sl < nperB && s < nitB && s * 16 + lhi * 8 < p.KB;
        accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (kv && vrow) ? wb[i] : zero),
                                                       __builtin_bit_cast(bf16x8, (kv && vcol) ? gb[i] : zero), accB, 0, 0, 0);
      }
    }
  }
  if (doQ) {
    const bool kv = wave * 16 + lhi * 8 < p.KQ;
    accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (kv && vrow) ? qa : zero),
                                                   __builtin_bit_cast(bf16x8, (kv && vcol) ? qb : zero), accM, 0, 0, 0);
  }
  // ---- workgroup sum of the 8 waves' tiles -> this piece's slab ------------------------------------------
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[(wave * 32 + r) * 64 + lane] = accM[r];
    red[(wave * 32 + 16 + r) * 64 + lane] = accB[r];
  }
  __syncthreads();
  float* const slab0 = p.part + ((size_t)(hb * gridDim.z + bb) * kCbSplit) * kCbSlabFloats;
#pragma unroll
  for (int q = 0; q < kCbSlabFloats / (64 * kCbWaves); ++q) {
    const int o = tid + 64 * kCbWaves * q;        // (register row rr = o / 64, lane o % 64)
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kCbWaves; ++w2) s += red[w2 * 32 * 64 + o];
    slab0[(size_t)kq * kCbSlabFloats + o] = s;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const ticket = p.ticket + hb * gridDim.z + bb;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != kCbSplit - 1) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero before and after every launch
  }
  __syncthreads();
  if (!elive) return;
  // ---- last arriver: the four slabs in piece order, then the gate derivatives -----------------------------
  float accm[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pc = 0; pc < kCbSplit; ++pc) {
    const float* const sl = slab0 + (size_t)pc * kCbSlabFloats;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      accm[e] += sl[(4 * wave + e) * 64 + lane];
      accb[e] += sl[(16 + 4 * wave + e) * 64 + lane];
    }
  }
  float dyv[4] = {accm[0] + bflo(e_dy[0]), accm[1] + bfhi(e_dy[0]), accm[2] + bflo(e_dy[1]), accm[3] + bfhi(e_dy[1])};
  if (p.out_keep < 1.f) {
    const unsigned long long idx = (unsigned long long)erow * H + ej;
    const uint32_t bits = dropout_bits8(p.out_seed, idx >> 3, p.out_keep) >> (ej & 4);
    const float inv = 1.f / p.out_keep;
#pragma unroll
    for (int e = 0; e < 4; ++e) dyv[e] = ((bits >> e) & 1u) ? dyv[e] * inv : 0.f;
  }
  float sv[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sv[g][0] = bflo(e_g[g][0]); sv[g][1] = bfhi(e_g[g][0]); sv[g][2] = bflo(e_g[g][1]); sv[g][3] = bfhi(e_g[g][1]);
  }
  f32x4 ndc;
  float dpre[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dh = dyv[e] + accb[e];
    const float ig = sv[0][e], fg = sv[1][e], gg = sv[2][e], og = sv[3][e];
    const float tc = tanhf(e_cv[e]);
    const float dc = dh * og * (1.f - tc * tc) + e_dcarry[e];
    dpre[0][e] = dc * gg * ig * (1.f - ig);         // i
    dpre[1][e] = dc * ig * (1.f - gg * gg);         // j
    dpre[2][e] = dc * e_cprev[e] * fg * (1.f - fg); // f
    dpre[3][e] = dh * tc * og * (1.f - og);         // o
    ndc[e] = dc * fg;
  }
  *reinterpret_cast<f32x4*>(p.dc_carry + (long long)brow * H + ej) = ndc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x2 pk;
    pk[0] = pack2bf(dpre[g][0], dpre[g][1]);
    pk[1] = pack2bf(dpre[g][2], dpre[g][3]);
    *reinterpret_cast<u32x2*>(p.dg_out + erow * (4 * H) + (long long)g * H + ej) = pk;
  }
}

// ------------------------------------------------------------------ post-loop passes
// dmem[b,s,m] = sum_t align[b,t,s] * dctx[b,t,m]   (bf16 out, zero past src_len)
// One workgroup per (256 columns, 16 source positions, sample): 13 x 2 x 32 = 832 workgroups for the
// Tacotron2 shapes (the first version looped over the source positions inside 64 workgroups and took
// 6.2 ms per step). The 16 alignment values of a time step are the same for every thread (scalar
// loads); four time steps are in flight per iteration.
__global__ __launch_bounds__(256) void ad_dvalues_kernel(const float* __restrict__ align,
                                                         const bf16_t* __restrict__ dctx,
                                                         const int32_t* __restrict__ src_len,
                                                         const int32_t* __restrict__ tgt_len, int T,
                                                         int S, int M, bf16_t* __restrict__ dmem) {
  const int b = blockIdx.z, s0 = blockIdx.y * 16, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int slen = min(max(src_len[b], 0), S);
  const int tl = tgt_len ? min(max(tgt_len[b], 0), T) : T;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (s0 < slen) {
    const bf16_t* dc = dctx + (long long)b * T * M + m;
    const float* al = align + (long long)b * T * S + s0;
    int t = 0;
    for (; t + 4 <= tl; t += 4) {
      float d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = bf2f(dc[(long long)(t + u) * M]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* ar = al + (long long)(t + u) * S;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += (s0 + i < S ? ar[i] : 0.f) * d[u];
      }
    }
    for (; t < tl; ++t) {
      const float d = bf2f(dc[(long long)t * M]);
      const float* ar = al + (long long)t * S;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += (s0 + i < S ? ar[i] : 0.f) * d;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (s0 + i < S) dmem[((long long)b * S + s0 + i) * M + m] = f2bf(s0 + i < slen ? acc[i] : 0.f);
}

// Wck[k,u] = sum_f conv_w[k,f] dense_w[f,u];  bd[u] = sum_f conv_b[f] dense_w[f,u]  (out: [K+1, U])
__global__ void ad_fold_location_kernel(const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                        const float* __restrict__ dense_w, int K, int F, int U,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (K + 1) * U) return;
  const int k = i / U, u = i - k * U;
  float a = 0.f;
  for (int f = 0; f < F; ++f) a += (k < K ? conv_w[k * F + f] : conv_b[f]) * dense_w[f * U + u];
  out[i] = a;
}

// gradients of conv_w [K,F], conv_b [F], dense_w [F,U] from d(Wck) and d(bd) partials per sample
__global__ __launch_bounds__(256) void ad_unfold_location_grads_kernel(
    const float* __restrict__ dwck_acc, const float* __restrict__ dbd_acc, int B, int K, int F, int U,
    const float* __restrict__ conv_w, const float* __restrict__ conv_b, const float* __restrict__ dense_w,
    float* __restrict__ dconv_w, float* __restrict__ dconv_b, float* __restrict__ ddense_w,
    float* __restrict__ tmp /* [(K+1)*U] */) {
  const int tid = threadIdx.x;
  for (int i = tid; i < (K + 1) * U; i += 256) {
    float a = 0.f;
    if (i < K * U) { for (int b = 0; b < B; ++b) a += dwck_acc[(long long)b * K * U + i]; }
    else { for (int b = 0; b < B; ++b) a += dbd_acc[(long long)b * U + (i - K * U)]; }
    tmp[i] = a;
  }
  __syncthreads();
  for (int i = tid; i < (K + 1) * F; i += 256) {   // d conv_w[k,f] (k < K), d conv_b[f] (k == K)
    const int k = i / F, f = i - k * F;
    float a = 0.f;
    for (int u = 0; u < U; ++u) a += tmp[k * U + u] * dense_w[f * U + u];
    if (k < K) dconv_w[k * F + f] += a; else dconv_b[f] += a;
  }
  for (int i = tid; i < F * U; i += 256) {
    const int f = i / U, u = i - f * U;
    float a = conv_b[f] * tmp[K * U + u];
    for (int k = 0; k < K; ++k) a += conv_w[k * F + f] * tmp[k * U + u];
    ddense_w[i] += a;
  }
}

// dkeys[b,s,u] = sum_{t < tgt_len[b]} dpre_seq[b,t,s,u]   (zero past src_len)
__global__ __launch_bounds__(256) void ad_dkeys_kernel(const bf16_t* __restrict__ dpre_seq,
                                                       const int32_t* __restrict__ src_len,
                                                       const int32_t* __restrict__ tgt_len, int T,
                                                       int S, int U, float* __restrict__ dkeys) {
  const int b = blockIdx.y;
  const long long su = (long long)S * U;
  const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= su) return;
  const int s = (int)(i8 / U);
  const int slen = min(max(src_len[b], 0), S);
  const int tl = tgt_len ? min(max(tgt_len[b], 0), T) : T;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (s < slen) {
    const bf16_t* src = dpre_seq + (long long)b * T * su + i8;
#pragma unroll 8
    for (int t = 0; t < tl; ++t) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long long)t * su);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[2 * e] += bflo(v[e]); a[2 * e + 1] += bfhi(v[e]); }
    }
  }
  float* o = dkeys + (long long)b * su + i8;
  *reinterpret_cast<f32x4*>(o) = f32x4{a[0], a[1], a[2], a[3]};
  *reinterpret_cast<f32x4*>(o + 4) = f32x4{a[4], a[5], a[6], a[7]};
}

// out[n] += sum_b acc[b, n]
__global__ void ad_reduce_rows_kernel(const float* __restrict__ acc, int B, long long N,
                                      float* __restrict__ out) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += acc[(long long)b * N + n];
  out[n] += s;
}

// gradient of the score vector parameters from d(normalised v) partials [B,U]
__global__ __launch_bounds__(256) void ad_score_vec_grads_kernel(const float* __restrict__ dnv_acc,
                                                                 int B, int U, int mode,
                                                                 const float* __restrict__ v,
                                                                 const float* __restrict__ g,
                                                                 float* __restrict__ dv,
                                                                 float* __restrict__ dg) {
  __shared__ float red[8];
  __shared__ float dn[1024];
  const int tid = threadIdx.x;
  float vv = 0.f, dv_dot = 0.f;
  for (int u = tid; u < U; u += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dnv_acc[(long long)b * U + u];
    dn[u] = s;
    vv += v[u] * v[u];
    dv_dot += s * v[u];
  }
  __syncthreads();
  if (mode != 1) {
    for (int u = tid; u < U; u += 256) dv[u] += dn[u];
    return;
  }
  vv = wave_sum_dpp(vv);
  dv_dot = wave_sum_dpp(dv_dot);
  if ((tid & 63) == 0) { red[tid >> 6] = vv; red[4 + (tid >> 6)] = dv_dot; }
  __syncthreads();
  vv = red[0] + red[1] + red[2] + red[3];
  dv_dot = red[4] + red[5] + red[6] + red[7];
  const float rn = rsqrtf(vv), gg = g[0];
  // nv = g * v / |v|:  dg = sum dn*v/|v|;  dv = g*(dn/|v| - v*(dn.v)/|v|^3)
  for (int u = tid; u < U; u += 256) dv[u] += gg * (dn[u] * rn - v[u] * dv_dot * rn * rn * rn);
  if (tid == 0) dg[0] += dv_dot * rn;
}

}  // namespace os2s

using namespace os2s;

#ifdef OS2S_ATTN_PHASE_TIMERS
extern "C" int os2s_debug_attn_phases(long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_attn_dbg), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

static size_t attn_lds_bytes(const os2s_attn_decoder_t* d, bool bwd) {
  return attn_lds_floats(d->H, d->M, d->U, d->S, d->score_mode, d->loc_k, bwd) * sizeof(float);
}

// the split location-attention kernels (OS2S_ATTN_SPLIT=0: the one-workgroup-per-sample kernels)
static bool loc_split(const os2s_attn_decoder_t* d) {
  static const int mode = [] { const char* e = getenv("OS2S_ATTN_SPLIT"); return e ? atoi(e) : 1; }();
  return mode != 0 && d->score_mode == 2 && d->U == kLocParts * kLocUnits && d->M % 8 == 0 &&
         loc_fwd_lds_floats(d->H, d->S) * sizeof(float) <= 64 * 1024 &&
         loc_bwd_lds_floats(d->S, d->loc_k) * sizeof(float) <= 160 * 1024;
}

extern "C" size_t os2s_attn_decoder_loc_ws_floats(int B, int S, int U, int loc_k) {
  return (size_t)(loc_k + 1) * U + (size_t)B * kLocParts * S;
}

static int ad_check(const os2s_attn_decoder_t* d) {
  OS2S_REQUIRE(d && d->B >= 1 && d->T >= 1 && d->S >= 1 && (d->L == 1 || d->L == 2));
  OS2S_REQUIRE(d->H % 8 == 0 && d->M % 8 == 0 && d->U % 128 == 0 && d->U <= 512);
  OS2S_REQUIRE(d->score_mode >= 0 && d->score_mode <= 3);
  if (d->score_mode == 3) OS2S_REQUIRE(d->U == d->H);     // Luong: the query IS the cell output
  OS2S_REQUIRE(d->t_begin >= 0 && d->t_begin <= d->t_end && d->t_end <= d->T);
  OS2S_REQUIRE(d->wcat[0] && d->wq && d->v && d->keys && d->values && d->src_len && d->gx0);
  OS2S_REQUIRE(d->cat[0] && d->c_seq[0] && d->align_seq && d->q_seq && d->y_top && d->ctx);
  if (d->L == 2) OS2S_REQUIRE(d->wcat[1] && d->cat[1] && d->c_seq[1]);
  if (d->score_mode == 1) OS2S_REQUIRE(d->g && d->b);
  if (d->score_mode == 2) {
    OS2S_REQUIRE(d->loc_k >= 1 && d->loc_k <= kLocKMax && d->loc_f >= 1 && d->U == 128);
    OS2S_REQUIRE(d->conv_w && d->conv_b && d->dense_w && d->cum_seq && d->loc_ws);
    if (d->use_bias) OS2S_REQUIRE(d->b);
  }
  if (attn_lds_bytes(d, true) > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  return OS2S_OK;
}

static void ad_fill_attn(const os2s_attn_decoder_t* d, AdAttn& a) {
  a.B = d->B; a.T = d->T; a.S = d->S; a.H = d->H; a.M = d->M; a.U = d->U;
  a.mode = d->score_mode; a.use_bias = d->use_bias; a.loc_k = d->loc_k;
  a.Kc0 = d->M + d->H;
  a.src_len = d->src_len; a.tgt_len = d->tgt_len;
  a.yq = (const bf16_t*)d->y_top; a.yq_bs = d->y_top_bs; a.yq_ts = d->y_top_ts;
  a.wq = (const bf16_t*)d->wq; a.keys = (const bf16_t*)d->keys; a.values = (const bf16_t*)d->values;
  a.v = d->v; a.g = d->g; a.bias = d->b; a.wck = d->loc_ws;
  a.cum_seq = d->cum_seq; a.align_seq = d->align_seq; a.q_seq = d->q_seq;
  a.ctx = (bf16_t*)d->ctx; a.ctx_bs = d->ctx_bs; a.ctx_ts = d->ctx_ts;
  a.cat0 = (bf16_t*)d->cat[0];
  a.attn_in_keep = d->attn_in_keep; a.attn_in_seed = d->attn_in_seed;
  a.dctx_ext = nullptr; a.dattn = nullptr; a.dctx_seq = nullptr; a.dcum = nullptr; a.dpre_seq = nullptr;
  a.dq_seq = nullptr; a.dhq = nullptr; a.dnv_acc = nullptr; a.dbd_acc = nullptr;
  a.dwck_acc = nullptr; a.last = 0; a.dctx_bs = a.dctx_ts = 0;
}

namespace os2s { struct TiLstm; }
static bool ad_fast_cells(const os2s_attn_decoder_t* d);
static int ad_launch_fast_cell(hipStream_t stream, const os2s_attn_decoder_t* d, int l, int t);
static int ad_launch_fast_scores(hipStream_t stream, const os2s::AdAttn& at, const os2s::AdLoc& lx, const os2s_attn_decoder_t* d);
static bool ad_fast_score_bwd(const os2s_attn_decoder_t* d);
static int ad_launch_fast_score_bwd(hipStream_t stream, const os2s::AdAttn& at, const os2s::AdLoc& lx, const os2s_attn_decoder_t* d);

extern "C" int os2s_attn_decoder_fwd(os2s_stream_t stream_, const os2s_attn_decoder_t* d) {
  const int rc = ad_check(d);
  if (rc != OS2S_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = attn_lds_bytes(d, false);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)ad_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  const int B = d->B, T = d->T, H = d->H, M = d->M, L = d->L;
  AdAttn at;
  ad_fill_attn(d, at);
  if (d->score_mode == 2) {
    const int n = (d->loc_k + 1) * d->U;
    OS2S_LAUNCH(ad_fold_location_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, d->conv_w,
                d->conv_b, d->dense_w, d->loc_k, d->loc_f, d->U, d->loc_ws);
  }
  // location-sensitive attention: the step is cut over kLocParts x B / ctx_parts x B workgroups
  const bool split = loc_split(d);
  AdLoc lx;
  lx.e_part = split ? d->loc_ws + (size_t)(d->loc_k + 1) * d->U : nullptr;
  lx.dal = nullptr; lx.dcum_part = nullptr;
  int ctx_parts = kLocCtxParts;
  while (ctx_parts > 1 && M % (8 * ctx_parts)) ctx_parts >>= 1;
  const int ncg = M / (8 * ctx_parts);
  const int nsp = 256 / ncg < 32 ? 256 / ncg : 32;
  const size_t lds_s = loc_fwd_lds_floats(H, d->S) * sizeof(float);
  const size_t lds_c = ((size_t)d->S + 8 + (size_t)nsp * ncg * 8) * sizeof(float);
  if (split && ncg > 256) return OS2S_ERR_UNSUPPORTED;
  dim3 cgrid(ceil_div(H, 8), ceil_div(B, 32));
  const bool fast = split && ad_fast_cells(d);
  for (int t = d->t_begin; t < d->t_end; ++t) {
    for (int l = 0; l < L; ++l) {
      if (fast) {
        const int r2 = ad_launch_fast_cell(stream, d, l, t);
        if (r2 != OS2S_OK) return r2;
        continue;
      }
      AdCellFwd c;
      c.B = B; c.T = T; c.H = H; c.t = t; c.forget_bias = d->forget_bias; c.lens = d->tgt_len;
      c.Kc = l == 0 ? M + H : 2 * H;
      c.cat = (const bf16_t*)d->cat[l]; c.w = (const bf16_t*)d->wcat[l]; c.bias = d->bias[l];
      c.w8 = d->wcat8[l]; c.w8_scale = d->wcat8_scale[l];
      if (c.w8 && !c.w8_scale) return OS2S_ERR_INVALID_ARG;
      c.gx = l == 0 ? (const bf16_t*)d->gx0 : nullptr;
      c.c_seq = d->c_seq[l]; c.gates = (bf16_t*)d->gates[l];
      c.h_next = (bf16_t*)d->cat[l] + (l == 0 ? M : H);
      if (l == L - 1) { c.y = (bf16_t*)d->y_top; c.y_bs = d->y_top_bs; c.y_ts = d->y_top_ts; }
      else { c.y = (bf16_t*)d->cat[l + 1]; c.y_bs = (long long)(T + 1) * 2 * H; c.y_ts = 2 * H; }
      c.out_keep = d->out_keep; c.out_seed = d->out_seed[l];
      OS2S_LAUNCH(ad_cell_fwd_kernel, cgrid, dim3(64 * kAdWaves), 0, stream, c);
    }
    at.t = t;
    if (split) {
      if (fast) {
        const int r2 = ad_launch_fast_scores(stream, at, lx, d);
        if (r2 != OS2S_OK) return r2;
      } else {
        OS2S_LAUNCH(ad_loc_scores_kernel, dim3(kLocParts, B), dim3(kAttnThreads), lds_s, stream, at, lx);
      }
      OS2S_LAUNCH(ad_loc_context_kernel, dim3(ctx_parts, B), dim3(256), lds_c, stream, at, lx, ncg, nsp);
    } else {
      OS2S_LAUNCH(ad_attn_fwd_kernel, dim3(B), dim3(kAttnThreads), lds, stream, at);
    }
  }
  return OS2S_OK;
}

extern "C" size_t os2s_attn_decoder_bwd_workspace_bytes(const os2s_attn_decoder_t* d) {
  if (!d) return 0;
  const size_t B = d->B;
  size_t n = B * d->M + B * d->H + 2 * B * d->H + B * d->S + B * d->U;
  if (d->score_mode == 2) n += B * d->U + B * d->loc_k * d->U + (size_t)(d->loc_k + 1) * d->U;
  if (d->score_mode == 2) n += B * d->S + B * kLocParts * d->S;     // split kernels: dal, dcum_part
  // cell backward cut along the reduction: partial tiles + tickets per (32 units, 32 samples)
  const size_t nblk = (size_t)ceil_div(d->H, 32) * ceil_div(d->B, 32);
  n += nblk * kCbSplit * kCbSlabFloats + nblk + 64;
  const size_t nda = (size_t)ceil_div(d->M, 32) * ceil_div(d->B, 32);        // d(attention input), same scheme
  n += nda * kDaSplit * kDaSlabFloats + nda + 64;
  return n * sizeof(float) + 1024;
}

// OS2S_CELL_SPLIT=0: the one-workgroup-per-8-units cell backward (A/B switch)
static bool cell_split(const os2s_attn_decoder_t* d) {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("OS2S_CELL_SPLIT");
    mode = e ? (atoi(e) != 0) : 1;
  }
  return mode == 1 && d->U <= 16 * kCbWaves && d->H % 8 == 0;
}

extern "C" int os2s_attn_decoder_bwd(os2s_stream_t stream_, const os2s_attn_decoder_t* d,
                                     const os2s_attn_decoder_grads_t* gr, void* workspace,
                                     size_t workspace_bytes) {
  const int rc = ad_check(d);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(gr && workspace && d->gates[0] && gr->wcatT[0] && gr->wqT && gr->dg[0] && gr->dq_seq && gr->dctx_seq && gr->dkeys && gr->dpre_seq && gr->dmem);
  OS2S_REQUIRE(gr->dy_top || gr->dctx_ext);
  OS2S_REQUIRE(gr->dv);
  if (d->L == 2) OS2S_REQUIRE(d->gates[1] && gr->wcatT[1] && gr->dg[1]);
  if (d->score_mode == 1) OS2S_REQUIRE(gr->dg_scalar);
  if (d->score_mode == 2) OS2S_REQUIRE(gr->dconv_w && gr->dconv_b && gr->ddense_w);
  if (workspace_bytes < os2s_attn_decoder_bwd_workspace_bytes(d)) return OS2S_ERR_WORKSPACE;
  OS2S_REQUIRE(d->t_begin == 0 && d->t_end == d->T);
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = attn_lds_bytes(d, true);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(d->score_mode == 2 ? (const void*)ad_attn_bwd_kernel<true> : (const void*)ad_attn_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  const int B = d->B, T = d->T, H = d->H, M = d->M, L = d->L, U = d->U, S = d->S;
  const int K = d->loc_k, F = d->loc_f;
  if (hipMemsetAsync(workspace, 0, os2s_attn_decoder_bwd_workspace_bytes(d), stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  float* ws = (float*)workspace;
  float* dattn = ws; ws += (size_t)B * M;
  float* dhq = ws; ws += (size_t)B * H;
  float* dcc[2]; dcc[0] = ws; ws += (size_t)B * H; dcc[1] = ws; ws += (size_t)B * H;
  float* dcum = ws; ws += (size_t)B * S;
  float* dnv_acc = ws; ws += (size_t)B * U;
  float *dbd_acc = nullptr, *dwck_acc = nullptr, *unfold_tmp = nullptr;
  if (d->score_mode == 2) {
    dbd_acc = ws; ws += (size_t)B * U;
    dwck_acc = ws; ws += (size_t)B * K * U;
    unfold_tmp = ws; ws += (size_t)(K + 1) * U;
  }
  const bool split = loc_split(d);
  AdLoc lx;
  lx.e_part = nullptr; lx.dal = nullptr; lx.dcum_part = nullptr;
  if (d->score_mode == 2) {
    lx.dal = ws; ws += (size_t)B * S;
    lx.dcum_part = ws; ws += (size_t)B * kLocParts * S;
  }
  const size_t ncb = (size_t)ceil_div(H, 32) * ceil_div(B, 32);
  float* const cb_part = ws; ws += ncb * kCbSplit * kCbSlabFloats;
  int* const cb_ticket = reinterpret_cast<int*>(ws); ws += ncb;
  const bool csplit = cell_split(d);
  const size_t nda = (size_t)ceil_div(M, 32) * ceil_div(B, 32);
  float* const da_part = ws; ws += nda * kDaSplit * kDaSlabFloats;
  int* const da_ticket = reinterpret_cast<int*>(ws); ws += nda;
  const size_t lds_da = ((size_t)M + ceil_div(S, kLocCtxParts)) * sizeof(float);
  const size_t lds_sb = loc_bwd_lds_floats(S, K) * sizeof(float);
  if (split && lds_sb > 64 * 1024 &&
      hipFuncSetAttribute((const void*)ad_loc_score_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sb) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  // gate gradients of finished steps are zero; dkeys accumulates
  for (int l = 0; l < L; ++l)
    if (hipMemsetAsync(gr->dg[l], 0, (size_t)B * T * 4 * H * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (hipMemsetAsync(gr->dq_seq, 0, (size_t)B * T * U * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (hipMemsetAsync(gr->dctx_seq, 0, (size_t)B * T * M * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;

  AdAttn at;
  ad_fill_attn(d, at);
  at.dctx_ext = (const bf16_t*)gr->dctx_ext; at.dctx_bs = gr->dctx_bs; at.dctx_ts = gr->dctx_ts;
  at.dattn = dattn; at.dctx_seq = (bf16_t*)gr->dctx_seq; at.dcum = dcum; at.dpre_seq = (bf16_t*)gr->dpre_seq;
  at.dq_seq = (bf16_t*)gr->dq_seq; at.dhq = dhq; at.dnv_acc = dnv_acc; at.dbd_acc = dbd_acc;
  at.dwck_acc = dwck_acc;
  const long long GH = 4LL * H;
  dim3 cgrid(ceil_div(H, kCellBwdRows), ceil_div(B, kCellBwdCols));
  const bool fast_sb = split && ad_fast_score_bwd(d);
  for (int t = T - 1; t >= 0; --t) {
    const int last = (t == T - 1);
    if (!last) {
      AdDattn g;
      g.B = B; g.M = M; g.K = (int)GH; g.wT = (const bf16_t*)gr->wcatT[0];
      g.dg = (const bf16_t*)gr->dg[0] + (long long)(t + 1) * GH; g.ld = (long long)T * GH; g.out = dattn;
      if (csplit && M % 4 == 0) {
        AdDattnSplit gs;
        gs.g = g; gs.part = da_part; gs.ticket = da_ticket;
        OS2S_LAUNCH(ad_dattn_split_kernel, dim3(ceil_div(M, 32), kDaSplit, ceil_div(B, 32)), dim3(64 * kDaWaves), 0,
                    stream, gs);
      } else {
        OS2S_LAUNCH(ad_dattn_kernel, dim3(ceil_div(M, 8), ceil_div(B, 32)), dim3(64 * kBwdWaves), 0, stream, g);
      }
    }
    at.t = t; at.last = last;
    if (split) {
      OS2S_LAUNCH(ad_loc_dalign_kernel, dim3(kLocCtxParts, B), dim3(256), lds_da, stream, at, lx);
      if (fast_sb) {
        const int r2 = ad_launch_fast_score_bwd(stream, at, lx, d);
        if (r2 != OS2S_OK) return r2;
      } else {
        OS2S_LAUNCH(ad_loc_score_bwd_kernel, dim3(kLocParts, B), dim3(kAttnThreads), lds_sb, stream, at, lx);
      }
    } else if (d->score_mode == 2) { OS2S_LAUNCH(ad_attn_bwd_kernel<true>, dim3(B), dim3(kAttnThreads), lds, stream, at); }
    else { OS2S_LAUNCH(ad_attn_bwd_kernel<false>, dim3(B), dim3(kAttnThreads), lds, stream, at); }
    for (int l = L - 1; l >= 0; --l) {
      AdCellBwd c;
      c.B = B; c.T = T; c.H = H; c.t = t; c.last = last; c.forget_bias = d->forget_bias; c.lens = d->tgt_len;
      c.dy_ext = nullptr; c.dy_bs = c.dy_ts = 0;
      c.dgA = nullptr; c.wAT = nullptr; c.KA = 0; c.dgA_ld = c.wAT_ld = 0;
      c.dq = nullptr; c.wqT = nullptr; c.KQ = 0; c.dq_ld = 0;
      if (l == L - 1) {
        c.dy_ext = (const bf16_t*)gr->dy_top; c.dy_bs = gr->dy_top_bs; c.dy_ts = gr->dy_top_ts;
        c.dq = (const bf16_t*)gr->dq_seq + (long long)t * U; c.dq_ld = (long long)T * U;
        c.wqT = (const bf16_t*)gr->wqT; c.KQ = U;
      } else {   // output feeds the layer above at the same step (columns 0..H of its cat)
        c.dgA = (const bf16_t*)gr->dg[l + 1] + (long long)t * GH; c.dgA_ld = (long long)T * GH;
        c.wAT = (const bf16_t*)gr->wcatT[l + 1]; c.wAT_ld = GH; c.KA = (int)GH;
      }
      c.dgB = (const bf16_t*)gr->dg[l] + (long long)(t + 1) * GH; c.dgB_ld = (long long)T * GH;
      c.wBT = (const bf16_t*)gr->wcatT[l] + (long long)(l == 0 ? M : H) * GH; c.wBT_ld = GH; c.KB = (int)GH;
      c.gates = (const bf16_t*)d->gates[l]; c.c_seq = d->c_seq[l]; c.dc_carry = dcc[l];
      c.dg_out = (bf16_t*)gr->dg[l]; c.out_keep = d->out_keep; c.out_seed = d->out_seed[l];
      c.part = cb_part; c.ticket = cb_ticket;
      if (csplit) {
        OS2S_LAUNCH(ad_cell_bwd_split_kernel, dim3(ceil_div(H, 32), kCbSplit, ceil_div(B, 32)), dim3(64 * kCbWaves), 0,
                    stream, c);
      } else {
        OS2S_LAUNCH(ad_cell_bwd_kernel, cgrid, dim3(64 * kBwdWaves), 0, stream, c);
      }
    }
  }
  OS2S_LAUNCH(ad_dkeys_kernel, dim3(ceil_div((long long)S * U / 8, 256), B), dim3(256), 0, stream,
              (const bf16_t*)gr->dpre_seq, d->src_len, d->tgt_len, T, S, U, gr->dkeys);
  OS2S_LAUNCH(ad_dvalues_kernel, dim3(ceil_div(M, 256), ceil_div(S, 16), B), dim3(256), 0, stream, d->align_seq,
              (const bf16_t*)gr->dctx_seq, d->src_len, d->tgt_len, T, S, M, (bf16_t*)gr->dmem);
  OS2S_LAUNCH(ad_score_vec_grads_kernel, dim3(1), dim3(256), 0, stream, dnv_acc, B, U, d->score_mode,
              d->v, d->g, gr->dv, gr->dg_scalar);
  if (d->score_mode == 2) {
    OS2S_LAUNCH(ad_unfold_location_grads_kernel, dim3(1), dim3(256), 0, stream, dwck_acc, dbd_acc, B, K,
                F, U, d->conv_w, d->conv_b, d->dense_w, gr->dconv_w, gr->dconv_b, gr->ddense_w, unfold_tmp);
  }
  return OS2S_OK;
}

// ---- e4m3 weight copies -----------------------------------------------------------------------
// q[r, k] = e4m3(w[r, k] / scale[r]), scale[r] = max_k |w[r, k]| / 448 (OCP e4m3fn: largest finite
// value 448; an all-zero row gets scale 1). One wave per row, two passes over the row.
namespace os2s {
__global__ __launch_bounds__(256) void quantize_rows_e4m3_kernel(const bf16_t* __restrict__ w, int rows,
                                                               int K, uint8_t* __restrict__ q,
                                                               float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bf16_t* wr = w + (long long)r * K;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(wr + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bflo(v[e])), fabsf(bfhi(v[e]))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.f : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) scale[r] = sc;
  for (int k = lane * 8; k < K; k += 512) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(wr + k);
    u32x2 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = fminf(fmaxf(bflo(v[2 * h]) * inv, -448.f), 448.f);
      const float a1 = fminf(fmaxf(bfhi(v[2 * h]) * inv, -448.f), 448.f);
      const float a2 = fminf(fmaxf(bflo(v[2 * h + 1]) * inv, -448.f), 448.f);
      const float a3 = fminf(fmaxf(bfhi(v[2 * h + 1]) * inv, -448.f), 448.f);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, pk, true);
      o[h] = (uint32_t)pk;
    }
    *reinterpret_cast<u32x2*>(q + (long long)r * K + k) = o;
  }
}
}  // namespace os2s

extern "C" int os2s_quantize_rows_e4m3(os2s_stream_t stream, const uint16_t* w, int rows, int K,
                                       uint8_t* q, float* scale) {
  using namespace os2s;
  OS2S_REQUIRE(w && q && scale && rows >= 1 && K >= 8 && K % 8 == 0);
  OS2S_LAUNCH(quantize_rows_e4m3_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, w,
              rows, K, q, scale);
  return OS2S_OK;
}

#include "tacotron_infer.hpp"
