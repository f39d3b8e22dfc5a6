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
#include "os2s_common.hpp"
#include "rnn_tile.hpp"

namespace os2s {

constexpr int kAdWaves = 8;
constexpr int kAttnThreads = 256;

__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f / (1.f + __expf(2.f * x)); }

// ------------------------------------------------------------------ cell forward
struct AdCellFwd {
  int B, T, H, Kc, t;
  float forget_bias;
  const int32_t* lens;
  const bf16_t* cat;   // [B, T+1, Kc]
  const bf16_t* w;     // [4H, Kc]
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

__global__ __launch_bounds__(64 * kAdWaves) void ad_cell_fwd_kernel(AdCellFwd p) {
  __shared__ float red[kAdWaves * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int H = p.H;
  f32x16 accw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[g][e] = 0.f;
  tile_gemm_splitk<4, kAdWaves, 2>(p.w, p.Kc, H, j0, H, p.cat + (long long)p.t * p.Kc,
                                   (long long)(p.T + 1) * p.Kc, b0, p.B, p.Kc, accw);
  float acc[4][4];
  tile_reduce_quarters<4, kAdWaves>(accw, red, acc);
  if (wave >= 4) return;
  const int b = b0 + l31;
  if (b >= p.B) return;
  if (p.lens && p.t >= p.lens[b]) return;   // finished sample: buffers stay zero
  const int j = j0 + 8 * wave + 4 * lhi;
  if (j >= H) return;
  const long long row = (long long)b * p.T + p.t;
  float pre[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pre[g][e] = acc[g][e];
    if (p.gx) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(p.gx + row * (4 * H) + (long long)g * H + j);
      pre[g][0] += bflo(v[0]); pre[g][1] += bfhi(v[0]); pre[g][2] += bflo(v[1]); pre[g][3] += bfhi(v[1]);
    }
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) pre[g][e] += p.bias[g * H + j + e];
    }
  }
  f32x4 cprev = {0.f, 0.f, 0.f, 0.f};
  if (p.t > 0) cprev = *reinterpret_cast<const f32x4*>(p.c_seq + (row - 1) * H + j);
  f32x4 cn;
  float hn[4], sv[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ig = sigmoidf_(pre[0][e]), gg = tanhf(pre[1][e]);
    const float fg = sigmoidf_(pre[2][e] + p.forget_bias), og = sigmoidf_(pre[3][e]);
    cn[e] = cprev[e] * fg + ig * gg;
    hn[e] = tanhf(cn[e]) * og;
    sv[0][e] = ig; sv[1][e] = fg; sv[2][e] = gg; sv[3][e] = og;
  }
  *reinterpret_cast<f32x4*>(p.c_seq + row * H + j) = cn;
  if (p.gates) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 pk;
      pk[0] = pack2bf(sv[g][0], sv[g][1]);
      pk[1] = pack2bf(sv[g][2], sv[g][3]);
      *reinterpret_cast<u32x2*>(p.gates + row * (4 * H) + (long long)g * H + j) = pk;
    }
  }
  u32x2 hr;
  hr[0] = pack2bf(hn[0], hn[1]);
  hr[1] = pack2bf(hn[2], hn[3]);
  *reinterpret_cast<u32x2*>(p.h_next + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc + j) = hr;
  if (p.out_keep < 1.f) {
    const unsigned long long idx = (unsigned long long)row * H + j;
    const uint32_t bits = dropout_bits8(p.out_seed, idx >> 3, p.out_keep) >> (j & 4);
    const float inv = 1.f / p.out_keep;
#pragma unroll
    for (int e = 0; e < 4; ++e) hn[e] = ((bits >> e) & 1u) ? hn[e] * inv : 0.f;
    hr[0] = pack2bf(hn[0], hn[1]);
    hr[1] = pack2bf(hn[2], hn[3]);
  }
  *reinterpret_cast<u32x2*>(p.y + (long long)b * p.y_bs + (long long)p.t * p.y_ts + j) = hr;
}

// ------------------------------------------------------------------ attention (shared)
struct AdAttn {
  int B, T, S, H, M, U, t, mode, use_bias, loc_k, loc_f, Kc0, last;
  const int32_t* src_len;
  const int32_t* tgt_len;
  const bf16_t* yq;      // query input rows: yq + b*yq_bs + t*yq_ts
  long long yq_bs, yq_ts;
  const bf16_t* wq;      // [U, H]
  const bf16_t* keys;    // [B, S, U]
  const bf16_t* values;  // [B, S, M]
  const float* v; const float* g; const float* bias;
  const float* conv_w; const float* conv_b; const float* dense_w;
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
  float* dkeys;          // [B, S, U] fp32 accumulator
  bf16_t* dq_seq;        // [B, T, U]
  float* dhq;            // [B, H]
  float* dnv_acc;        // [B, U]
  float* ddense_acc;     // [B, F, U]
  float* dconvw_acc;     // [B, K, F]
  float* dconvb_acc;     // [B, F]
};

__device__ __forceinline__ float block_sum(float x, float* red) {
  x = wave_sum(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kAttnThreads / 64; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ float block_max(float x, float* red) {
  x = wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int w = 1; w < kAttnThreads / 64; ++w) s = fmaxf(s, red[w]);
  return s;
}

// LDS carve-up (floats) shared by the forward and backward attention kernels
struct AttnLds {
  float *hq, *q, *nv, *bs, *e, *red, *cum, *convw, *convb, *locfeat, *densew, *scratch;
};
__host__ __device__ inline size_t attn_lds_floats(int H, int M, int U, int S, int mode, int K, int F,
                                                  bool bwd) {
  size_t n = (size_t)H + 3 * U + S + 64;
  if (mode == 2) n += (size_t)(S + K) + (size_t)K * F + F + (size_t)S * F + (size_t)F * U;
  size_t scratch = (size_t)2 * M;                 // forward: context partials (<= 2 x M used)
  if (bwd) {
    size_t s2 = (size_t)M + 2 * S + 8 * U;        // dctx, dalign, de, per-wave dq/dnv partials
    if (mode == 2) s2 += (size_t)S * F + (size_t)F * U;   // dlocfeat, ddense
    scratch = s2;
  }
  return n + scratch + 64;
}
__device__ __forceinline__ AttnLds attn_lds_carve(float* base, int H, int U, int S, int mode, int K, int F) {
  AttnLds l;
  l.hq = base; base += H;
  l.q = base; base += U;
  l.nv = base; base += U;
  l.bs = base; base += U;
  l.e = base; base += S;
  l.red = base; base += 64;
  l.cum = l.convw = l.convb = l.locfeat = l.densew = nullptr;
  if (mode == 2) {
    l.cum = base; base += S + K;
    l.convw = base; base += K * F;
    l.convb = base; base += F;
    l.locfeat = base; base += S * F;
    l.densew = base; base += F * U;
  }
  l.scratch = base;
  return l;
}

// score parameters -> LDS: nv (normalised v for mode 1), bs (bias or zeros)
__device__ __forceinline__ void attn_load_score_params(const AdAttn& p, const AttnLds& l) {
  const int tid = threadIdx.x, U = p.U;
  float ss = 0.f;
  for (int u = tid; u < U; u += kAttnThreads) {
    const float v = p.v[u];
    l.nv[u] = v;
    ss += v * v;
    l.bs[u] = ((p.mode == 1 || (p.mode == 2 && p.use_bias)) && p.bias) ? p.bias[u] : 0.f;
  }
  if (p.mode == 1) {
    const float tot = block_sum(ss, l.red);
    const float sc = p.g[0] * rsqrtf(tot);
    for (int u = tid; u < U; u += kAttnThreads) l.nv[u] *= sc;
  }
  __syncthreads();
}

// location features of the cumulative alignments (state BEFORE step t) -> l.locfeat [S][F]
__device__ __forceinline__ void attn_location_features(const AdAttn& p, const AttnLds& l, int b) {
  const int tid = threadIdx.x, S = p.S, K = p.loc_k, F = p.loc_f, U = p.U;
  const int padl = (K - 1) / 2;
  const float* cum = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
  for (int i = tid; i < S + K; i += kAttnThreads) {
    const int s = i - padl;
    l.cum[i] = (s >= 0 && s < S) ? cum[s] : 0.f;
  }
  for (int i = tid; i < K * F; i += kAttnThreads) l.convw[i] = p.conv_w[i];
  for (int i = tid; i < F; i += kAttnThreads) l.convb[i] = p.conv_b[i];
  for (int i = tid; i < F * U; i += kAttnThreads) l.densew[i] = p.dense_w[i];
  __syncthreads();
  for (int i = tid; i < S * F; i += kAttnThreads) {
    const int s = i / F, f = i - s * F;
    float a = l.convb[f];
    for (int k = 0; k < K; ++k) a += l.cum[s + k] * l.convw[k * F + f];
    l.locfeat[i] = a;
  }
  __syncthreads();
}

// pre-activation of the score for (s, u pair) handled by this lane
__device__ __forceinline__ void attn_pre2(const AdAttn& p, const AttnLds& l, int b, int s, int u,
                                          float& x0, float& x1) {
  const uint32_t kv = *reinterpret_cast<const uint32_t*>(p.keys + ((long long)b * p.S + s) * p.U + u);
  x0 = bflo(kv) + l.q[u] + l.bs[u];
  x1 = bfhi(kv) + l.q[u + 1] + l.bs[u + 1];
  if (p.mode == 2) {
    const int F = p.loc_f;
    for (int f = 0; f < F; ++f) {
      const float lf = l.locfeat[s * F + f];
      x0 += lf * l.densew[f * p.U + u];
      x1 += lf * l.densew[f * p.U + u + 1];
    }
  }
}

// ------------------------------------------------------------------ attention forward
__global__ __launch_bounds__(kAttnThreads) void ad_attn_fwd_kernel(AdAttn p) {
  extern __shared__ float lds_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int H = p.H, M = p.M, U = p.U, S = p.S;
  const AttnLds l = attn_lds_carve(lds_raw, H, U, S, p.mode, p.loc_k, p.loc_f);
  const int slen = min(max(p.src_len[b], 0), S);
  // query input
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  for (int h8 = tid; h8 < H / 8; h8 += kAttnThreads) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(yq + h8 * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { l.hq[h8 * 8 + 2 * e] = bflo(v[e]); l.hq[h8 * 8 + 2 * e + 1] = bfhi(v[e]); }
  }
  attn_load_score_params(p, l);   // ends with a barrier
  // q[u] = hq . Wq[u, :]
  for (int u0 = wave * 4; u0 < U; u0 += 16) {
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int h = lane * 8; h < H; h += 512) {
      u32x4 wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        wv[i] = (u0 + i < U) ? *reinterpret_cast<const u32x4*>(p.wq + (long long)(u0 + i) * H + h)
                             : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          part[i] += bflo(wv[i][e]) * l.hq[h + 2 * e] + bfhi(wv[i][e]) * l.hq[h + 2 * e + 1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s = wave_sum(part[i]);
      if (lane == 0 && u0 + i < U) {
        l.q[u0 + i] = s;
        p.q_seq[((long long)b * p.T + p.t) * U + u0 + i] = s;
      }
    }
  }
  __syncthreads();
  if (p.mode == 2) attn_location_features(p, l, b);
  // scores
  for (int s = wave; s < slen; s += kAttnThreads / 64) {
    float part = 0.f;
    for (int u = 2 * lane; u < U; u += 128) {
      float x0, x1;
      attn_pre2(p, l, b, s, u, x0, x1);
      part += l.nv[u] * tanh_fast(x0) + l.nv[u + 1] * tanh_fast(x1);
    }
    part = wave_sum(part);
    if (lane == 0) l.e[s] = part;
  }
  __syncthreads();
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
  // context = sum_s align[s] * values[b, s, :]
  const int CG = min(M / 8, kAttnThreads);
  const int nsplit = min(kAttnThreads / CG, 2);
  float* part = l.scratch;   // [nsplit][M]
  for (int c0 = 0; c0 < M / 8; c0 += CG) {
    const int cg = c0 + tid % CG, sp = tid / CG;
    if (sp < nsplit && cg < M / 8) {
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bf16_t* vp = p.values + (long long)b * S * M + cg * 8;
#pragma unroll 4
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
  bf16_t* ctx = p.ctx + (long long)b * p.ctx_bs + (long long)p.t * p.ctx_ts;
  bf16_t* cat = p.cat0 + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc0;
  for (int m8 = tid; m8 < M / 8; m8 += kAttnThreads) {
    float c8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c8[e] = part[m8 * 8 + e];
      if (nsplit == 2) c8[e] += part[M + m8 * 8 + e];
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
}

// ------------------------------------------------------------------ d(attention input) GEMM
struct AdDattn {
  int B, M, K;
  const bf16_t* wT;     // [M rows, K] (first M rows of Wcat0^T)
  const bf16_t* dg;     // rows: dg + b*ld
  long long ld;
  float* out;           // [B, M]
};
__global__ __launch_bounds__(64 * kAdWaves) void ad_dattn_kernel(AdDattn p) {
  __shared__ float red[kAdWaves * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  f32x16 accw[1];
#pragma unroll
  for (int e = 0; e < 16; ++e) accw[0][e] = 0.f;
  tile_gemm_splitk<1, kAdWaves, 4>(p.wT, p.K, 0, j0, p.M, p.dg, p.ld, b0, p.B, p.K, accw);
  float acc[1][4];
  tile_reduce_quarters<1, kAdWaves>(accw, red, acc);
  if (wave >= 4) return;
  const int b = b0 + l31, j = j0 + 8 * wave + 4 * lhi;
  if (b >= p.B || j >= p.M) return;
  f32x4 o = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
  *reinterpret_cast<f32x4*>(p.out + (long long)b * p.M + j) = o;
}

// ------------------------------------------------------------------ attention backward
__global__ __launch_bounds__(kAttnThreads) void ad_attn_bwd_kernel(AdAttn p) {
  extern __shared__ float lds_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.tgt_len && p.t >= p.tgt_len[b]) return;
  const int H = p.H, M = p.M, U = p.U, S = p.S, F = p.loc_f, K = p.loc_k;
  const AttnLds l = attn_lds_carve(lds_raw, H, U, S, p.mode, K, F);
  float* dctx = l.scratch;            // [M]
  float* dal = dctx + M;              // [S]
  float* de = dal + S;                // [S]
  float* dqp = de + S;                // [4][U]
  float* dnvp = dqp + 4 * U;          // [4][U]
  float* dlocfeat = dnvp + 4 * U;     // [S][F]   (mode 2)
  float* ddense = dlocfeat + (p.mode == 2 ? S * F : 0);   // [F][U] (mode 2)
  const int slen = min(max(p.src_len[b], 0), S);
  const long long row = (long long)b * p.T + p.t;
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
  for (int u = tid; u < U; u += kAttnThreads) l.q[u] = p.q_seq[row * U + u];
  for (int s = tid; s < S; s += kAttnThreads) l.e[s] = p.align_seq[row * S + s];
  attn_load_score_params(p, l);   // barrier inside
  if (p.mode == 2) {
    attn_location_features(p, l, b);
    for (int i = tid; i < S * F; i += kAttnThreads) dlocfeat[i] = 0.f;
    for (int i = tid; i < F * U; i += kAttnThreads) ddense[i] = 0.f;
  }
  // dalign[s] = dctx . values[b,s,:] (+ carry of the cumulative-alignment state)
  for (int s = wave; s < slen; s += kAttnThreads / 64) {
    float part = 0.f;
    const bf16_t* vp = p.values + ((long long)b * S + s) * M;
    for (int m = lane * 8; m < M; m += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(vp + m);
#pragma unroll
      for (int e = 0; e < 4; ++e) part += bflo(v[e]) * dctx[m + 2 * e] + bfhi(v[e]) * dctx[m + 2 * e + 1];
    }
    part = wave_sum(part);
    if (lane == 0) dal[s] = part + (p.mode == 2 ? p.dcum[(long long)b * S + s] : 0.f);
  }
  __syncthreads();
  float dot = 0.f;
  for (int s = tid; s < slen; s += kAttnThreads) dot += l.e[s] * dal[s];
  dot = block_sum(dot, l.red);
  for (int s = tid; s < S; s += kAttnThreads) de[s] = s < slen ? l.e[s] * (dal[s] - dot) : 0.f;
  __syncthreads();
  // through the score: dpre[s,u] = de[s] * nv[u] * (1 - tanh^2)
  constexpr int UP = 4;   // u pairs per lane supported (U <= 512)
  float dq_acc[2 * UP], dnv_acc[2 * UP];
#pragma unroll
  for (int i = 0; i < 2 * UP; ++i) { dq_acc[i] = 0.f; dnv_acc[i] = 0.f; }
  for (int s = wave; s < slen; s += kAttnThreads / 64) {
    const float des = de[s];
    float dpre[2 * UP];
#pragma unroll
    for (int i = 0; i < UP; ++i) {
      const int u = 2 * lane + 128 * i;
      dpre[2 * i] = dpre[2 * i + 1] = 0.f;
      if (u < U) {
        float x0, x1;
        attn_pre2(p, l, b, s, u, x0, x1);
        const float t0 = tanh_fast(x0), t1 = tanh_fast(x1);
        dpre[2 * i] = des * l.nv[u] * (1.f - t0 * t0);
        dpre[2 * i + 1] = des * l.nv[u + 1] * (1.f - t1 * t1);
        dq_acc[2 * i] += dpre[2 * i];
        dq_acc[2 * i + 1] += dpre[2 * i + 1];
        dnv_acc[2 * i] += des * t0;
        dnv_acc[2 * i + 1] += des * t1;
        float* dk = p.dkeys + ((long long)b * S + s) * U + u;
        f32x2 kv = *reinterpret_cast<f32x2*>(dk);
        kv[0] += dpre[2 * i];
        kv[1] += dpre[2 * i + 1];
        *reinterpret_cast<f32x2*>(dk) = kv;
      }
    }
    if (p.mode == 2) {
      // dlocfeat[s,f] = sum_u dpre[s,u] * dense_w[f,u];  ddense[f,u] += locfeat[s,f] * dpre[s,u]
      float mine = 0.f;
      for (int f = 0; f < F; ++f) {
        float pf = 0.f;
        const float lf = l.locfeat[s * F + f];
#pragma unroll
        for (int i = 0; i < UP; ++i) {
          const int u = 2 * lane + 128 * i;
          if (u < U) {
            pf += dpre[2 * i] * l.densew[f * U + u] + dpre[2 * i + 1] * l.densew[f * U + u + 1];
            atomicAdd(&ddense[f * U + u], lf * dpre[2 * i]);
            atomicAdd(&ddense[f * U + u + 1], lf * dpre[2 * i + 1]);
          }
        }
        pf = wave_sum(pf);
        if (lane == f) mine = pf;
      }
      if (lane < F) dlocfeat[s * F + lane] = mine;   // F <= 64
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
  __syncthreads();
  for (int u = tid; u < U; u += kAttnThreads) {
    const float dq = dqp[u] + dqp[U + u] + dqp[2 * U + u] + dqp[3 * U + u];
    const float dn = dnvp[u] + dnvp[U + u] + dnvp[2 * U + u] + dnvp[3 * U + u];
    l.q[u] = dq;   // q is no longer needed: reuse as dq
    p.dq_seq[row * U + u] = f2bf(dq);
    p.dnv_acc[(long long)b * U + u] += dn;
  }
  __syncthreads();
  // dhq[h] = sum_u dq[u] * Wq[u,h]
  for (int h4 = tid; h4 < H / 4; h4 += kAttnThreads) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const bf16_t* wp = p.wq + h4 * 4;
#pragma unroll 8
    for (int u = 0; u < U; ++u) {
      const u32x2 w = *reinterpret_cast<const u32x2*>(wp + (long long)u * H);
      const float dq = l.q[u];
      a0 += dq * bflo(w[0]); a1 += dq * bfhi(w[0]); a2 += dq * bflo(w[1]); a3 += dq * bfhi(w[1]);
    }
    f32x4 o = {a0, a1, a2, a3};
    *reinterpret_cast<f32x4*>(p.dhq + (long long)b * H + h4 * 4) = o;
  }
  if (p.mode == 2) {
    const int padl = (K - 1) / 2;
    // gradient of the cumulative-alignment state before this step (carry to step t-1)
    for (int s2 = tid; s2 < S; s2 += kAttnThreads) {
      float a = p.dcum[(long long)b * S + s2];
      for (int k = 0; k < K; ++k) {
        const int s = s2 - k + padl;   // output position whose window touches s2 with tap k
        if (s < 0 || s >= S) continue;
        for (int f = 0; f < F; ++f) a += dlocfeat[s * F + f] * l.convw[k * F + f];
      }
      p.dcum[(long long)b * S + s2] = a;
    }
    for (int i = tid; i < K * F; i += kAttnThreads) {
      const int k = i / F, f = i - k * F;
      float a = 0.f;
      for (int s = 0; s < slen; ++s) a += l.cum[s + k] * dlocfeat[s * F + f];
      p.dconvw_acc[(long long)b * K * F + i] += a;
    }
    for (int f = tid; f < F; f += kAttnThreads) {
      float a = 0.f;
      for (int s = 0; s < slen; ++s) a += dlocfeat[s * F + f];
      p.dconvb_acc[(long long)b * F + f] += a;
    }
    for (int i = tid; i < F * U; i += kAttnThreads) p.ddense_acc[(long long)b * F * U + i] += ddense[i];
  }
}

// ------------------------------------------------------------------ cell backward
struct AdCellBwd {
  int B, T, H, t, last, KA, KB;
  float forget_bias;
  const int32_t* lens;
  const bf16_t* dy_ext; long long dy_bs, dy_ts;   // gradient of the (dropped) output or null
  const float* add32;                              // [B,H] more of the same or null
  const bf16_t* dgA; long long dgA_ld; const bf16_t* wAT; long long wAT_ld;   // upper layer, same step
  const bf16_t* dgB; long long dgB_ld; const bf16_t* wBT; long long wBT_ld;   // own layer, step t+1
  const bf16_t* gates;   // [B,T,4H]
  const float* c_seq;    // [B,T,H]
  float* dc_carry;       // [B,H]
  bf16_t* dg_out;        // [B,T,4H]
  float out_keep;
  unsigned long long out_seed;
};

__global__ __launch_bounds__(64 * kAdWaves) void ad_cell_bwd_kernel(AdCellBwd p) {
  __shared__ float red[kAdWaves * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int H = p.H;
  float accA[1][4] = {{0.f, 0.f, 0.f, 0.f}}, accB[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  if (p.KA > 0) {
    f32x16 accw[1];
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[0][e] = 0.f;
    tile_gemm_splitk<1, kAdWaves, 4>(p.wAT, p.wAT_ld, 0, j0, H, p.dgA, p.dgA_ld, b0, p.B, p.KA, accw);
    tile_reduce_quarters<1, kAdWaves>(accw, red, accA);
  }
  if (!p.last && p.KB > 0) {
    f32x16 accw[1];
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[0][e] = 0.f;
    tile_gemm_splitk<1, kAdWaves, 4>(p.wBT, p.wBT_ld, 0, j0, H, p.dgB, p.dgB_ld, b0, p.B, p.KB, accw);
    tile_reduce_quarters<1, kAdWaves>(accw, red, accB);
  }
  if (wave >= 4) return;
  const int b = b0 + l31;
  if (b >= p.B) return;
  if (p.lens && p.t >= p.lens[b]) return;
  const int j = j0 + 8 * wave + 4 * lhi;
  if (j >= H) return;
  const long long row = (long long)b * p.T + p.t;
  float dyv[4] = {accA[0][0], accA[0][1], accA[0][2], accA[0][3]};
  if (p.dy_ext) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p.dy_ext + (long long)b * p.dy_bs + (long long)p.t * p.dy_ts + j);
    dyv[0] += bflo(v[0]); dyv[1] += bfhi(v[0]); dyv[2] += bflo(v[1]); dyv[3] += bfhi(v[1]);
  }
  if (p.add32) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.add32 + (long long)b * H + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) dyv[e] += v[e];
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
    const float dh = dyv[e] + accB[0][e];
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

// ------------------------------------------------------------------ post-loop passes
// dmem[b,s,m] = sum_t align[b,t,s] * dctx[b,t,m]   (bf16 out, zero past src_len)
__global__ __launch_bounds__(256) void ad_dvalues_kernel(const float* __restrict__ align,
                                                         const bf16_t* __restrict__ dctx,
                                                         const int32_t* __restrict__ src_len,
                                                         const int32_t* __restrict__ tgt_len, int T,
                                                         int S, int M, bf16_t* __restrict__ dmem) {
  const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int slen = min(max(src_len[b], 0), S);
  const int tl = tgt_len ? min(max(tgt_len[b], 0), T) : T;
  for (int s0 = 0; s0 < S; s0 += 16) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (s0 < slen) {
      for (int t = 0; t < tl; ++t) {
        const float d = bf2f(dctx[((long long)b * T + t) * M + m]);
        const float* ar = align + ((long long)b * T + t) * S + s0;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += (s0 + i < S ? ar[i] : 0.f) * d;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (s0 + i < S) dmem[((long long)b * S + s0 + i) * M + m] = f2bf(s0 + i < slen ? acc[i] : 0.f);
  }
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
  vv = wave_sum(vv);
  dv_dot = wave_sum(dv_dot);
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

static size_t attn_lds_bytes(const os2s_attn_decoder_t* d, bool bwd) {
  return attn_lds_floats(d->H, d->M, d->U, d->S, d->score_mode, d->loc_k, d->loc_f, bwd) * sizeof(float);
}

static int ad_check(const os2s_attn_decoder_t* d) {
  OS2S_REQUIRE(d && d->B >= 1 && d->T >= 1 && d->S >= 1 && (d->L == 1 || d->L == 2));
  OS2S_REQUIRE(d->H % 8 == 0 && d->M % 8 == 0 && d->U % 128 == 0 && d->U <= 512);
  OS2S_REQUIRE(d->score_mode >= 0 && d->score_mode <= 2);
  OS2S_REQUIRE(d->t_begin >= 0 && d->t_begin <= d->t_end && d->t_end <= d->T);
  OS2S_REQUIRE(d->wcat[0] && d->wq && d->v && d->keys && d->values && d->src_len && d->gx0);
  OS2S_REQUIRE(d->cat[0] && d->c_seq[0] && d->align_seq && d->q_seq && d->y_top && d->ctx);
  if (d->L == 2) OS2S_REQUIRE(d->wcat[1] && d->cat[1] && d->c_seq[1]);
  if (d->score_mode == 1) OS2S_REQUIRE(d->g && d->b);
  if (d->score_mode == 2) {
    OS2S_REQUIRE(d->loc_k >= 1 && d->loc_f >= 1 && d->loc_f <= 64 && d->conv_w && d->conv_b && d->dense_w && d->cum_seq);
    if (d->use_bias) OS2S_REQUIRE(d->b);
  }
  if (attn_lds_bytes(d, true) > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  return OS2S_OK;
}

static void ad_fill_attn(const os2s_attn_decoder_t* d, AdAttn& a) {
  a.B = d->B; a.T = d->T; a.S = d->S; a.H = d->H; a.M = d->M; a.U = d->U;
  a.mode = d->score_mode; a.use_bias = d->use_bias; a.loc_k = d->loc_k; a.loc_f = d->loc_f;
  a.Kc0 = d->M + d->H;
  a.src_len = d->src_len; a.tgt_len = d->tgt_len;
  a.yq = (const bf16_t*)d->y_top; a.yq_bs = d->y_top_bs; a.yq_ts = d->y_top_ts;
  a.wq = (const bf16_t*)d->wq; a.keys = (const bf16_t*)d->keys; a.values = (const bf16_t*)d->values;
  a.v = d->v; a.g = d->g; a.bias = d->b; a.conv_w = d->conv_w; a.conv_b = d->conv_b; a.dense_w = d->dense_w;
  a.cum_seq = d->cum_seq; a.align_seq = d->align_seq; a.q_seq = d->q_seq;
  a.ctx = (bf16_t*)d->ctx; a.ctx_bs = d->ctx_bs; a.ctx_ts = d->ctx_ts;
  a.cat0 = (bf16_t*)d->cat[0];
  a.attn_in_keep = d->attn_in_keep; a.attn_in_seed = d->attn_in_seed;
  a.dctx_ext = nullptr; a.dattn = nullptr; a.dctx_seq = nullptr; a.dcum = nullptr; a.dkeys = nullptr;
  a.dq_seq = nullptr; a.dhq = nullptr; a.dnv_acc = nullptr; a.ddense_acc = nullptr;
  a.dconvw_acc = nullptr; a.dconvb_acc = nullptr; a.last = 0; a.dctx_bs = a.dctx_ts = 0;
}

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
  dim3 cgrid(ceil_div(H, 32), ceil_div(B, 32));
  for (int t = d->t_begin; t < d->t_end; ++t) {
    for (int l = 0; l < L; ++l) {
      AdCellFwd c;
      c.B = B; c.T = T; c.H = H; c.t = t; c.forget_bias = d->forget_bias; c.lens = d->tgt_len;
      c.Kc = l == 0 ? M + H : 2 * H;
      c.cat = (const bf16_t*)d->cat[l]; c.w = (const bf16_t*)d->wcat[l]; c.bias = d->bias[l];
      c.gx = l == 0 ? (const bf16_t*)d->gx0 : nullptr;
      c.c_seq = d->c_seq[l]; c.gates = (bf16_t*)d->gates[l];
      c.h_next = (bf16_t*)d->cat[l] + (l == 0 ? M : H);
      if (l == L - 1) { c.y = (bf16_t*)d->y_top; c.y_bs = d->y_top_bs; c.y_ts = d->y_top_ts; }
      else { c.y = (bf16_t*)d->cat[l + 1]; c.y_bs = (long long)(T + 1) * 2 * H; c.y_ts = 2 * H; }
      c.out_keep = d->out_keep; c.out_seed = d->out_seed[l];
      OS2S_LAUNCH(ad_cell_fwd_kernel, cgrid, dim3(64 * kAdWaves), 0, stream, c);
    }
    at.t = t;
    OS2S_LAUNCH(ad_attn_fwd_kernel, dim3(B), dim3(kAttnThreads), lds, stream, at);
  }
  return OS2S_OK;
}

extern "C" size_t os2s_attn_decoder_bwd_workspace_bytes(const os2s_attn_decoder_t* d) {
  if (!d) return 0;
  const size_t B = d->B;
  size_t n = B * d->M + B * d->H + 2 * B * d->H + B * d->S + B * d->U;
  if (d->score_mode == 2) n += B * d->loc_f * d->U + B * d->loc_k * d->loc_f + B * d->loc_f;
  return n * sizeof(float) + 1024;
}

extern "C" int os2s_attn_decoder_bwd(os2s_stream_t stream_, const os2s_attn_decoder_t* d,
                                     const os2s_attn_decoder_grads_t* gr, void* workspace,
                                     size_t workspace_bytes) {
  const int rc = ad_check(d);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(gr && workspace && d->gates[0] && gr->wcatT[0] && gr->dg[0] && gr->dq_seq && gr->dctx_seq && gr->dkeys && gr->dmem);
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
      hipFuncSetAttribute((const void*)ad_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
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
  float *ddense_acc = nullptr, *dconvw_acc = nullptr, *dconvb_acc = nullptr;
  if (d->score_mode == 2) {
    ddense_acc = ws; ws += (size_t)B * F * U;
    dconvw_acc = ws; ws += (size_t)B * K * F;
    dconvb_acc = ws; ws += (size_t)B * F;
  }
  // gate gradients of finished steps are zero; dkeys accumulates
  for (int l = 0; l < L; ++l)
    if (hipMemsetAsync(gr->dg[l], 0, (size_t)B * T * 4 * H * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (hipMemsetAsync(gr->dkeys, 0, (size_t)B * S * U * 4, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (hipMemsetAsync(gr->dq_seq, 0, (size_t)B * T * U * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (hipMemsetAsync(gr->dctx_seq, 0, (size_t)B * T * M * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;

  AdAttn at;
  ad_fill_attn(d, at);
  at.dctx_ext = (const bf16_t*)gr->dctx_ext; at.dctx_bs = gr->dctx_bs; at.dctx_ts = gr->dctx_ts;
  at.dattn = dattn; at.dctx_seq = (bf16_t*)gr->dctx_seq; at.dcum = dcum; at.dkeys = gr->dkeys;
  at.dq_seq = (bf16_t*)gr->dq_seq; at.dhq = dhq; at.dnv_acc = dnv_acc; at.ddense_acc = ddense_acc;
  at.dconvw_acc = dconvw_acc; at.dconvb_acc = dconvb_acc;
  const long long GH = 4LL * H;
  dim3 cgrid(ceil_div(H, 32), ceil_div(B, 32));
  for (int t = T - 1; t >= 0; --t) {
    const int last = (t == T - 1);
    if (!last) {
      AdDattn g;
      g.B = B; g.M = M; g.K = (int)GH; g.wT = (const bf16_t*)gr->wcatT[0];
      g.dg = (const bf16_t*)gr->dg[0] + (long long)(t + 1) * GH; g.ld = (long long)T * GH; g.out = dattn;
      OS2S_LAUNCH(ad_dattn_kernel, dim3(ceil_div(M, 32), ceil_div(B, 32)), dim3(64 * kAdWaves), 0, stream, g);
    }
    at.t = t; at.last = last;
    OS2S_LAUNCH(ad_attn_bwd_kernel, dim3(B), dim3(kAttnThreads), lds, stream, at);
    for (int l = L - 1; l >= 0; --l) {
      AdCellBwd c;
      c.B = B; c.T = T; c.H = H; c.t = t; c.last = last; c.forget_bias = d->forget_bias; c.lens = d->tgt_len;
      c.dy_ext = nullptr; c.dy_bs = c.dy_ts = 0; c.add32 = nullptr;
      c.dgA = nullptr; c.wAT = nullptr; c.KA = 0; c.dgA_ld = c.wAT_ld = 0;
      if (l == L - 1) {
        c.dy_ext = (const bf16_t*)gr->dy_top; c.dy_bs = gr->dy_top_bs; c.dy_ts = gr->dy_top_ts;
        c.add32 = dhq;
      } else {   // output feeds the layer above at the same step (columns 0..H of its cat)
        c.dgA = (const bf16_t*)gr->dg[l + 1] + (long long)t * GH; c.dgA_ld = (long long)T * GH;
        c.wAT = (const bf16_t*)gr->wcatT[l + 1]; c.wAT_ld = GH; c.KA = (int)GH;
      }
      c.dgB = (const bf16_t*)gr->dg[l] + (long long)(t + 1) * GH; c.dgB_ld = (long long)T * GH;
      c.wBT = (const bf16_t*)gr->wcatT[l] + (long long)(l == 0 ? M : H) * GH; c.wBT_ld = GH; c.KB = (int)GH;
      c.gates = (const bf16_t*)d->gates[l]; c.c_seq = d->c_seq[l]; c.dc_carry = dcc[l];
      c.dg_out = (bf16_t*)gr->dg[l]; c.out_keep = d->out_keep; c.out_seed = d->out_seed[l];
      OS2S_LAUNCH(ad_cell_bwd_kernel, cgrid, dim3(64 * kAdWaves), 0, stream, c);
    }
  }
  OS2S_LAUNCH(ad_dvalues_kernel, dim3(ceil_div(M, 256), B), dim3(256), 0, stream, d->align_seq,
              (const bf16_t*)gr->dctx_seq, d->src_len, d->tgt_len, T, S, M, (bf16_t*)gr->dmem);
  OS2S_LAUNCH(ad_score_vec_grads_kernel, dim3(1), dim3(256), 0, stream, dnv_acc, B, U, d->score_mode,
              d->v, d->g, gr->dv, gr->dg_scalar);
  if (d->score_mode == 2) {
    OS2S_LAUNCH(ad_reduce_rows_kernel, dim3(ceil_div((long long)F * U, 256)), dim3(256), 0, stream,
                ddense_acc, B, (long long)F * U, gr->ddense_w);
    OS2S_LAUNCH(ad_reduce_rows_kernel, dim3(ceil_div((long long)K * F, 256)), dim3(256), 0, stream,
                dconvw_acc, B, (long long)K * F, gr->dconv_w);
    OS2S_LAUNCH(ad_reduce_rows_kernel, dim3(ceil_div(F, 256)), dim3(256), 0, stream, dconvb_acc, B,
                (long long)F, gr->dconv_b);
  }
  return OS2S_OK;
}
