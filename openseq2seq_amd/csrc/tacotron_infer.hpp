// Free-running Tacotron2 decoding (eval / infer) and the small-code kernels of the location-sensitive decoder's
// training pass, gfx950. Included at the end of attn_decoder.hip (one translation unit: the training driver
// launches these kernels, and they share its argument structs).
//
// Reference: Tacotron2Decoder._decode in eval / infer mode (open_seq2seq/decoders/tacotron2_decoder.py:378-428)
// = tf.contrib.seq2seq.dynamic_decode(TacotronDecoder(helper = TacotronHelper), impute_finished = False,
// maximum_iterations = 10 * max(src_len)); per step (parts/tacotron/tacotron_decoder.py:153-190,
// tacotron_helper.py:138-226):
//   x_t     = prenet(frame_{t-1})            2 x (Dense + ReLU + dropout(0.5), ALWAYS on; frame_{-1} = 0)
//   h0, h1  = LSTMCell stack on [x_t, attention_{t-1}], state
//   a_t     = location-sensitive attention(query = h1, cumulative alignments); attention_t = sum_s a_t[s] values[s]
//   frame_t = W_out [h1, attention_t] + b;   stop_t = W_stop frame_t + b
//   finished |= round(sigmoid(stop_t)) (mask_decoder_sequence); the loop ends when every sample has finished
//
// The training pass hoists everything that does not depend on the previous step out of the loop; here EVERYTHING
// depends on the previous frame, so a step is a chain of dependent launches and what matters is their number and
// the dependent memory round trips inside each. One step = FOUR launches, no host interaction:
//   ti_lstm_kernel (layer 0)   gates = [x_t | attention_{t-1} | h0_{t-1}] . W0x^T — the pre-net columns are part of
//                              the streamed matrix (no separate input-projection GEMM); 16 gate rows (4 units) x
//                              all samples per workgroup = H/4 workgroups, MFMA 16x16x32 with the reduction cut
//                              over 8 or 9 waves, EVERY load of a wave (weights as e4m3 or bf16, inputs straight from
//                              the row-major state rows — no LDS staging) in flight before its first MFMA
//   ti_lstm_kernel (layer 1)
//   ti_scores_kernel           query projection + partial scores (location filter on the matrix cores), 4 unit parts
//                              x B, and as a fifth part per sample W_out[:, :H] h1 (the half of the frame that only
//                              needs the cell output)
//   ti_context_kernel          softmax, alignments, context columns (4 or 8 parts x B: MFMA weighted sums over the
//                              transposed memory) and, as one more part per sample, the frame: W_out[:, :H] h1 +
//                              sum_s a[s] PV[s] + b with PV = values W_out[:, H:]^T computed ONCE per batch (the
//                              context half of the projection commutes with the attention sum), stop token,
//                              finished / length bookkeeping, and the pre-net of the NEXT step
// The stop decision stays on the device: the launch that sees the last sample finish writes the step count to
// state[1]; every later launch returns at once, so the host may enqueue steps ahead and poll every N steps —
// the result does not depend on N.
#pragma once

namespace os2s {

// ti_lstm_kernel's template geometry: waves per workgroup (the reduction split) x 64-wide k chunks ("request slots")
// per wave and round; ti_geom() picks 8 x 4, 9 x 4 or 8 x 5 by K.
// OS2S_TI_NT_WEIGHTS (build flag, experiment): non-temporal weight loads — measured no change.

#ifdef OS2S_TI_NT_WEIGHTS
#define TI_WLOAD(p) __builtin_nontemporal_load(p)
#else
#define TI_WLOAD(p) (*(p))
#endif

struct TiLstm {
  int B, H, K, Ka;                   // K = Ka + Kb input columns
  const bf16_t* in_a; long long lda; // row b: in_a + b * lda  (Ka columns; Ka == 0: unused)
  const bf16_t* in_b; long long ldb; // row b: in_b + b * ldb  (K - Ka columns)
  const void* w;                     // [4H, K] e4m3 (FP8) or bf16
  const float* scale;                // [4H] row scales (FP8)
  const float* bias;                 // [4H] or null
  float forget_bias;
  const float* c_prev; long long ldc_prev;   // row b at c_prev + b * ldc_prev, or null (zeros)
  float* c_out; long long ldc_out;
  bf16_t* h1; long long ldh1;        // h destinations (row b at h + b * ld; either may be null)
  bf16_t* h2; long long ldh2;        //   h2 takes the output dropout (training: the cell's OUTPUT, not its state)
  const int32_t* state;              // state[1] != 0: decoding has ended (null: no stop flag — the training pass)
  // training pass (os2s_attn_decoder_fwd): input projection of the step, saved gates, output dropout
  const bf16_t* gx; long long ldgx;  // row b: gx + b * ldgx, [4H] (or null)
  bf16_t* gates; long long ldgates;  // row b: gates + b * ldgates, [4H] = i, f, g, o activations (or null)
  float out_keep; unsigned long long out_seed; long long drop_t, drop_T;   // element index ((b * T + t) * H + j)
};

// rows of a 16-row tile: r = 4 * unit + gate, so that after the MFMA (acc[i] = row 4 * (lane >> 4) + i,
// column lane & 15) a lane holds the four gates of ONE (unit, sample). MT row tiles (4 * MT units) x NT
// 16-sample column tiles per workgroup.
template <bool FP8, int MT, int NT, int kTiWaves, int kTiCpw>
__global__ __launch_bounds__(64 * kTiWaves) void ti_lstm_kernel(TiLstm p) {
  __shared__ float red[kTiWaves * MT * NT * 4 * 64];
  const int done = p.state ? p.state[1] : 0;     // consumed after the loads are in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // an SGPR: the chunk tests below are scalar
  const int r = lane & 15, q = lane >> 4;
  const int H = p.H, K = p.K;
  const int j0 = blockIdx.x * 4 * MT;
  // epilogue operands of thread (mt, nt, u, n) first: their round trip overlaps the weight stream. Loads are
  // UNCONDITIONAL (clamped addresses, selected afterwards): a load under a branch is waited for at the join
  const int e_mt = wave / NT, e_nt = wave % NT, e_u = 4 * e_mt + (lane >> 4), e_n = lane & 15;
  const int e_b = e_nt * 16 + e_n, e_j = j0 + e_u;
  const bool e_live = wave < MT * NT && e_b < p.B && e_j < H;
  float e_sc[4], e_bias[4], e_c, e_gx[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const int cj = min(e_j, H - 1), cb = min(e_b, p.B - 1);
    // (placeholders for absent operands: any readable 16 H bytes — the weight matrix itself)
    const float* scp = FP8 ? p.scale : reinterpret_cast<const float*>(p.w);
    const float* bip = p.bias ? p.bias : scp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      e_sc[g] = scp[g * H + cj];
      e_bias[g] = bip[g * H + cj];
    }
    if (p.gx) {                                              // uniform branch; its loads join the same round
#pragma unroll
      for (int g = 0; g < 4; ++g) e_gx[g] = bf2f(p.gx[(long long)cb * p.ldgx + g * H + cj]);
    }
    e_c = (p.c_prev ? p.c_prev : p.c_out)[(long long)cb * (p.c_prev ? p.ldc_prev : p.ldc_out) + cj];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (!FP8) e_sc[g] = 1.f;
      if (!p.bias) e_bias[g] = 0.f;
    }
    if (!p.c_prev) e_c = 0.f;
  }
  // operand rows of this lane
  const uint8_t* w8[MT];
  const bf16_t* w16[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int wrow = (r & 3) * H + min(j0 + 4 * mt + (r >> 2), H - 1);
    w8[mt] = reinterpret_cast<const uint8_t*>(p.w) + (long long)wrow * K;
    w16[mt] = reinterpret_cast<const bf16_t*>(p.w) + (long long)wrow * K;
  }
  const bf16_t* ia[NT];
  const bf16_t* ib[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int b = min(nt * 16 + r, p.B - 1);          // padding columns re-read the last sample (discarded)
    ia[nt] = p.Ka > 0 ? p.in_a + (long long)b * p.lda : p.in_b + (long long)b * p.ldb;
    ib[nt] = p.in_b + (long long)b * p.ldb - p.Ka;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = K >> 6;
  for (int base = 0; base < nchunks; base += kTiWaves * kTiCpw) {
    u32x4 wa[kTiCpw][MT], wb[kTiCpw][MT];              // FP8: wa holds the 16 bytes, wb unused
    u32x4 va[kTiCpw][NT], vb[kTiCpw][NT];
#pragma unroll
    for (int i = 0; i < kTiCpw; ++i) {
      const int c = min(base + i * kTiWaves + wave, nchunks - 1);
      const int k = c * 64 + q * 16;                  // this lane's 16 consecutive k of the chunk
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (FP8) wa[i][mt] = TI_WLOAD(reinterpret_cast<const u32x4*>(w8[mt] + k));
        else {
          wa[i][mt] = TI_WLOAD(reinterpret_cast<const u32x4*>(w16[mt] + k));
          wb[i][mt] = TI_WLOAD(reinterpret_cast<const u32x4*>(w16[mt] + k + 8));
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16_t* src = (k < p.Ka ? ia[nt] : ib[nt]) + k;
        va[i][nt] = *reinterpret_cast<const u32x4*>(src);
        vb[i][nt] = *reinterpret_cast<const u32x4*>(src + 8);
      }
    }
    // every load of the round is in flight before the first MFMA (the scheduler otherwise sinks the loads
    // next to their uses to save registers: one exposed round trip per chunk instead of one per round)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kTiCpw; ++i) {
      const bool live = base + i * kTiWaves + wave < nchunks;      // scalar; a dead chunk multiplies zeros
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4 a0, a1;
        if (FP8) {
          const u32x4 w = wa[i][mt];
          a0[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[0], 1.0f, false));
          a0[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[0], 1.0f, true));
          a0[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[1], 1.0f, false));
          a0[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[1], 1.0f, true));
          a1[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[2], 1.0f, false));
          a1[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[2], 1.0f, true));
          a1[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[3], 1.0f, false));
          a1[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[3], 1.0f, true));
        } else {
          a0 = wa[i][mt];
          a1 = wb[i][mt];
        }
        a0 = live ? a0 : zero;
        a1 = live ? a1 : zero;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0),
                                                                __builtin_bit_cast(bf16x8, va[i][nt]), acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1),
                                                                __builtin_bit_cast(bf16x8, vb[i][nt]), acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  if (done != 0) return;               // decoding has ended (uniform): nothing is written
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(((wave * MT + mt) * NT + nt) * 4 + i) * 64 + lane] = acc[mt][nt][i];
  __syncthreads();
  if (!e_live) return;
  // thread (mt, nt, l): unit 4 * mt + (l >> 4), sample column l & 15 — the accumulator lane with the same (q, r)
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kTiWaves; ++w) s += red[(((w * MT + e_mt) * NT + e_nt) * 4 + g) * 64 + lane];
    pre[g] = s * e_sc[g] + e_bias[g] + e_gx[g];
  }
  // (tanh through one exp + one reciprocal instead of the library tanhf's several dozen instructions per call: see
  // "score launch" below; |error| ~1e-7 against bf16 outputs)
  const float ig = sigmoidf_(pre[0]), gg = tanh_fast(pre[1]);
  const float fg = sigmoidf_(pre[2] + p.forget_bias), og = sigmoidf_(pre[3]);
  const float cn = e_c * fg + ig * gg;
  float hv = tanh_fast(cn) * og;
  p.c_out[(long long)e_b * p.ldc_out + e_j] = cn;
  if (p.gates) {
    bf16_t* gp = p.gates + (long long)e_b * p.ldgates + e_j;
    gp[0] = f2bf(ig); gp[H] = f2bf(fg); gp[2 * H] = f2bf(gg); gp[3 * H] = f2bf(og);
  }
  if (p.h1) p.h1[(long long)e_b * p.ldh1 + e_j] = f2bf(hv);
  if (p.out_keep < 1.f) {
    const unsigned long long idx = ((unsigned long long)e_b * p.drop_T + p.drop_t) * H + e_j;
    const uint32_t bits = dropout_bits8(p.out_seed, idx >> 3, p.out_keep);
    hv = ((bits >> (e_j & 7)) & 1u) ? hv / p.out_keep : 0.f;
  }
  if (p.h2) p.h2[(long long)e_b * p.ldh2 + e_j] = f2bf(hv);
}

typedef __attribute__((ext_vector_type(2))) __bf16 ti_bf2;
__device__ __forceinline__ float ti_dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ti_bf2, a), __builtin_bit_cast(ti_bf2, b), c, false);
}
__device__ __forceinline__ float ti_dot8(const u32x4& a, const u32x4& b, float c) {
  c = ti_dot2(a[0], b[0], c);
  c = ti_dot2(a[1], b[1], c);
  c = ti_dot2(a[2], b[2], c);
  return ti_dot2(a[3], b[3], c);
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_mov<0xB1, 0xf>(0.f, x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E, 0xf>(0.f, x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x124, 0xf>(0.f, x);   // row_ror:4
  x += dpp_mov<0x128, 0xf>(0.f, x);   // row_ror:8
  return x;
}


// ---- context + frame ----------------------------------------------------------------------------------
struct TiTail {
  int P, n_mel, mask_seq, first;       // first = 1: only the pre-net of step 0 (frame = 0) runs
  int dbg;                             // profiling aid (OS2S_TI_DEBUG): bit 0 / 1 skip the frame / score parts of the score
                                       // launch, bit 2 / 3 the frame / context parts of the context launch (WRONG results)
  float keep; unsigned long long seed[2];
  const bf16_t* wp1; const float* bp1; // [P, n_mel], [P]
  const bf16_t* wp2; const float* bp2; // [P, P], [P]
  const bf16_t* wout_h;                // [n_mel, H]
  const bf16_t* pv_t;                  // [B, n_mel16, Sp] bf16: (values W_out[:, H:]^T)^T, positions contiguous
  const bf16_t* values_t;              // [B, M, Sp] bf16: values^T
  const float* bout;                   // [n_mel]
  const bf16_t* wstop; const float* bstop;   // [n_mel], [1]
  float* mh;                           // [B, n_mel] scratch: W_out[:, :H] h1 of the step (ti_scores_kernel -> ti_context_kernel)
  bf16_t* x_seq;                       // [B, T+1, P]
  bf16_t* mel;                         // [B, T, n_mel]
  float* stop;                         // [B, T]
  int32_t* state;                      // [0] unused, [1] steps at which everything had finished (0 = running),
                                       // [2] finished samples, [3] unused, [4 .. 4+B) finished, [4+B .. 4+2B) lengths
};

constexpr int kTiCtxThreads = 512;
constexpr int kTiW2Rows = 16;          // W2 rows per thread (P <= 256: P * P / 8 sixteen-byte pieces over 512 threads)
constexpr int kTiW1Pieces = 8;         // W1 pieces per thread
constexpr int kTiKs = 8;               // 32-position steps whose operands are requested up front (S <= 256)

__host__ __device__ inline int ti_spad(int S) { return (S + 31) & ~31; }      // row pitch of values_t / pv_t

__host__ __device__ inline size_t ti_ctx_lds_floats(int S, int P, int n_mel) {
  // e [Sp] + red [32] + hi / lo alignments (bf16, max(Sp, 256) each) + frame [n_mel] + x1 [P] + mc [n_mel + 16] +
  // partials max(P * (P / 8 + 1), P * 8)
  const size_t Sp = ti_spad(S), Sa = Sp > 256 ? Sp : 256;
  size_t part = (size_t)P * (P / 8 + 1);
  if ((size_t)P * 8 > part) part = (size_t)P * 8;
  return Sp + 32 + Sa + n_mel + P + n_mel + 16 + part + 64;
}

// Softmax over the summed partial scores -> alignments: e[s] (fp32, zero past the length) and their bf16 hi / lo
// halves ah / al (zero up to max(Sp, 256): the A operand of the weighted sums below). Every part of a sample
// computes them; `store` (part 0) writes the alignment row and advances the cumulative alignments.
__device__ __forceinline__ void ti_alignments(const AdAttn& p, const AdLoc& x, int b, int slen, float* e, float* red,
                                              uint16_t* ah, uint16_t* al, bool store) {
  const int tid = threadIdx.x, S = p.S, Sa = max(ti_spad(S), 256);
  const float* ep = x.e_part + (long long)b * kLocParts * S;
  float mx = -INFINITY;
  for (int sp = tid; sp < slen; sp += kTiCtxThreads) {
    float v = ep[sp];
#pragma unroll
    for (int k = 1; k < kLocParts; ++k) v += ep[k * S + sp];
    e[sp] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max_dpp(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < kTiCtxThreads / 64; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
  for (int sp = tid; sp < slen; sp += kTiCtxThreads) {
    const float ex = __expf(e[sp] - mx);
    e[sp] = ex;
    sum += ex;
  }
  sum = wave_sum_dpp(sum);
  if ((tid & 63) == 0) red[8 + (tid >> 6)] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < kTiCtxThreads / 64; ++w) sum += red[8 + w];
  const float inv = slen > 0 ? 1.f / sum : 0.f;
  const long long row = (long long)b * p.T + p.t;
  for (int sp = tid; sp < Sa; sp += kTiCtxThreads) {
    const float a = sp < slen ? e[sp] * inv : 0.f;
    const bf16_t hi = f2bf(a);
    ah[sp] = hi;
    al[sp] = f2bf(a - bf2f(hi));
    if (store && sp < S) {
      p.align_seq[row * S + sp] = a;
      const long long ci = ((long long)b * (p.T + 1) + p.t) * S + sp;
      p.cum_seq[ci + S] = p.cum_seq[ci] + a;
    }
  }
  __syncthreads();
}

// out[n] = sum_s a[s] mat_t[row0 + n][s] for the 16 rows of one tile of a TRANSPOSED operand (bf16 [rows, Sp],
// positions contiguous): the alignments are the A operand of a 16x16x32 MFMA (every A row the same, bf16 hi + lo),
// 16 positions-contiguous bytes per lane are the B operand. Result for column n = lane & 15 in every lane.
// breq: this lane's operands of the first kTiKs position steps, requested by the caller before the softmax.
__device__ __forceinline__ float ti_weighted_sum(const bf16_t* __restrict__ mrow, int Sp, const u32x4 (&breq)[kTiKs],
                                                 const uint16_t* ah, const uint16_t* al) {
  const int kb = (threadIdx.x & 63) >> 4;
  const int nks = Sp >> 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < kTiKs; ++ks) {
    // steps past nks re-read the last step's operand against zero alignments (ah / al are zero up to 256)
    const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(ah + ks * 32 + kb * 8);
    const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(al + ks * 32 + kb * 8);
    const bf16x8 bv = __builtin_bit_cast(bf16x8, breq[ks]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bv, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, bv, acc, 0, 0, 0);
  }
#pragma unroll 1
  for (int ks = kTiKs; ks < nks; ++ks) {           // S > 256: the rest, not requested ahead
    const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(ah + ks * 32 + kb * 8);
    const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(al + ks * 32 + kb * 8);
    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(mrow + ks * 32 + kb * 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bv, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, bv, acc, 0, 0, 0);
  }
  return acc[0];                                   // rows are identical: row 4 * kb
}

__device__ __forceinline__ void ti_request_rows(const bf16_t* __restrict__ mrow, int Sp, u32x4 (&breq)[kTiKs]) {
  const int kb = (threadIdx.x & 63) >> 4;
  const int nks = Sp >> 5;
#pragma unroll
  for (int ks = 0; ks < kTiKs; ++ks)
    breq[ks] = *reinterpret_cast<const u32x4*>(mrow + min(ks, nks - 1) * 32 + kb * 8);
}

// The frame part of a sample: one workgroup, ONE dependent chain (alignments -> frame -> stop token -> pre-net
// layer 1 -> layer 2), so everything that does not depend on the chain is requested first: both pre-net
// matrices (168 KB) and this sample's rows of PV^T sit in registers before the softmax starts.
__device__ __forceinline__ void ti_tail(const AdAttn& p, const AdLoc& x, const TiTail& q, float* lds) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.S, P = q.P, nm = q.n_mel, B = p.B, T = p.T;
  const int Sp = ti_spad(S), Sa = max(Sp, 256), nm16 = (nm + 15) & ~15;
  float* e = lds;                      // [Sp]
  float* red = e + Sp;                 // [32]
  uint16_t* ah = reinterpret_cast<uint16_t*>(red + 32);    // [Sa] bf16
  uint16_t* al = ah + Sa;                                  // [Sa]
  uint32_t* frp = reinterpret_cast<uint32_t*>(al + Sa);    // [nm / 2] the frame as packed bf16 pairs (nm floats reserved)
  uint32_t* x1p = frp + nm;            // [P / 2] pre-net layer-1 output, packed bf16 (P floats reserved)
  float* mc = reinterpret_cast<float*>(x1p + P);            // [nm16] context half of the frame
  float* part = mc + nm + 16;          // partial sums of the pre-net stages
  uint16_t* fr16 = reinterpret_cast<uint16_t*>(frp);
  uint16_t* x116 = reinterpret_cast<uint16_t*>(x1p);
  const int t_next = q.first ? 0 : p.t + 1;
  const int slen = q.first ? 0 : min(max(p.src_len[b], 0), S);
  // ---- requests ---------------------------------------------------------------------------------------
  // W2 [P, P]: thread = (piece pc of a row, row group rg); rows rg, rg + RG, ...
  const int pcs2 = P >> 3, RG = kTiCtxThreads / pcs2;
  const int pc2 = tid % pcs2, rg = tid / pcs2;
  u32x4 w2[kTiW2Rows];
#pragma unroll
  for (int i = 0; i < kTiW2Rows; ++i) {
    const int j = min(rg + i * RG, P - 1);
    w2[i] = *reinterpret_cast<const u32x4*>(q.wp2 + (long long)j * P + pc2 * 8);
  }
  // W1 [P, nm]: TPR threads per row, each pieces pc, pc + TPR, ...
  const int pcs1 = nm >> 3, TPR = kTiCtxThreads / P;          // P <= 256: TPR >= 2
  const int j1 = tid / TPR, s1 = tid % TPR;
  u32x4 w1[kTiW1Pieces];
#pragma unroll
  for (int i = 0; i < kTiW1Pieces; ++i) {
    const int pc = min(s1 + i * TPR, pcs1 - 1);
    w1[i] = *reinterpret_cast<const u32x4*>(q.wp1 + (long long)min(j1, P - 1) * nm + pc * 8);
  }
  // PV^T rows of this sample: wave w < nm16 / 16 owns frame columns 16 w ... (the other waves re-read tile 0)
  const int ct = wave < nm16 / 16 ? wave : 0;
  const bf16_t* pvrow = q.pv_t + ((long long)b * nm16 + ct * 16 + (lane & 15)) * Sp;
  u32x4 breq[kTiKs];
  ti_request_rows(pvrow, Sp, breq);
  float mh = 0.f, bo = 0.f;
  if (!q.first && tid < nm) { mh = q.mh[(long long)b * nm + tid]; bo = q.bout[tid]; }
  __builtin_amdgcn_sched_barrier(0);
  if (q.first) {
    for (int k = tid; k < nm; k += kTiCtxThreads) fr16[k] = 0;
    __syncthreads();
  } else {
    ti_alignments(p, x, b, slen, e, red, ah, al, false);
    // ---- context half of the frame: sum_s a[s] PV[s, :] ------------------------------------------------------
    const float cs = ti_weighted_sum(pvrow, Sp, breq, ah, al);
    if (wave < nm16 / 16 && lane < 16) mc[wave * 16 + lane] = cs;
    __syncthreads();
    if (tid < nm) {
      const bf16_t fb = f2bf(mh + bo + mc[tid]);
      q.mel[((long long)b * T + p.t) * nm + tid] = fb;
      fr16[tid] = fb;
    }
    __syncthreads();
    // ---- stop token, finished / length bookkeeping (wave 0; the others go on) ----------------------------------
    if (tid < 64) {
      float sacc = 0.f;
      for (int k = tid; k < nm / 2; k += 64) sacc = ti_dot2(reinterpret_cast<const uint32_t*>(q.wstop)[k], frp[k], sacc);
      sacc = wave_sum_dpp(sacc);
      if (tid == 0) {
        sacc = bf2f(f2bf(sacc + q.bstop[0]));            // the stop projection's output tensor is bf16
        q.stop[(long long)b * T + p.t] = sacc;
        int32_t* fin = q.state + 4;
        int32_t* len = q.state + 4 + B;
        const int was = fin[b];
        if (!was) len[b] = p.t + 1;                      // dynamic_decode: lengths count the step that finished
        // round(sigmoid(s)) == 1  <=>  sigmoid(s) > 0.5  <=>  s > 0  (round half to even: 0.5 -> 0)
        if (q.mask_seq && !was && sacc > 0.f) {
          fin[b] = 1;
          const int n = atomicAdd(&q.state[2], 1);
          if (n == B - 1) q.state[1] = p.t + 1;          // visible to the next launch (kernel boundary)
        }
      }
    }
  }
  if (t_next > T) return;
  // ---- pre-net of the next step ---------------------------------------------------------------------------
  const float ik = 1.f / q.keep;
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kTiW1Pieces; ++i) {
      const int pc = min(s1 + i * TPR, pcs1 - 1);
      const float d = ti_dot8(w1[i], *reinterpret_cast<const u32x4*>(frp + pc * 4), 0.f);
      s += s1 + i * TPR < pcs1 ? d : 0.f;
    }
    if (j1 < P) part[j1 * 8 + s1] = s;                   // TPR <= 8
  }
  __syncthreads();
  if (tid < P) {
    float s = q.bp1[tid];
    for (int i = 0; i < TPR; ++i) s += part[tid * 8 + i];
    s = fmaxf(s, 0.f);
    if (q.keep < 1.f) {
      const unsigned long long idx = ((unsigned long long)t_next * B + b) * P + tid;
      const uint32_t bits = dropout_bits8(q.seed[0], idx >> 3, q.keep);
      s = ((bits >> (idx & 7)) & 1u) ? s * ik : 0.f;
    }
    x116[tid] = f2bf(s);                // the layer's output is a bf16 tensor in the teacher-forced pass too
  }
  __syncthreads();
  {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(x1p + pc2 * 4);
#pragma unroll
    for (int i = 0; i < kTiW2Rows; ++i) {
      const int j = rg + i * RG;
      const float d = ti_dot8(w2[i], xv, 0.f);
      if (j < P && rg < RG) part[j * (pcs2 + 1) + pc2] = d;
    }
  }
  __syncthreads();
  if (tid < P) {
    float s = q.bp2[tid];
    for (int i = 0; i < pcs2; ++i) s += part[tid * (pcs2 + 1) + i];
    s = fmaxf(s, 0.f);
    if (q.keep < 1.f) {
      const unsigned long long idx = ((unsigned long long)t_next * B + b) * P + tid;
      const uint32_t bits = dropout_bits8(q.seed[1], idx >> 3, q.keep);
      s = ((bits >> (idx & 7)) & 1u) ? s * ik : 0.f;
    }
    q.x_seq[((long long)b * (T + 1) + t_next) * P + tid] = f2bf(s);
  }
}

// grid (ctx_parts + 1, B), 512 threads: part c < ctx_parts = context columns [c * MQ, (c + 1) * MQ) (a 16-column
// tile per wave: attention_t[m] = sum_s a[s] values^T[m][s] on the matrix cores); the last part is the frame
__global__ __launch_bounds__(kTiCtxThreads) void ti_context_kernel(AdAttn p, AdLoc x, TiTail q, int ctx_parts,
                                                                   int MQ) {
  extern __shared__ float lds_raw[];
  const int cpart = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.M, S = p.S;
  // "decoding has ended" for THIS launch means: ended at an EARLIER step (state[1] = t' + 1 <= t). The tail
  // workgroup of the last sample to finish writes state[1] = t + 1 during this very launch; a workgroup that
  // is scheduled after that write (more workgroups than CUs, a partitioned GPU, a co-running stream) must not
  // take it as a reason to skip step t — the host counts step t, its rows have to be complete.
  const int ended = q.state[1];
  const bool over = ended != 0 && ended <= p.t;
  if (cpart == ctx_parts) {
    if (!q.first && (over || (q.dbg & 4))) return;
    ti_tail(p, x, q, lds_raw);
    return;
  }
  if (q.first || over || (q.dbg & 8)) return;
  const int Sp = ti_spad(S), Sa = max(Sp, 256);
  float* e = lds_raw;                  // [Sp]
  float* red = e + Sp;                 // [32]
  uint16_t* ah = reinterpret_cast<uint16_t*>(red + 32);
  uint16_t* al = ah + Sa;
  const int slen = min(max(p.src_len[b], 0), S);
  const int ntile = MQ >> 4;           // 16-column tiles of this part: wave w owns tiles w and w + 8 (requested up
  const int m0 = cpart * MQ;           // front), further ones (M > 2048 / parts) in a rolled loop
  constexpr int NW = kTiCtxThreads / 64;
  const int cta = wave < ntile ? wave : 0, ctb = wave + NW < ntile ? wave + NW : cta;
  const bf16_t* vbase = q.values_t + ((long long)b * M + m0 + (lane & 15)) * Sp;
  u32x4 breq_a[kTiKs], breq_b[kTiKs];
  ti_request_rows(vbase + (long long)cta * 16 * Sp, Sp, breq_a);
  ti_request_rows(vbase + (long long)ctb * 16 * Sp, Sp, breq_b);
  __builtin_amdgcn_sched_barrier(0);
  ti_alignments(p, x, b, slen, e, red, ah, al, cpart == 0);
  bf16_t* ctx = p.ctx + (long long)b * p.ctx_bs + (long long)p.t * p.ctx_ts;
  bf16_t* cat = p.cat0 + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc0;
  {
    const float ca = ti_weighted_sum(vbase + (long long)cta * 16 * Sp, Sp, breq_a, ah, al);
    const float cb = ti_weighted_sum(vbase + (long long)ctb * 16 * Sp, Sp, breq_b, ah, al);
    if (lane < 16) {
      if (wave < ntile) { const bf16_t v = f2bf(ca); ctx[m0 + wave * 16 + lane] = v; cat[m0 + wave * 16 + lane] = v; }
      if (wave + NW < ntile) {
        const bf16_t v = f2bf(cb);
        ctx[m0 + (wave + NW) * 16 + lane] = v;
        cat[m0 + (wave + NW) * 16 + lane] = v;
      }
    }
  }
#pragma unroll 1
  for (int ct = wave + 2 * NW; ct < ntile; ct += NW) {
    const bf16_t* vr = vbase + (long long)ct * 16 * Sp;
    u32x4 br[kTiKs];
    ti_request_rows(vr, Sp, br);
    const float c = ti_weighted_sum(vr, Sp, br, ah, al);
    if (lane < 16) {
      const bf16_t v = f2bf(c);
      ctx[m0 + ct * 16 + lane] = v;
      cat[m0 + ct * 16 + lane] = v;
    }
  }
}

// ---- score launch -----------------------------------------------------------------------------------------
// What a step kernel costs, once its loads are requested up front, is the instructions its waves ISSUE: one
// workgroup per CU = two waves per SIMD and nothing else to switch to, and a wave64 VALU instruction holds the SIMD
// for 4 cycles — the 2 600 instructions of the round-3 score kernel's unrolled unpack + FMA window are its 10.5 us
// (a 10 KB unrolled block measured 18 us by itself, OS2S_TI_DEBUG; a cold instruction cache was ruled out with
// tools/probe_icache.hip). So the inference kernels are written for few issued instructions: packed bf16 dot
// products (v_dot2c_f32_bf16, one instruction per pair; ti_dot2 / ti_dot8 above) instead of unpack + FMA, rolled
// loops, and the location term on the matrix cores instead of a 32 x 32 unrolled FMA window.
__host__ __device__ inline size_t ti_scores_lds_floats(int S) {
  const size_t Sp = ((size_t)S + 15) & ~(size_t)15;
  return 64 + (Sp + 48) + Sp * kLocUnits / 2 + 64;     // q/bias, padded cumulative alignments, key columns (bf16)
}

// W_out[:, :H] h1 of the step -> mh [B, n_mel] (the half of the frame that only needs the cell output): a wave
// per output row round, the row pieces and the cell output straight from global memory in packed bf16
__device__ __forceinline__ void ti_frame_hpart(const AdAttn& p, const TiTail& q) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, nm = q.n_mel;
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  constexpr int kRows = 16;            // rows wave, wave + 8, ...: n_mel <= 128; H <= 1024: 2 pieces per lane
  const int k0 = min(lane * 8, H - 8), k1 = min(lane * 8 + 512, H - 8);
  const u32x4 h0 = *reinterpret_cast<const u32x4*>(yq + k0);
  u32x4 h1v = *reinterpret_cast<const u32x4*>(yq + k1);
  u32x4 w0[kRows], w1[kRows];
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    const bf16_t* wr = q.wout_h + (long long)min(wave + i * kAttnWaves, nm - 1) * H;
    w0[i] = *reinterpret_cast<const u32x4*>(wr + k0);
    w1[i] = *reinterpret_cast<const u32x4*>(wr + k1);
  }
  __builtin_amdgcn_sched_barrier(0);
  const u32x4 zero = {0u, 0u, 0u, 0u};
  const u32x4 h0m = lane * 8 < H ? h0 : zero;
  h1v = lane * 8 + 512 < H ? h1v : zero;
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    float sacc = ti_dot8(w0[i], h0m, 0.f);
    sacc = ti_dot8(w1[i], h1v, sacc);
    sacc = wave_sum_dpp(sacc);
    const int m = wave + i * kAttnWaves;
    if (lane == 0 && m < nm) q.mh[(long long)b * nm + m] = sacc;
  }
}

// location-sensitive scores of one unit part (32 units) of one sample:
//   e_part[s] = sum_u v[u] tanh(keys[s,u] + q[u] + bias[u] + sum_k cum[s + k - padl] Wck[k,u])
// The location term is a [S x 32 taps] x [32 taps x 32 units] product on the matrix cores: A = the Toeplitz rows
// of the (zero padded) cumulative alignments, B = the folded location filter, both split into bf16 hi + lo
// (three MFMAs per tile: the cumulative alignments grow with the step count and a single bf16 would lose them).
__device__ __forceinline__ void ti_scores_part(const AdAttn& p, const AdLoc& x, float* lds) {
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, U = p.U, S = p.S, K = p.loc_k;
  const int Sp = (S + 15) & ~15;
  float* qb = lds;                                   // [32] q + bias + folded location bias
  float* nv = qb + kLocUnits;                        // [32]
  float* cum = nv + kLocUnits;                       // [Sp + 48] cum[i] = cumulative[i - padl], zero padded
  uint16_t* keys = reinterpret_cast<uint16_t*>(cum + Sp + 48);   // [Sp][32] bf16
  const int slen = min(max(p.src_len[b], 0), S);
  const int u0 = part * kLocUnits;
  const long long row = (long long)b * p.T + p.t;
  // ---- requests: this wave's rows of Wq + the cell output (packed bf16, no staging) ---------------------------
  constexpr int UB = kLocUnits / kAttnWaves;         // 4 units per wave
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  const int k0 = min(lane * 8, H - 8), k1 = min(lane * 8 + 512, H - 8);
  const u32x4 h0 = *reinterpret_cast<const u32x4*>(yq + k0);
  u32x4 h1v = *reinterpret_cast<const u32x4*>(yq + k1);
  u32x4 wq0[UB], wq1[UB];
#pragma unroll
  for (int i = 0; i < UB; ++i) {
    const bf16_t* wr = p.wq + (long long)(u0 + wave * UB + i) * H;
    wq0[i] = *reinterpret_cast<const u32x4*>(wr + k0);
    wq1[i] = *reinterpret_cast<const u32x4*>(wr + k1);
  }
  // the folded location filter as the MFMA B operand: column n = lane & 15 (+ 16), taps (lane >> 4) * 8 ... + 8
  const int r16 = lane & 15, kb = lane >> 4;
  float wf[2][8];
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 8 + j;
      wf[ut][j] = p.wck[(long long)min(k, K - 1) * U + u0 + ut * 16 + r16];
    }
  // staging: padded cumulative alignments, this part's key columns
  {
    const int padl = (K - 1) / 2;
    const float* cs = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
    for (int i = tid; i < Sp + 48; i += kAttnThreads) {
      const int sp = i - padl;
      cum[i] = (sp >= 0 && sp < S) ? cs[sp] : 0.f;
    }
    const bf16_t* kp = p.keys + (long long)b * S * U + u0;
    for (int i = tid; i < slen * 4; i += kAttnThreads) {
      const int sp = i >> 2, c = i & 3;
      *reinterpret_cast<u32x4*>(keys + sp * kLocUnits + c * 8) = *reinterpret_cast<const u32x4*>(kp + (long long)sp * U + c * 8);
    }
  }
  float e_b = 0.f, e_v = 0.f;
  if (tid < kLocUnits) {
    const int u = u0 + tid;
    e_v = p.v[u];
    e_b = ((p.use_bias && p.bias) ? p.bias[u] : 0.f) + p.wck[(long long)K * U + u];
  }
  // ---- q of this wave's four units ------------------------------------------------------------------------
  {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 h0m = lane * 8 < H ? h0 : zero;
    h1v = lane * 8 + 512 < H ? h1v : zero;
    float qv[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      float sacc = ti_dot8(wq0[i], h0m, 0.f);
      sacc = ti_dot8(wq1[i], h1v, sacc);
      qv[i] = wave_sum_dpp(sacc);
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        qb[wave * UB + i] = qv[i];
        p.q_seq[row * U + u0 + wave * UB + i] = qv[i];
      }
    }
  }
  __syncthreads();
  if (tid < kLocUnits) { qb[tid] += e_b; nv[tid] = e_v; }
  // B operand registers: hi / lo halves of the filter taps (taps >= K are zero)
  bf16x8 bh[2], bl[2];
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = kb * 8 + j < K ? wf[ut][j] : 0.f;
      const __bf16 hi = (__bf16)w;
      bh[ut][j] = hi;
      bl[ut][j] = (__bf16)(w - (float)hi);
    }
  __syncthreads();
  const float q0 = qb[r16], q1 = qb[16 + r16], v0 = nv[r16], v1 = nv[16 + r16];
  float* eo = x.e_part + ((long long)b * kLocParts + part) * S;
  const int ntiles = (slen + 15) >> 4;
#pragma unroll 1
  for (int pt = wave; pt < ntiles; pt += kAttnWaves) {
    // A operand: row = position pt * 16 + (lane & 15), taps kb * 8 ... + 8 -> cum[pos + tap]
    const float* cp = cum + pt * 16 + r16 + kb * 8;
    bf16x8 ah, al;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float c = cp[j];
      const __bf16 hi = (__bf16)c;
      ah[j] = hi;
      al[j] = (__bf16)(c - (float)hi);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[1], acc1, 0, 0, 0);
    // accumulator element i: position pt * 16 + 4 * (lane >> 4) + i, unit (lane & 15) (+ 16)
    float pr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sp = pt * 16 + 4 * kb + i;
      const uint16_t* kr = keys + sp * kLocUnits + r16;
      const float x0 = acc0[i] + q0 + (sp < slen ? bf2f(kr[0]) : 0.f);
      const float x1 = acc1[i] + q1 + (sp < slen ? bf2f(kr[16]) : 0.f);
      pr[i] = row16_sum(v0 * tanh_fast(x0) + v1 * tanh_fast(x1));
    }
    if (r16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sp = pt * 16 + 4 * kb + i;
        if (sp < slen) eo[sp] = pr[i];
      }
    }
  }
}

// grid (kLocParts + 1, B): the location-sensitive scores + the cell-output half of the frame
__global__ __launch_bounds__(kAttnThreads) void ti_scores_kernel(AdAttn p, AdLoc x, TiTail q) {
  extern __shared__ float lds_raw[];
  if (q.state[1] != 0) return;
  if (blockIdx.x == kLocParts) { if (!(q.dbg & 1)) ti_frame_hpart(p, q); }
  else if (!(q.dbg & 2)) ti_scores_part(p, x, lds_raw);
}

}  // namespace os2s

using namespace os2s;

extern "C" size_t os2s_tacotron_infer_state_ints(int B) { return (size_t)4 + 2 * (size_t)B; }

static int ti_check(const os2s_tacotron_infer_t* x) {
  OS2S_REQUIRE(x && x->loop);
  const os2s_attn_decoder_t* d = x->loop;
  const int rc = ad_check(d);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(x->P >= 8 && x->n_mel >= 8 && x->x_seq && x->mel && x->stop && x->state && x->pv_t && x->values_t && x->wout_h);
  OS2S_REQUIRE(x->wp1 && x->bp1 && x->wp2 && x->bp2 && x->bout && x->wstop && x->bstop && x->bias0);
  OS2S_REQUIRE((x->w0x != nullptr) != (x->w0x8 != nullptr));
  if (x->w0x8) OS2S_REQUIRE(x->w0x8_scale && (d->L == 1 || (d->wcat8[1] && d->wcat8_scale[1])));
  OS2S_REQUIRE(x->prenet_keep > 0.f && x->prenet_keep <= 1.f);
  // what the step kernels are built for (anything else: the caller drives os2s_attn_decoder_fwd step by step)
  if (d->score_mode != 2 || !loc_split(d) || d->B > 32 || d->H % 64 || d->M % 64 || x->P % 64 || x->n_mel % 8 ||
      x->n_mel > 128 || x->P > 1024 || d->attn_in_keep < 1.f || d->out_keep < 1.f || d->tgt_len)
    return OS2S_ERR_UNSUPPORTED;
  if (x->P > 256 || x->n_mel / 8 > kTiW1Pieces * (kTiCtxThreads / x->P) || x->n_mel > 16 * kAttnWaves ||
      d->H > 1024 || !x->mh || d->loc_k > 32 || ti_scores_lds_floats(d->S) * sizeof(float) > 64 * 1024)
    return OS2S_ERR_UNSUPPORTED;
  if (ti_ctx_lds_floats(d->S, x->P, x->n_mel) * sizeof(float) > 64 * 1024 || d->M % 16) return OS2S_ERR_UNSUPPORTED;
  return OS2S_OK;
}

extern "C" int os2s_tacotron_infer_supported(const os2s_tacotron_infer_t* x) { return ti_check(x) == OS2S_OK; }

// units per workgroup: 4 (16 gate rows: H / 4 workgroups) or 8 (32 rows: half the workgroups, half the reads of the
// state rows); OS2S_TI_ROWS = 16 | 32 and OS2S_TI_WAVES = 8 | 16 override the defaults (experiments)
static int ti_rows() {
  static const int v = [] { const char* e = getenv("OS2S_TI_ROWS"); return e ? atoi(e) : 16; }();
  return v == 32 ? 32 : 16;
}
// waves x chunk slots per wave of the 16-row kernel: by default the smallest number of slots that covers K with 8
// or 9 waves (every slot of requests costs about a microsecond: 8 x 5 -> 8 x 4 measured 9.7 -> 8.7 us on the
// K = 2048 layer; K = 2304 = 36 chunks takes 9 waves x 4). OS2S_TI_GEOM = "<waves>x<slots>" overrides (experiments).
static void ti_geom(int K, int* nw, int* cpw) {
  static const int env = [] {
    const char* e = getenv("OS2S_TI_GEOM");
    int a = 0, b = 0;
    return (e && sscanf(e, "%dx%d", &a, &b) == 2) ? a * 100 + b : 0;
  }();
  if (env) { *nw = env / 100; *cpw = env % 100; return; }
  const int nchunks = K >> 6;
  if (nchunks <= 32) { *nw = 8; *cpw = 4; }
  else if (nchunks <= 36) { *nw = 9; *cpw = 4; }
  else { *nw = 8; *cpw = 5; }
}

template <bool FP8, int NT>
static int ti_launch_lstm_nt(hipStream_t stream, const TiLstm& c) {
  if (ti_rows() == 32 && c.H % 8 == 0) {
    OS2S_LAUNCH((ti_lstm_kernel<FP8, 2, NT, 8, 5>), dim3(ceil_div(c.H, 8)), dim3(512), 0, stream, c);
    return OS2S_OK;
  }
  const dim3 grid(ceil_div(c.H, 4));
  int nw, cpw;
  ti_geom(c.K, &nw, &cpw);
  switch (nw * 100 + cpw) {
    case 804: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 8, 4>), grid, dim3(512), 0, stream, c); break;
    case 904: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 9, 4>), grid, dim3(576), 0, stream, c); break;
    case 1203: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 12, 3>), grid, dim3(768), 0, stream, c); break;
    case 1103: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 11, 3>), grid, dim3(704), 0, stream, c); break;
    case 1603: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 16, 3>), grid, dim3(1024), 0, stream, c); break;
    case 1602: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 16, 2>), grid, dim3(1024), 0, stream, c); break;
    default: OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, NT, 8, 5>), grid, dim3(512), 0, stream, c); break;
  }
  return OS2S_OK;
}

template <bool FP8>
static int ti_launch_lstm(hipStream_t stream, const TiLstm& c) {
  return c.B <= 16 ? ti_launch_lstm_nt<FP8, 1>(stream, c) : ti_launch_lstm_nt<FP8, 2>(stream, c);
}

// the training pass's cell launch (os2s_attn_decoder_fwd) on the same kernel: returns false when the shape is not
// covered (the caller keeps ad_cell_fwd_kernel)
static bool ti_cell_supported(int B, int H, int Kc) { return B <= 32 && H % 4 == 0 && Kc % 64 == 0; }
static int ti_launch_cell(hipStream_t stream, const TiLstm& c, bool fp8) {
  return fp8 ? ti_launch_lstm<true>(stream, c) : ti_launch_lstm<false>(stream, c);
}

extern "C" int os2s_tacotron_infer_steps(os2s_stream_t stream_, const os2s_tacotron_infer_t* x, int t_begin,
                                         int t_end) {
  const int rc = ti_check(x);
  if (rc != OS2S_OK) return rc;
  const os2s_attn_decoder_t* d = x->loop;
  OS2S_REQUIRE(t_begin >= 0 && t_begin <= t_end && t_end <= d->T);
  hipStream_t stream = (hipStream_t)stream_;
  const int B = d->B, T = d->T, H = d->H, M = d->M, L = d->L, P = x->P;
  AdAttn at;
  ad_fill_attn(d, at);
  AdLoc lx;
  lx.e_part = d->loc_ws + (size_t)(d->loc_k + 1) * d->U;
  lx.dal = nullptr; lx.dcum_part = nullptr;
  // context columns per part: 16-column MFMA tiles, up to two per wave with their operands requested up front.
  // (parts + 1) * B workgroups should not exceed the CUs: two workgroups on one CU measured 12 us for the launch
  // against 7.6 / 9.1 for the context / frame parts alone. OS2S_TI_CTX_PARTS overrides (experiments).
  static const int parts_env = [] { const char* e = getenv("OS2S_TI_CTX_PARTS"); return e ? atoi(e) : 0; }();
  int ctx_parts = parts_env > 0 ? parts_env : (B > 28 ? 4 : kLocCtxParts);
  while (ctx_parts > 1 && M % (16 * ctx_parts)) ctx_parts >>= 1;
  const int MQ = M / ctx_parts;
  const size_t lds_s = ti_scores_lds_floats(d->S) * sizeof(float);
  const size_t lds_c = ti_ctx_lds_floats(d->S, P, x->n_mel) * sizeof(float);
  TiTail q;
  q.P = P; q.n_mel = x->n_mel; q.mask_seq = x->mask_decoder_sequence; q.first = 0;
  { static const int dbg = [] { const char* e = getenv("OS2S_TI_DEBUG"); return e ? atoi(e) : 0; }(); q.dbg = dbg; }
  q.keep = x->prenet_keep; q.seed[0] = x->prenet_seed[0]; q.seed[1] = x->prenet_seed[1];
  q.wp1 = (const bf16_t*)x->wp1; q.bp1 = x->bp1; q.wp2 = (const bf16_t*)x->wp2; q.bp2 = x->bp2;
  q.wout_h = (const bf16_t*)x->wout_h; q.pv_t = (const bf16_t*)x->pv_t; q.values_t = (const bf16_t*)x->values_t;
  q.bout = x->bout;
  q.wstop = (const bf16_t*)x->wstop; q.bstop = x->bstop; q.mh = x->mh;
  q.x_seq = (bf16_t*)x->x_seq; q.mel = (bf16_t*)x->mel; q.stop = x->stop; q.state = x->state;
  if (t_begin == 0) {
    const int n = (d->loc_k + 1) * d->U;
    OS2S_LAUNCH(ad_fold_location_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, d->conv_w, d->conv_b,
                d->dense_w, d->loc_k, d->loc_f, d->U, d->loc_ws);
    TiTail q0 = q;
    q0.first = 1;
    at.t = 0;
    OS2S_LAUNCH(ti_context_kernel, dim3(ctx_parts + 1, B), dim3(kTiCtxThreads), lds_c, stream, at, lx, q0, ctx_parts, MQ);
  }
  const bool fp8 = x->w0x8 != nullptr;
  for (int t = t_begin; t < t_end; ++t) {
    for (int l = 0; l < L; ++l) {
      TiLstm c;
      c.B = B; c.H = H; c.forget_bias = d->forget_bias; c.state = x->state;
      c.gx = nullptr; c.ldgx = 0; c.gates = nullptr; c.ldgates = 0; c.out_keep = 1.f; c.out_seed = 0; c.drop_t = 0; c.drop_T = 0;
      if (l == 0) {
        c.K = P + M + H; c.Ka = P;
        c.in_a = (const bf16_t*)x->x_seq + (long long)t * P; c.lda = (long long)(T + 1) * P;
        c.in_b = (const bf16_t*)d->cat[0] + (long long)t * (M + H); c.ldb = (long long)(T + 1) * (M + H);
        c.w = fp8 ? (const void*)x->w0x8 : (const void*)x->w0x; c.scale = x->w0x8_scale; c.bias = x->bias0;
      } else {
        c.K = 2 * H; c.Ka = 0;
        c.in_a = nullptr; c.lda = 0;
        c.in_b = (const bf16_t*)d->cat[1] + (long long)t * 2 * H; c.ldb = (long long)(T + 1) * 2 * H;
        c.w = fp8 ? (const void*)d->wcat8[1] : (const void*)d->wcat[1]; c.scale = d->wcat8_scale[1];
        c.bias = d->bias[1];
      }
      c.c_prev = t > 0 ? d->c_seq[l] + (long long)(t - 1) * H : nullptr; c.ldc_prev = (long long)T * H;
      c.c_out = d->c_seq[l] + (long long)t * H; c.ldc_out = (long long)T * H;
      // recurrent slot of the next step's input row
      const int Kc = l == 0 ? M + H : 2 * H;
      c.h1 = (bf16_t*)d->cat[l] + (long long)(t + 1) * Kc + (l == 0 ? M : H); c.ldh1 = (long long)(T + 1) * Kc;
      if (l == L - 1) { c.h2 = (bf16_t*)d->y_top + (long long)t * d->y_top_ts; c.ldh2 = d->y_top_bs; }
      else { c.h2 = (bf16_t*)d->cat[l + 1] + (long long)t * 2 * H; c.ldh2 = (long long)(T + 1) * 2 * H; }
      const int r2 = fp8 ? ti_launch_lstm<true>(stream, c) : ti_launch_lstm<false>(stream, c);
      if (r2 != OS2S_OK) return r2;
    }
    at.t = t;
    OS2S_LAUNCH(ti_scores_kernel, dim3(kLocParts + 1, B), dim3(kAttnThreads), lds_s, stream, at, lx, q);
    OS2S_LAUNCH(ti_context_kernel, dim3(ctx_parts + 1, B), dim3(kTiCtxThreads), lds_c, stream, at, lx, q, ctx_parts, MQ);
  }
  return OS2S_OK;
}

// ---- the training pass on the small-code kernels -------------------------------------------------------------
// os2s_attn_decoder_fwd (location-sensitive mode, no per-sample target lengths, B <= 32) launches ti_lstm_kernel
// for the cells (+ input projection of the step, saved gates, output dropout) and ti_scores_part for the scores:
// 13.3 -> ~8.7 us per cell launch and 10.4 -> ~6.5 us per score launch of a Tacotron2 decoder step. OS2S_AD_FAST=0
// keeps the round-3 kernels.
namespace os2s {
__global__ __launch_bounds__(kAttnThreads) void ad_loc_scores_mfma_kernel(AdAttn p, AdLoc x) {
  extern __shared__ float lds_raw[];
  ti_scores_part(p, x, lds_raw);
}
}  // namespace os2s

// ---- score gradient of the location-sensitive attention on the matrix cores ------------------------------------
// ad_loc_score_bwd_kernel's arithmetic (softmax backward, score gradient of one 32-unit part of one sample) with its
// 36 KB of unrolled register-window code replaced by four small MFMA products:
//   x[s,u]     = keys + q + bias + sum_k cum[s+k-p] Wck[k,u]            (as the forward: A = Toeplitz rows of cum)
//   d0[s,u]    = de[s] v[u] (1 - tanh^2 x)                               -> dpre_seq (bf16), dq = sum_s d0, dv += sum_s de tanh x
//   dWck[k,u] += sum_s cum[s+k-p] d0[s,u]        A = Toeplitz COLUMNS of cum (taps x positions), B = d0^T[u][s]
//   G[s,k]     = sum_u d0[s,u] Wck[k,u]          A = d0[s][u], B = Wck;   dcum[c] = sum_k G[c + p - k][k]
// every operand bf16 hi + lo (three MFMAs per product). Waves 0-3 own the four 16x16 tiles of dWck over all
// positions, waves 4-7 the position tiles of G: no cross-wave sums, fixed summation order (deterministic).
namespace os2s {
constexpr int kSbD0sPitch = 40;                      // bf16 elements per position row of d0[s][u]
__host__ __device__ inline size_t ad_score_bwd_mfma_lds_floats(int S) {
  const size_t Sp = ((size_t)S + 31) & ~(size_t)31;
  return 64 + (Sp + 48) + 3 * Sp + Sp * kLocUnits / 2 + Sp * kSbD0sPitch + (size_t)kLocUnits * (Sp + 8) +
         Sp * 33 + 2 * (size_t)kAttnWaves * 4 * kLocUnits + 64;
}

__global__ __launch_bounds__(kAttnThreads) void ad_loc_score_bwd_mfma_kernel(AdAttn p, AdLoc x) {
  extern __shared__ float lds_raw[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int U = p.U, S = p.S, K = p.loc_k;
  const int Sp = (S + 31) & ~31, TP = Sp + 8;
  float* qb = lds_raw;                               // [32]
  float* nv = qb + kLocUnits;                        // [32]
  float* cum = nv + kLocUnits;                       // [Sp + 48] zero padded, cum[i] = cumulative[i - padl]
  float* de = cum + Sp + 48;                         // [Sp] softmax-backward of the alignments, zero past the length
  float* ea = de + Sp;                               // [Sp] alignments   (scratch of the preamble)
  float* da = ea + Sp;                               // [Sp] d(alignment)
  uint16_t* keys = reinterpret_cast<uint16_t*>(da + Sp);                    // [Sp][32] bf16
  uint16_t* d0s_hi = keys + (size_t)Sp * kLocUnits;                         // [Sp][40] bf16: d0[s][u]
  uint16_t* d0s_lo = d0s_hi + (size_t)Sp * kSbD0sPitch;
  uint16_t* d0t_hi = d0s_lo + (size_t)Sp * kSbD0sPitch;                     // [32][Sp + 8] bf16: d0^T[u][s]
  uint16_t* d0t_lo = d0t_hi + (size_t)kLocUnits * TP;
  float* G = reinterpret_cast<float*>(d0t_lo + (size_t)kLocUnits * TP);     // [Sp][33]
  float* pq = G + (size_t)Sp * 33;                   // [waves][4][32] partial dq
  float* pn = pq + kAttnWaves * 4 * kLocUnits;       // [waves][4][32] partial dv terms
  float* red = pn + kAttnWaves * 4 * kLocUnits;      // [64]
  const int slen = min(max(p.src_len[b], 0), S);
  const int u0 = part * kLocUnits;
  const long long row = (long long)b * p.T + p.t;
  const int padl = (K - 1) / 2;
  const int r16 = lane & 15, kb = lane >> 4;
  // the folded location filter: forward B operand (taps x units) ...
  float wf[2][8];
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[ut][j] = p.wck[(long long)min(kb * 8 + j, K - 1) * U + u0 + ut * 16 + r16];
  // ... and as the B operand of G (units x taps): tap = nt * 16 + r16, units kb * 8 ... + 8 (contiguous)
  float wg[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float* wr = p.wck + (long long)min(nt * 16 + r16, K - 1) * U + u0 + kb * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(wr), c = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wg[nt][j] = a[j]; wg[nt][4 + j] = c[j]; }
  }
  {
    const float* cs = p.cum_seq + ((long long)b * (p.T + 1) + p.t) * S;
    for (int i = tid; i < Sp + 48; i += kAttnThreads) {
      const int sp = i - padl;
      cum[i] = (sp >= 0 && sp < S) ? cs[sp] : 0.f;
    }
    const bf16_t* kp = p.keys + (long long)b * S * U + u0;
    for (int i = tid; i < slen * 4; i += kAttnThreads) {
      const int sp = i >> 2, c = i & 3;
      *reinterpret_cast<u32x4*>(keys + sp * kLocUnits + c * 8) = *reinterpret_cast<const u32x4*>(kp + (long long)sp * U + c * 8);
    }
    for (int sp = tid; sp < Sp; sp += kAttnThreads) {
      ea[sp] = sp < slen ? p.align_seq[row * S + sp] : 0.f;
      da[sp] = sp < slen ? x.dal[(long long)b * S + sp] : 0.f;
    }
  }
  if (tid < kLocUnits) {
    const int u = u0 + tid;
    qb[tid] = p.q_seq[row * U + u] + ((p.use_bias && p.bias) ? p.bias[u] : 0.f) + p.wck[(long long)K * U + u];
    nv[tid] = p.v[u];
  }
  __syncthreads();
  // softmax backward: de[s] = a[s] (dal[s] - sum_s' a[s'] dal[s'])
  float dot = 0.f;
  for (int sp = tid; sp < slen; sp += kAttnThreads) dot += ea[sp] * da[sp];
  dot = block_sum(dot, red);
  for (int sp = tid; sp < Sp; sp += kAttnThreads) de[sp] = sp < slen ? ea[sp] * (da[sp] - dot) : 0.f;
  bf16x8 bh[2], bl[2];
#pragma unroll
  for (int ut = 0; ut < 2; ++ut)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = kb * 8 + j < K ? wf[ut][j] : 0.f;
      const __bf16 hi = (__bf16)w;
      bh[ut][j] = hi;
      bl[ut][j] = (__bf16)(w - (float)hi);
    }
  __syncthreads();
  // ---- phase A: score gradient of every 16-position tile (all tiles up to Sp: dead ones write zeros) ----------
  const float q0 = qb[r16], q1 = qb[16 + r16], v0 = nv[r16], v1 = nv[16 + r16];
  float dq0 = 0.f, dq1 = 0.f, dn0 = 0.f, dn1 = 0.f;
  bf16_t* dps = p.dpre_seq + (row * S) * U + u0;
#pragma unroll 1
  for (int pt = wave; pt < Sp / 16; pt += kAttnWaves) {
    const float* cp = cum + pt * 16 + r16 + kb * 8;
    bf16x8 ah, al;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float c = cp[j];
      const __bf16 hi = (__bf16)c;
      ah[j] = hi;
      al[j] = (__bf16)(c - (float)hi);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[1], acc1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sp = pt * 16 + 4 * kb + i;
      const bool on = sp < slen;
      const uint16_t* kr = keys + sp * kLocUnits + r16;
      const float t0 = tanh_fast(acc0[i] + q0 + (on ? bf2f(kr[0]) : 0.f));
      const float t1 = tanh_fast(acc1[i] + q1 + (on ? bf2f(kr[16]) : 0.f));
      const float des = de[sp];
      const float d0 = des * v0 * (1.f - t0 * t0), d1 = des * v1 * (1.f - t1 * t1);
      dq0 += d0; dq1 += d1;
      dn0 += des * t0; dn1 += des * t1;
      const bf16_t h0 = f2bf(d0), h1 = f2bf(d1);
      const bf16_t l0 = f2bf(d0 - bf2f(h0)), l1 = f2bf(d1 - bf2f(h1));
      if (on) {
        dps[(long long)sp * U + r16] = h0;
        dps[(long long)sp * U + 16 + r16] = h1;
      }
      d0s_hi[sp * kSbD0sPitch + r16] = h0; d0s_hi[sp * kSbD0sPitch + 16 + r16] = h1;
      d0s_lo[sp * kSbD0sPitch + r16] = l0; d0s_lo[sp * kSbD0sPitch + 16 + r16] = l1;
      d0t_hi[r16 * TP + sp] = h0; d0t_hi[(16 + r16) * TP + sp] = h1;
      d0t_lo[r16 * TP + sp] = l0; d0t_lo[(16 + r16) * TP + sp] = l1;
    }
  }
  pq[(wave * 4 + kb) * kLocUnits + r16] = dq0; pq[(wave * 4 + kb) * kLocUnits + 16 + r16] = dq1;
  pn[(wave * 4 + kb) * kLocUnits + r16] = dn0; pn[(wave * 4 + kb) * kLocUnits + 16 + r16] = dn1;
  __syncthreads();
  if (wave < 4) {
    // ---- phase B: dWck tile (taps mt * 16 ..., units nt * 16 ...) over all positions ---------------------------
    const int mt = wave >> 1, nt = wave & 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < Sp / 32; ++ks) {
      const float* cp = cum + ks * 32 + kb * 8 + mt * 16 + r16;      // row = tap, 8 consecutive positions
      bf16x8 ah, al;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = cp[j];
        const __bf16 hi = (__bf16)c;
        ah[j] = hi;
        al[j] = (__bf16)(c - (float)hi);
      }
      const bf16x8 th = *reinterpret_cast<const bf16x8*>(d0t_hi + (nt * 16 + r16) * TP + ks * 32 + kb * 8);
      const bf16x8 tl = *reinterpret_cast<const bf16x8*>(d0t_lo + (nt * 16 + r16) * TP + ks * 32 + kb * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, th, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, th, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, tl, acc, 0, 0, 0);
    }
    float* dwa = p.dwck_acc + (long long)b * K * U + u0 + nt * 16 + r16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = mt * 16 + 4 * kb + i;
      if (k < K) dwa[(long long)k * U] += acc[i];
    }
  } else {
    // ---- phase C: G[s, k] = sum_u d0[s, u] Wck[k, u] for the position tiles of this wave ---------------------------
    bf16x8 gh[2], gl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = nt * 16 + r16 < K ? wg[nt][j] : 0.f;
        const __bf16 hi = (__bf16)w;
        gh[nt][j] = hi;
        gl[nt][j] = (__bf16)(w - (float)hi);
      }
#pragma unroll 1
    for (int pt = wave - 4; pt < Sp / 16; pt += 4) {
      const bf16x8 sh = *reinterpret_cast<const bf16x8*>(d0s_hi + (pt * 16 + r16) * kSbD0sPitch + kb * 8);
      const bf16x8 sl = *reinterpret_cast<const bf16x8*>(d0s_lo + (pt * 16 + r16) * kSbD0sPitch + kb * 8);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, gh[nt], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sl, gh[nt], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, gl[nt], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) G[(pt * 16 + 4 * kb + i) * 33 + nt * 16 + r16] = acc[i];
      }
    }
  }
  __syncthreads();
  // ---- phase D: state gradient along the anti-diagonals of G; dq / dv sums in a fixed order ------------------------
  float* dpo = x.dcum_part + ((long long)b * kLocParts + part) * S;
  for (int c = tid; c < S; c += kAttnThreads) {
    float a = 0.f;
    for (int k = 0; k < K; ++k) {
      const int sp = c + padl - k;
      if (sp >= 0 && sp < Sp) a += G[sp * 33 + k];
    }
    dpo[c] = a;
  }
  if (tid < kLocUnits) {
    float dq = 0.f, dn = 0.f;
    for (int w = 0; w < kAttnWaves * 4; ++w) { dq += pq[w * kLocUnits + tid]; dn += pn[w * kLocUnits + tid]; }
    p.dq_seq[row * U + u0 + tid] = f2bf(dq);
    p.dnv_acc[(long long)b * U + u0 + tid] += dn;
    p.dbd_acc[(long long)b * U + u0 + tid] += dq;
  }
}
}  // namespace os2s

static bool ad_fast_score_bwd(const os2s_attn_decoder_t* d);

static bool ad_fast_cells(const os2s_attn_decoder_t* d) {
  static const int on = [] { const char* e = getenv("OS2S_AD_FAST"); return e ? atoi(e) : 1; }();
  if (!on || d->tgt_len || d->score_mode != 2 || d->B > 32 || d->H > 1024 || d->H % 64 || d->M % 64 || d->loc_k > 32)
    return false;
  if (ti_scores_lds_floats(d->S) * sizeof(float) > 64 * 1024) return false;
  for (int l = 0; l < d->L; ++l)
    if (!ti_cell_supported(d->B, d->H, l == 0 ? d->M + d->H : 2 * d->H)) return false;
  return true;
}

static int ad_launch_fast_cell(hipStream_t stream, const os2s_attn_decoder_t* d, int l, int t) {
  const int B = d->B, T = d->T, H = d->H, M = d->M, L = d->L;
  const int Kc = l == 0 ? M + H : 2 * H;
  TiLstm c;
  c.B = B; c.H = H; c.K = Kc; c.Ka = 0; c.in_a = nullptr; c.lda = 0;
  c.in_b = (const bf16_t*)d->cat[l] + (long long)t * Kc; c.ldb = (long long)(T + 1) * Kc;
  const bool fp8 = d->wcat8[l] != nullptr;
  if (fp8 && !d->wcat8_scale[l]) return OS2S_ERR_INVALID_ARG;
  c.w = fp8 ? (const void*)d->wcat8[l] : (const void*)d->wcat[l]; c.scale = d->wcat8_scale[l]; c.bias = d->bias[l];
  c.forget_bias = d->forget_bias;
  c.c_prev = t > 0 ? d->c_seq[l] + (long long)(t - 1) * H : nullptr; c.ldc_prev = (long long)T * H;
  c.c_out = d->c_seq[l] + (long long)t * H; c.ldc_out = (long long)T * H;
  c.h1 = (bf16_t*)d->cat[l] + (long long)(t + 1) * Kc + (l == 0 ? M : H); c.ldh1 = (long long)(T + 1) * Kc;
  if (l == L - 1) { c.h2 = (bf16_t*)d->y_top + (long long)t * d->y_top_ts; c.ldh2 = d->y_top_bs; }
  else { c.h2 = (bf16_t*)d->cat[l + 1] + (long long)t * 2 * H; c.ldh2 = (long long)(T + 1) * 2 * H; }
  c.state = nullptr;
  c.gx = l == 0 ? (const bf16_t*)d->gx0 + (long long)t * 4 * H : nullptr; c.ldgx = (long long)T * 4 * H;
  c.gates = d->gates[l] ? (bf16_t*)d->gates[l] + (long long)t * 4 * H : nullptr; c.ldgates = (long long)T * 4 * H;
  c.out_keep = d->out_keep; c.out_seed = d->out_seed[l]; c.drop_t = t; c.drop_T = T;
  return ti_launch_cell(stream, c, fp8);
}

static int ad_launch_fast_scores(hipStream_t stream, const os2s::AdAttn& at, const os2s::AdLoc& lx,
                                 const os2s_attn_decoder_t* d) {
  const size_t lds = ti_scores_lds_floats(d->S) * sizeof(float);
  OS2S_LAUNCH(ad_loc_scores_mfma_kernel, dim3(kLocParts, d->B), dim3(kAttnThreads), lds, stream, at, lx);
  return OS2S_OK;
}

// the backward pass's score-gradient launch on the MFMA kernel (same conditions as the forward fast path)
static bool ad_fast_score_bwd(const os2s_attn_decoder_t* d) {
  return ad_fast_cells(d) && ad_score_bwd_mfma_lds_floats(d->S) * sizeof(float) <= 160 * 1024 && d->U % 4 == 0;
}
static int ad_launch_fast_score_bwd(hipStream_t stream, const os2s::AdAttn& at, const os2s::AdLoc& lx,
                                    const os2s_attn_decoder_t* d) {
  const size_t lds = ad_score_bwd_mfma_lds_floats(d->S) * sizeof(float);
  static size_t attr_for = 0;
  if (lds > 64 * 1024 && lds > attr_for) {
    if (hipFuncSetAttribute((const void*)ad_loc_score_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return OS2S_ERR_LAUNCH;
    attr_for = lds;
  }
  OS2S_LAUNCH(ad_loc_score_bwd_mfma_kernel, dim3(kLocParts, d->B), dim3(kAttnThreads), lds, stream, at, lx);
  return OS2S_OK;
}
