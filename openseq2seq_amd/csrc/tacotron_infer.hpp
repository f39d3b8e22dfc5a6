// Free-running Tacotron2 decoding (eval / infer), gfx950. Included at the end of attn_decoder.hip: it reuses
// that translation unit's location-sensitive score kernel.
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
//                              over 8 waves, EVERY load of a wave (weights as e4m3 or bf16, inputs straight from
//                              the row-major state rows — no LDS staging) in flight before its first MFMA
//   ti_lstm_kernel (layer 1)
//   ti_scores_kernel           query projection + partial scores (attn_decoder.hip), 4 unit parts x B, and as a fifth
//                              part per sample W_out[:, :H] h1 (the half of the frame that only needs the cell output)
//   ti_context_kernel          softmax, alignments, context columns (8 parts x B) and, as a ninth part per sample,
//                              the frame: W_out[:, :H] h1 + sum_s a[s] PV[s] + b with PV = values W_out[:, H:]^T
//                              computed ONCE per batch (the context half of the projection commutes with the
//                              attention sum), stop token, finished / length bookkeeping, and the pre-net of the
//                              NEXT step
// The stop decision stays on the device: the launch that sees the last sample finish writes the step count to
// state[1]; every later launch returns at once, so the host may enqueue steps ahead and poll every N steps —
// the result does not depend on N.
#pragma once

namespace os2s {

constexpr int kTiWaves = 8;          // waves per LSTM workgroup (reduction split)
constexpr int kTiCpw = 5;            // 64-wide k chunks per wave and round (K <= 2560 in one round)

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
  bf16_t* h2; long long ldh2;
  const int32_t* state;              // state[1] != 0: decoding has ended
};

// rows of a 16-row tile: r = 4 * unit + gate, so that after the MFMA (acc[i] = row 4 * (lane >> 4) + i,
// column lane & 15) a lane holds the four gates of ONE (unit, sample). MT row tiles (4 * MT units) x NT
// 16-sample column tiles per workgroup.
template <bool FP8, int MT, int NT>
__global__ __launch_bounds__(64 * kTiWaves) void ti_lstm_kernel(TiLstm p) {
  __shared__ float red[kTiWaves * MT * NT * 4 * 64];
  const int done = p.state[1];         // consumed after the loads are in flight
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
  float e_sc[4], e_bias[4], e_c;
  {
    const int cj = min(e_j, H - 1), cb = min(e_b, p.B - 1);
    const float* scp = FP8 ? p.scale : p.c_out;            // any readable fp32 array of >= 4H elements
    const float* bip = p.bias ? p.bias : scp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      e_sc[g] = scp[g * H + cj];
      e_bias[g] = bip[g * H + cj];
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
        if (FP8) wa[i][mt] = *reinterpret_cast<const u32x4*>(w8[mt] + k);
        else {
          wa[i][mt] = *reinterpret_cast<const u32x4*>(w16[mt] + k);
          wb[i][mt] = *reinterpret_cast<const u32x4*>(w16[mt] + k + 8);
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
    pre[g] = s * e_sc[g] + e_bias[g];
  }
  const float ig = sigmoidf_(pre[0]), gg = tanhf(pre[1]);
  const float fg = sigmoidf_(pre[2] + p.forget_bias), og = sigmoidf_(pre[3]);
  const float cn = e_c * fg + ig * gg;
  const bf16_t hn = f2bf(tanhf(cn) * og);
  p.c_out[(long long)e_b * p.ldc_out + e_j] = cn;
  if (p.h1) p.h1[(long long)e_b * p.ldh1 + e_j] = hn;
  if (p.h2) p.h2[(long long)e_b * p.ldh2 + e_j] = hn;
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
  const float* pv;                     // [B, S, n_mel]
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
constexpr int kTiPvLoads = 8;          // PV rows per thread and round

__host__ __device__ inline size_t ti_tail_lds_floats(int S, int P, int n_mel) {
  // e [S] + red [32] + fr [n_mel] + x1 [P] + partials: max(G * n_mel, P * (P / 8 + 1), P * 8)
  const size_t G = kTiCtxThreads / (n_mel / 4);
  size_t part = G * n_mel;
  if ((size_t)P * (P / 8 + 1) > part) part = (size_t)P * (P / 8 + 1);
  if ((size_t)P * 8 > part) part = (size_t)P * 8;
  return (size_t)S + 4 + 32 + n_mel + P + part + 64;
}

// The frame part of a sample: one workgroup, ONE dependent chain (alignments -> frame -> stop token -> pre-net
// layer 1 -> layer 2), so everything that does not depend on the chain is requested first: both pre-net
// matrices (168 KB) and this sample's rows of PV sit in registers before the softmax starts.
__device__ __forceinline__ void ti_tail(const AdAttn& p, const AdLoc& x, const TiTail& q, float* lds) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int S = p.S, P = q.P, nm = q.n_mel, B = p.B, T = p.T;
  float* e = lds;                      // [S]
  float* red = e + ((S + 3) & ~3);     // [32]
  float* fr = red + 32;                // [nm] the frame, bf16-rounded
  float* x1 = fr + nm;                 // [P]
  float* part = x1 + P;                // partial sums of the stage at hand
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
  // PV rows of this sample: thread = (quad of frame columns m4, position group sg); positions sg, sg + G, ...
  const int nm4 = nm >> 2, G = kTiCtxThreads / nm4;
  const int m4 = tid % nm4, sg = tid / nm4;
  f32x4 pvr[kTiPvLoads];
  const float* pvb = q.pv + (long long)b * S * nm + m4 * 4;
#pragma unroll
  for (int i = 0; i < kTiPvLoads; ++i) {
    const int sp = min(sg + i * G, S - 1);
    pvr[i] = *reinterpret_cast<const f32x4*>(pvb + (long long)sp * nm);
  }
  float mh = 0.f, bo = 0.f;
  if (!q.first && tid < nm) { mh = q.mh[(long long)b * nm + tid]; bo = q.bout[tid]; }
  __builtin_amdgcn_sched_barrier(0);
  if (q.first) {
    for (int k = tid; k < nm; k += kTiCtxThreads) fr[k] = 0.f;
    __syncthreads();
  } else {
    // ---- alignments ---------------------------------------------------------------------------------------
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
    if (lane == 0) red[tid >> 6] = mx;
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
    if (lane == 0) red[8 + (tid >> 6)] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < kTiCtxThreads / 64; ++w) sum += red[8 + w];
    const float inv = slen > 0 ? 1.f / sum : 0.f;
    // ---- context half of the frame: sum_s a[s] PV[s, :] ------------------------------------------------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (sg < G) {
#pragma unroll
      for (int i = 0; i < kTiPvLoads; ++i) {
        const int sp = sg + i * G;
        const float a = sp < slen ? e[sp] * inv : 0.f;
        acc += a * pvr[i];
      }
      for (int sp = sg + kTiPvLoads * G; sp < slen; sp += G) {       // S > 8 G: the rest, not prefetched
        const f32x4 v = *reinterpret_cast<const f32x4*>(pvb + (long long)sp * nm);
        acc += (e[sp] * inv) * v;
      }
      *reinterpret_cast<f32x4*>(part + sg * nm + m4 * 4) = acc;
    }
    __syncthreads();
    if (tid < nm) {
      float sacc = mh + bo;
      for (int g2 = 0; g2 < G; ++g2) sacc += part[g2 * nm + tid];
      const bf16_t fb = f2bf(sacc);
      q.mel[((long long)b * T + p.t) * nm + tid] = fb;
      fr[tid] = bf2f(fb);
    }
    __syncthreads();
    // ---- stop token, finished / length bookkeeping (wave 0; the others go on) ----------------------------------
    if (tid < 64) {
      float sacc = 0.f;
      for (int k = tid; k < nm; k += 64) sacc += bf2f(q.wstop[k]) * fr[k];
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
      const int pc = s1 + i * TPR;
      if (pc < pcs1) {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
          s += bflo(w1[i][k2]) * fr[pc * 8 + 2 * k2] + bfhi(w1[i][k2]) * fr[pc * 8 + 2 * k2 + 1];
      }
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
    x1[tid] = bf2f(f2bf(s));            // the layer's output is a bf16 tensor in the teacher-forced pass too
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kTiW2Rows; ++i) {
    const int j = rg + i * RG;
    if (j < P && rg < RG) {
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2)
        s += bflo(w2[i][k2]) * x1[pc2 * 8 + 2 * k2] + bfhi(w2[i][k2]) * x1[pc2 * 8 + 2 * k2 + 1];
      part[j * (pcs2 + 1) + pc2] = s;
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

// grid (ctx_parts + 1, B), 512 threads: parts < ctx_parts are the context columns (ad_loc_context_kernel's
// arithmetic with twice the position slices); the last part is the frame
__global__ __launch_bounds__(kTiCtxThreads) void ti_context_kernel(AdAttn p, AdLoc x, TiTail q, int ctx_parts,
                                                                   int ncg, int nsp) {
  extern __shared__ float lds_raw[];
  const int cpart = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int M = p.M, S = p.S;
  if (cpart == ctx_parts) {
    if (!q.first && (q.state[1] != 0 || (q.dbg & 4))) return;
    ti_tail(p, x, q, lds_raw);
    return;
  }
  if (q.first || q.state[1] != 0 || (q.dbg & 8)) return;
  float* e = lds_raw;                  // [S]
  float* red = e + S;                  // [32]
  float* part = red + 32;              // [nsp][ncg * 8]
  const int slen = min(max(p.src_len[b], 0), S);
  const long long row = (long long)b * p.T + p.t;
  // this thread's rows of `values` do not depend on the alignments: requested before the softmax
  const int MQ = ncg * 8, m0 = cpart * MQ;
  const int cg = tid % ncg, sq = tid / ncg;
  constexpr int kVr = 8;
  u32x4 vr[kVr];
  const bf16_t* vp = p.values + (long long)b * S * M + m0 + cg * 8;
#pragma unroll
  for (int i = 0; i < kVr; ++i) vr[i] = *reinterpret_cast<const u32x4*>(vp + (long long)min(sq + i * nsp, S - 1) * M);
  __builtin_amdgcn_sched_barrier(0);
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
  for (int sp = tid; sp < S; sp += kTiCtxThreads) {
    const float a = sp < slen ? e[sp] * inv : 0.f;
    e[sp] = a;
    if (cpart == 0) {
      p.align_seq[row * S + sp] = a;
      const long long ci = ((long long)b * (p.T + 1) + p.t) * S + sp;
      p.cum_seq[ci + S] = p.cum_seq[ci] + a;
    }
  }
  __syncthreads();
  if (sq < nsp) {
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kVr; ++i) {
      const int sp = sq + i * nsp;
      const float a = sp < slen ? e[sp] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { a8[2 * k] += a * bflo(vr[i][k]); a8[2 * k + 1] += a * bfhi(vr[i][k]); }
    }
    for (int sp = sq + kVr * nsp; sp < slen; sp += nsp) {
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
      for (int s2 = 0; s2 < nsp; ++s2) c8[k] += part[s2 * MQ + tid * 8 + k];
    }
    const int m8 = (m0 >> 3) + tid;
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(c8[2 * k], c8[2 * k + 1]);
    *reinterpret_cast<u32x4*>(p.ctx + (long long)b * p.ctx_bs + (long long)p.t * p.ctx_ts + m8 * 8) = o;
    *reinterpret_cast<u32x4*>(p.cat0 + ((long long)b * (p.T + 1) + p.t + 1) * p.Kc0 + m8 * 8) = o;
  }
}

// W_out[:, :H] h1 of the step -> mh [B, n_mel]: it only needs the cell output, so it rides in the score launch as a
// fifth part per sample (a wave per output row round, every load in flight at once)
__device__ __forceinline__ void ti_frame_hpart(const AdAttn& p, const TiTail& q) {
  extern __shared__ float lds_raw[];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, nm = q.n_mel;
  float* hq = lds_raw;
  const bf16_t* yq = p.yq + (long long)b * p.yq_bs + (long long)p.t * p.yq_ts;
  // rows wave, wave + 8, ...: kTiHRows rows per wave, H / 8 pieces per row over 64 lanes (H <= 1024: 2 per lane)
  constexpr int kRows = 16, kPc = 2;
  u32x4 wv[kRows][kPc];
#pragma unroll
  for (int i = 0; i < kRows; ++i)
#pragma unroll
    for (int c = 0; c < kPc; ++c)
      wv[i][c] = *reinterpret_cast<const u32x4*>(q.wout_h + (long long)min(wave + i * kAttnWaves, nm - 1) * H +
                                                 min((lane + 64 * c) * 8, H - 8));
  for (int h8 = tid; h8 < H / 8; h8 += kAttnThreads) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(yq + h8 * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) { hq[h8 * 8 + 2 * k] = bflo(v[k]); hq[h8 * 8 + 2 * k + 1] = bfhi(v[k]); }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kRows; ++i) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kPc; ++c) {
      const int k = (lane + 64 * c) * 8;
      if (k < H) {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) s += bflo(wv[i][c][k2]) * hq[k + 2 * k2] + bfhi(wv[i][c][k2]) * hq[k + 2 * k2 + 1];
      }
    }
    s = wave_sum_dpp(s);
    const int m = wave + i * kAttnWaves;
    if (lane == 0 && m < nm) q.mh[(long long)b * nm + m] = s;
  }
}

// grid (kLocParts + 1, B): the location-sensitive scores (attn_decoder.hip) + the cell-output half of the frame
__global__ __launch_bounds__(kAttnThreads) void ti_scores_kernel(AdAttn p, AdLoc x, TiTail q) {
  if (q.state[1] != 0) return;
  if (blockIdx.x == kLocParts) { if (!(q.dbg & 1)) ti_frame_hpart(p, q); }
  else if (!(q.dbg & 2)) ad_loc_scores_body(p, x);
}

}  // namespace os2s

using namespace os2s;

extern "C" size_t os2s_tacotron_infer_state_ints(int B) { return (size_t)4 + 2 * (size_t)B; }

static int ti_check(const os2s_tacotron_infer_t* x) {
  OS2S_REQUIRE(x && x->loop);
  const os2s_attn_decoder_t* d = x->loop;
  const int rc = ad_check(d);
  if (rc != OS2S_OK) return rc;
  OS2S_REQUIRE(x->P >= 8 && x->n_mel >= 8 && x->x_seq && x->mel && x->stop && x->state && x->pv && x->wout_h);
  OS2S_REQUIRE(x->wp1 && x->bp1 && x->wp2 && x->bp2 && x->bout && x->wstop && x->bstop && x->bias0);
  OS2S_REQUIRE((x->w0x != nullptr) != (x->w0x8 != nullptr));
  if (x->w0x8) OS2S_REQUIRE(x->w0x8_scale && (d->L == 1 || (d->wcat8[1] && d->wcat8_scale[1])));
  OS2S_REQUIRE(x->prenet_keep > 0.f && x->prenet_keep <= 1.f);
  // what the step kernels are built for (anything else: the caller drives os2s_attn_decoder_fwd step by step)
  if (d->score_mode != 2 || !loc_split(d) || d->B > 32 || d->H % 64 || d->M % 64 || x->P % 64 || x->n_mel % 8 ||
      x->n_mel > 128 || x->P > 1024 || d->attn_in_keep < 1.f || d->out_keep < 1.f || d->tgt_len)
    return OS2S_ERR_UNSUPPORTED;
  if (x->P > 256 || x->n_mel / 8 > kTiW1Pieces * (kTiCtxThreads / x->P) || x->n_mel > 16 * kAttnWaves ||
      d->H > 1024 || !x->mh)
    return OS2S_ERR_UNSUPPORTED;
  if (ti_tail_lds_floats(d->S, x->P, x->n_mel) * sizeof(float) > 64 * 1024) return OS2S_ERR_UNSUPPORTED;
  return OS2S_OK;
}

extern "C" int os2s_tacotron_infer_supported(const os2s_tacotron_infer_t* x) { return ti_check(x) == OS2S_OK; }

// units per workgroup: 4 (16 gate rows: H / 4 workgroups) or 8 (32 rows: half the workgroups, half the reads of the
// state rows); OS2S_TI_ROWS = 16 | 32 overrides the default (experiments)
static int ti_rows() {
  static const int v = [] { const char* e = getenv("OS2S_TI_ROWS"); return e ? atoi(e) : 16; }();
  return v == 32 ? 32 : 16;
}

template <bool FP8>
static int ti_launch_lstm(hipStream_t stream, const TiLstm& c) {
  const dim3 blk(64 * kTiWaves);
  if (ti_rows() == 32 && c.H % 8 == 0) {
    const dim3 grid(ceil_div(c.H, 8));
    if (c.B <= 16) { OS2S_LAUNCH((ti_lstm_kernel<FP8, 2, 1>), grid, blk, 0, stream, c); }
    else { OS2S_LAUNCH((ti_lstm_kernel<FP8, 2, 2>), grid, blk, 0, stream, c); }
    return OS2S_OK;
  }
  const dim3 grid(ceil_div(c.H, 4));
  if (c.B <= 16) { OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, 1>), grid, blk, 0, stream, c); }
  else { OS2S_LAUNCH((ti_lstm_kernel<FP8, 1, 2>), grid, blk, 0, stream, c); }
  return OS2S_OK;
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
  int ctx_parts = kLocCtxParts;
  while (ctx_parts > 1 && M % (8 * ctx_parts)) ctx_parts >>= 1;
  const int ncg = M / (8 * ctx_parts);
  const int nsp = kTiCtxThreads / ncg < 32 ? kTiCtxThreads / ncg : 32;
  if (ncg > 256) return OS2S_ERR_UNSUPPORTED;
  const size_t lds_s = loc_fwd_lds_floats(H, d->S) * sizeof(float);
  const size_t lds_c = std::max(((size_t)d->S + 32 + (size_t)nsp * ncg * 8) * sizeof(float),
                                ti_tail_lds_floats(d->S, P, x->n_mel) * sizeof(float));
  TiTail q;
  q.P = P; q.n_mel = x->n_mel; q.mask_seq = x->mask_decoder_sequence; q.first = 0;
  { static const int dbg = [] { const char* e = getenv("OS2S_TI_DEBUG"); return e ? atoi(e) : 0; }(); q.dbg = dbg; }
  q.keep = x->prenet_keep; q.seed[0] = x->prenet_seed[0]; q.seed[1] = x->prenet_seed[1];
  q.wp1 = (const bf16_t*)x->wp1; q.bp1 = x->bp1; q.wp2 = (const bf16_t*)x->wp2; q.bp2 = x->bp2;
  q.wout_h = (const bf16_t*)x->wout_h; q.pv = x->pv; q.bout = x->bout;
  q.wstop = (const bf16_t*)x->wstop; q.bstop = x->bstop; q.mh = x->mh;
  q.x_seq = (bf16_t*)x->x_seq; q.mel = (bf16_t*)x->mel; q.stop = x->stop; q.state = x->state;
  if (t_begin == 0) {
    const int n = (d->loc_k + 1) * d->U;
    OS2S_LAUNCH(ad_fold_location_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, d->conv_w, d->conv_b,
                d->dense_w, d->loc_k, d->loc_f, d->U, d->loc_ws);
    TiTail q0 = q;
    q0.first = 1;
    at.t = 0;
    OS2S_LAUNCH(ti_context_kernel, dim3(ctx_parts + 1, B), dim3(kTiCtxThreads), lds_c, stream, at, lx, q0, ctx_parts,
                ncg, nsp);
  }
  const bool fp8 = x->w0x8 != nullptr;
  for (int t = t_begin; t < t_end; ++t) {
    for (int l = 0; l < L; ++l) {
      TiLstm c;
      c.B = B; c.H = H; c.forget_bias = d->forget_bias; c.state = x->state;
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
    OS2S_LAUNCH(ti_context_kernel, dim3(ctx_parts + 1, B), dim3(kTiCtxThreads), lds_c, stream, at, lx, q, ctx_parts,
                ncg, nsp);
  }
  return OS2S_OK;
}
