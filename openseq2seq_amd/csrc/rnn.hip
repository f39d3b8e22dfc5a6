// Recurrent layers (GRU / LSTM), forward and backward-through-time, gfx950.
//
// Reference cells on the hot path:
//   * tf.contrib.cudnn_rnn.CudnnGRU (DeepSpeech2: 5 bidirectional layers of 800 units,
//     open_seq2seq/encoders/ds2_encoder.py:294-328): cuDNN gate form
//       r = s(Wr x + Rr h + bWr + bRr), z = s(Wz x + Rz h + bWz + bRz),
//       n = tanh(Wn x + bWn + r * (Rn h + bRn)), h' = (1 - z) * n + z * h
//   * tf.contrib.cudnn_rnn.CudnnLSTM (Tacotron2 encoder, tacotron2_encoder.py:254-263):
//       gate order i, f, g, o, two bias vectors
//   * tf.nn.rnn_cell.LSTMCell (NMT encoder/decoder and Tacotron2 decoder,
//     rnn_encoders.py:292-300, parts/rnns/utils.py:17-89): gate order i, j, f, o,
//       c' = c * s(f + forget_bias) + s(i) * tanh(j), h' = tanh(c') * s(o)
//
// Structure (MI355X-first): the input projection of ALL time steps is one MFMA GEMM
// (os2s_conv1d_fwd, K = 1) done by the caller; only the recurrence runs here. One launch
// per time step — the C entry point owns the time loop, so the host pays one call per
// layer, not per step. Per step, one WAVE owns 32 hidden units x 32 batch columns: it
// streams its slice of the recurrent weights (k-contiguous rows, 16-byte loads straight
// into MFMA A fragments, no LDS), multiplies with the previous hidden state (bf16 copy) on
// the matrix cores in the swapped orientation (rows = hidden units), so all G gates of a
// (unit, sample) pair land in the same lane/register and the cell non-linearity, the state
// update and every store are lane-local. The step is latency-bound (B is 16..128), so the
// reduction dimension is split over the waves of the workgroup with batched loads
// (rnn_tile.hpp), and the two directions of a bidirectional layer run in ONE launch
// (blockIdx.z = direction). The backward step fuses "dh_{t-1} = dgates_t . Wh"
// with the gate derivatives of step t-1 the same way, using the transposed weight copy.
// dWh, dWx, dX and the bias gradients are large GEMMs over the saved gate gradients, done
// by the caller with the conv/wgrad kernels (h_{t-1} as a time-shifted operand).
#include "os2s_common.hpp"
#include "rnn_tile.hpp"

namespace os2s {

enum { kGruCudnn = 0, kLstmCudnn = 1, kLstmTf = 2 };
constexpr int kRnnWaves = 8;

struct RnnDirFwd {
  const bf16_t* gx;        // [B, T, G*H]  input projections (+ input bias)
  const bf16_t* wh;        // [G*H, H]
  const float* bh;         // [G*H] recurrent bias or null
  const bf16_t* h_prev16;  // [B, H]
  bf16_t* h_next16;        // [B, H]
  float* h32;              // [B, H] fp32 state (in/out)
  float* c32;              // [B, H] (LSTM)
  bf16_t* y;               // outputs of this direction: row (b,t) at y + (b*T+t)*ldy
  long long ldy;
  bf16_t* gates;           // [B, T, Gs*H] saved activations (Gs = 4)
  float* c_seq;            // [B, T, H] saved cell states (LSTM)
  int reverse;
};
struct RnnStepArgs {
  int cell, B, T, H, G, step;
  float forget_bias;
  const int32_t* lens;     // [B] or null
  RnnDirFwd d[2];          // blockIdx.z selects the direction
};

// t index processed by sample b at loop step s, or -1 if the sample is past its length
__device__ __forceinline__ int time_of(int s, int len, int reverse) {
  if (s >= len) return -1;
  return reverse ? len - 1 - s : s;
}

// One workgroup = 8 hidden units x 32 samples of one direction: the 32 MFMA tile rows are
// (gate, unit) pairs (row = 8*gate + unit; G = 3 leaves rows 24..31 empty), so H/8 workgroups
// stream disjoint slices of the recurrent weights and after the split-K reduction a lane
// owns all gates of one (unit, sample).
template <int G>
__global__ __launch_bounds__(64 * kRnnWaves) void rnn_step_fwd_kernel(RnnStepArgs pa) {
  __shared__ float red[kRnnWaves * 16 * 64];
  const RnnDirFwd& p = pa.d[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
  const int H = pa.H;
  // the epilogue operands of this lane's (unit, sample) are fetched BEFORE the GEMM so that
  // their round trip overlaps the weight stream instead of following it
  const int eb = b0 + l31, ej = j0 + (wave & 3) + 4 * lhi;
  const bool evalid = wave < 4 && eb < pa.B && ej < H;
  int t = -1;
  float hprev = 0.f, cprev = 0.f, pre[G], bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G; ++g) pre[g] = 0.f;
  if (evalid) {
    const int len = pa.lens ? min(max(pa.lens[eb], 0), pa.T) : pa.T;
    t = time_of(pa.step, len, p.reverse);
    hprev = p.h32[(long long)eb * H + ej];
    if (pa.cell != kGruCudnn) cprev = p.c32[(long long)eb * H + ej];
    if (t >= 0) {
      const long long row = (long long)eb * pa.T + t;
#pragma unroll
      for (int g = 0; g < G; ++g) pre[g] = bf2f(p.gx[row * (G * H) + (long long)g * H + ej]);
    }
    if (p.bh) {
#pragma unroll
      for (int g = 0; g < G; ++g) bv[g] = p.bh[g * H + ej];
    }
  }
  f32x16 accw;
#pragma unroll
  for (int e = 0; e < 16; ++e) accw[e] = 0.f;
  {
    const int g = l31 >> 3, uu = min(j0 + (l31 & 7), H - 1);
    const bf16_t* wrow = g < G ? p.wh + ((long long)g * H + uu) * H : nullptr;
    const int brow = b0 + l31;
    const bf16_t* irow = brow < pa.B ? p.h_prev16 + (long long)brow * H : nullptr;
    tile_gemm_prefetch<kRnnWaves, 8>(wrow, irow, H, accw, p.wh);
  }
  float acc[4];
  tile_reduce_units<kRnnWaves>(accw, red, acc);
  if (!evalid) return;
  const int b = eb, j = ej;
  float hnew = hprev;      // past the sequence end the state passes through (dynamic_rnn)
  if (t >= 0) {
    const long long row = (long long)b * pa.T + t;
    float sv[4];
    if (pa.cell == kGruCudnn) {
      const float rg = sigmoidf_(pre[0] + acc[0] + bv[0]);
      const float zg = sigmoidf_(pre[1] + acc[1] + bv[1]);
      const float hn = acc[2] + bv[2];
      const float ng = tanhf(pre[2] + rg * hn);
      hnew = (1.f - zg) * ng + zg * hprev;
      sv[0] = rg; sv[1] = zg; sv[2] = ng; sv[3] = hn;
    } else {
      const float a0 = pre[0] + acc[0] + bv[0], a1 = pre[1] + acc[1] + bv[1];
      const float a2 = pre[2] + acc[2] + bv[2], a3 = pre[G - 1] + acc[3] + bv[3];
      float ig, fg, gg, og;
      if (pa.cell == kLstmCudnn) { ig = sigmoidf_(a0); fg = sigmoidf_(a1); gg = tanhf(a2); og = sigmoidf_(a3); }
      else { ig = sigmoidf_(a0); gg = tanhf(a1); fg = sigmoidf_(a2 + pa.forget_bias); og = sigmoidf_(a3); }
      const float cn = cprev * fg + ig * gg;
      hnew = tanhf(cn) * og;
      sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og;
      p.c32[(long long)b * H + j] = cn;
      if (p.c_seq) p.c_seq[row * H + j] = cn;
    }
    if (p.gates) {
      bf16_t* gp = p.gates + row * (4 * H) + j;
      gp[0] = f2bf(sv[0]); gp[H] = f2bf(sv[1]); gp[2 * H] = f2bf(sv[2]); gp[3 * H] = f2bf(sv[3]);
    }
    p.y[row * p.ldy + j] = f2bf(hnew);
  }
  p.h32[(long long)b * H + j] = hnew;
  p.h_next16[(long long)b * H + j] = f2bf(hnew);
}

// ---------------------------------------------------------------------------
// backward step: dh(s) = dY[t(s)] + dgates(s+1) . Wh + direct terms; then the gate
// derivatives of step s, written to dgx[b, t, :] and to the compact [B, G*H] buffer the
// next (earlier) step multiplies with Wh^T.
// ---------------------------------------------------------------------------
struct RnnDirBwd {
  const bf16_t* whT;        // [H, G*H]  transposed recurrent weights
  const bf16_t* dy;         // row (b,t) at dy + (b*T+t)*lddy
  long long lddy, ldy;
  const bf16_t* gates;      // [B, T, 4*H] saved activations
  const float* c_seq;       // [B, T, H] (LSTM)
  const bf16_t* y;          // [B, T, H] forward outputs (GRU needs h_{t-1})
  const bf16_t* dg_next;    // [B, G*H] gate grads of loop step s+1 (zero if first)
  bf16_t* dg_cur;           // [B, G*H]
  bf16_t* dgx;              // [B, T, G*H]  gate grads w.r.t. the input projections
  bf16_t* dgr;              // [B, T, G*H]  same on the recurrent side (GRU: differs for n) or null
  float* dh_carry;          // [B, H] fp32: direct (non-matmul) part of dh flowing to s-1
  float* dc_carry;          // [B, H] fp32 (LSTM)
  int reverse;
};
struct RnnBwdArgs {
  int cell, B, T, H, G, step, first;
  float forget_bias;
  const int32_t* lens;
  RnnDirBwd d[2];
};

// One workgroup = ROWS hidden units x 32 samples. Every load of a wave's share of the reduction
// is issued before its first MFMA (rnn_tile.hpp; round 1 ran five dependent load rounds over the
// 2400-deep reduction of the DeepSpeech2 GRU: 10.5 us per launch against 6.7 us forward).
// ROWS = 8 (rows 8..31 of the MFMA tile are padding) when the 32-row grid would leave most of
// the chip idle (H = 800, B = 32: 25 workgroups per direction -> 100); ROWS = 32 otherwise — every
// workgroup reads the whole [32, G*H] gate-gradient block of its batch tile, so a finer row
// split multiplies that traffic.
template <int G, int ROWS>
__global__ __launch_bounds__(64 * kRnnWaves) void rnn_step_bwd_kernel(RnnBwdArgs pa) {
  __shared__ float red[kRnnWaves * (ROWS == 8 ? 4 : 16) * 64];
  const RnnDirBwd& p = pa.d[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int j0 = blockIdx.x * ROWS, b0 = blockIdx.y * 32;
  const int H = pa.H, GH = G * pa.H;
  // the epilogue operands of the (unit, sample) pairs of waves 0 .. ROWS/8 - 1 are fetched BEFORE
  // the GEMM so that their round trip overlaps the weight stream instead of following it
  const int b = b0 + l31, j = j0 + 8 * wave + 4 * lhi;
  const bool evalid = wave < ROWS / 8 && b < pa.B && j < H;
  int t = -1;
  if (evalid) {
    const int len = pa.lens ? min(max(pa.lens[b], 0), pa.T) : pa.T;
    t = time_of(pa.step, len, p.reverse);
  }
  const bool act = evalid && t >= 0;
  const int tp = p.reverse ? t + 1 : t - 1;    // previous time index in processing order (s-1)
  const bool has_prev = pa.step > 0;
  const long long row = (long long)b * pa.T + (act ? t : 0);
  f32x4 carry = {0.f, 0.f, 0.f, 0.f}, dcarry = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f},
        cprev = {0.f, 0.f, 0.f, 0.f};
  u32x2 dyv = {0u, 0u}, hv = {0u, 0u}, gv[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
  if (act) {
    if (!pa.first) carry = *reinterpret_cast<const f32x4*>(p.dh_carry + (long long)b * H + j);
    dyv = *reinterpret_cast<const u32x2*>(p.dy + row * p.lddy + j);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      gv[g] = *reinterpret_cast<const u32x2*>(p.gates + row * (4 * H) + (long long)g * H + j);
    if (pa.cell == kGruCudnn) {
      if (has_prev) hv = *reinterpret_cast<const u32x2*>(p.y + ((long long)b * pa.T + tp) * p.ldy + j);
    } else {
      if (!pa.first) dcarry = *reinterpret_cast<const f32x4*>(p.dc_carry + (long long)b * H + j);
      cv = *reinterpret_cast<const f32x4*>(p.c_seq + row * H + j);
      if (has_prev) cprev = *reinterpret_cast<const f32x4*>(p.c_seq + ((long long)b * pa.T + tp) * H + j);
    }
  }
  float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  if (!pa.first) {
    f32x16 accw;
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[e] = 0.f;
    const bf16_t* wrow = (l31 < ROWS && j0 + l31 < H) ? p.whT + (long long)(j0 + l31) * GH : nullptr;
    const int brow = b0 + l31;
    const bf16_t* irow = brow < pa.B ? p.dg_next + (long long)brow * GH : nullptr;
    // G = 3, H = 800: 150 k-slices = 19 per wave in one round; G = 4, H = 1024: two rounds of 16
    tile_gemm_prefetch<kRnnWaves, G == 3 ? 19 : 16>(wrow, irow, GH, accw, p.whT);
    if (ROWS == 8) tile_reduce_rows8<kRnnWaves>(accw, red, acc[0]);
    else tile_reduce_rows<kRnnWaves>(accw, red, acc[0]);
  }
  if (!evalid) return;
  {
    bf16_t* dgc = p.dg_cur + (long long)b * GH;
    if (t < 0) {
      // inactive sample: its dg_next is zero (nothing was active later either) and the
      // carries stay untouched; publish zero gate grads
#pragma unroll
      for (int g = 0; g < G; ++g) {
        u32x2 z = {0u, 0u};
        *reinterpret_cast<u32x2*>(dgc + (long long)g * H + j) = z;
      }
      return;
    }
    float dh[4] = {bflo(dyv[0]) + acc[0][0] + carry[0], bfhi(dyv[0]) + acc[0][1] + carry[1],
                   bflo(dyv[1]) + acc[0][2] + carry[2], bfhi(dyv[1]) + acc[0][3] + carry[3]};
    float sv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sv[g][0] = bflo(gv[g][0]); sv[g][1] = bfhi(gv[g][0]); sv[g][2] = bflo(gv[g][1]); sv[g][3] = bfhi(gv[g][1]);
    }
    float dpre[G][4];
    f32x4 ncarry;
    if (pa.cell == kGruCudnn) {
      const float hprev[4] = {bflo(hv[0]), bfhi(hv[0]), bflo(hv[1]), bfhi(hv[1])};   // zero at step 0
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rg = sv[0][e], zg = sv[1][e], ng = sv[2][e], hn = sv[3][e];
        const float dn = dh[e] * (1.f - zg);
        const float dz = dh[e] * (hprev[e] - ng);
        const float dnpre = dn * (1.f - ng * ng);
        const float dr = dnpre * hn;
        dpre[0][e] = dr * rg * (1.f - rg);            // d(pre r)
        dpre[1][e] = dz * zg * (1.f - zg);            // d(pre z)
        dpre[2][e] = dnpre;                           // d(Wn x + bWn)   (input side)
        ncarry[e] = dh[e] * zg;                       // direct path to h_{t-1}
        sv[3][e] = dnpre * rg;                        // d(Rn h + bRn)   (recurrent side)
      }
    } else {
      f32x4 ndc;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ig = sv[0][e], fg = sv[1][e], gg = sv[2][e], og = sv[3][e];
        const float tc = tanhf(cv[e]);
        const float dox = dh[e] * tc * og * (1.f - og);
        const float dc = dh[e] * og * (1.f - tc * tc) + dcarry[e];
        const float dix = dc * gg * ig * (1.f - ig);
        const float dfx = dc * cprev[e] * fg * (1.f - fg);
        const float dgx_ = dc * ig * (1.f - gg * gg);
        ndc[e] = dc * fg;
        if (pa.cell == kLstmCudnn) { dpre[0][e] = dix; dpre[1][e] = dfx; dpre[2][e] = dgx_; dpre[3 % G][e] = dox; }
        else { dpre[0][e] = dix; dpre[1][e] = dgx_; dpre[2][e] = dfx; dpre[3 % G][e] = dox; }
        ncarry[e] = 0.f;
      }
      *reinterpret_cast<f32x4*>(p.dc_carry + (long long)b * H + j) = ndc;
    }
    *reinterpret_cast<f32x4*>(p.dh_carry + (long long)b * H + j) = ncarry;
    // gate gradients: dgx (w.r.t. the input projections) and the recurrent-side copy
#pragma unroll
    for (int g = 0; g < G; ++g) {
      u32x2 pk;
      pk[0] = pack2bf(dpre[g][0], dpre[g][1]);
      pk[1] = pack2bf(dpre[g][2], dpre[g][3]);
      *reinterpret_cast<u32x2*>(p.dgx + ((long long)b * pa.T + t) * GH + (long long)g * H + j) = pk;
      u32x2 rk = pk;
      if (pa.cell == kGruCudnn && g == 2) {   // the recurrent n-gate term sees r * dnpre
        rk[0] = pack2bf(sv[3][0], sv[3][1]);
        rk[1] = pack2bf(sv[3][2], sv[3][3]);
      }
      *reinterpret_cast<u32x2*>(dgc + (long long)g * H + j) = rk;
      if (p.dgr)
        *reinterpret_cast<u32x2*>(p.dgr + ((long long)b * pa.T + t) * GH + (long long)g * H + j) = rk;
    }
  }
}

}  // namespace os2s

using namespace os2s;

static int rnn_gates(int cell) { return cell == kGruCudnn ? 3 : 4; }

// rnn_xcd.hip: the persistent, XCD-local forward pass of a cuDNN-form GRU layer
extern "C" size_t os2s_gru_xcd_workspace_bytes(int B, int H);
bool gru_xcd_supported(int B, int T, int H, int ndir);
int launch_gru_xcd_fwd(hipStream_t stream, int ndir, const os2s_rnn_dir_fwd_t* dirs, float* const* h32,
                       void* const* xws, int* flags, const int32_t* lens, int B, int T, int H);
extern "C" size_t os2s_gru_xcd_bwd_workspace_bytes(int B, int H);
bool gru_xcd_bwd_supported(int B, int T, int H, int ndir);
int launch_gru_xcd_bwd(hipStream_t stream, int ndir, const os2s_rnn_dir_bwd_t* dirs, void* const* xws, int* flags,
                       const int32_t* lens, int B, int T, int H);

// workspace per direction: h16[2][B,H] bf16 + h32[B,H] + c32[B,H] fp32 (+ the exchange buffer and the
// flags of the persistent GRU kernel)
static size_t rnn_fwd_state_bytes(int B, int H) { return (size_t)B * H * (2 * 2 + 4 + 4) + 256; }
extern "C" size_t os2s_rnn_fwd_workspace_bytes(int B, int H) {
  return rnn_fwd_state_bytes(B, H) + os2s_gru_xcd_workspace_bytes(B, H) + 256;
}

extern "C" int os2s_rnn_layer_fwd_multi(os2s_stream_t stream_, int cell, int ndir,
                                        const os2s_rnn_dir_fwd_t* dirs, const int32_t* lens,
                                        int B, int T, int H, float forget_bias, void* workspace,
                                        size_t workspace_bytes) {
  OS2S_REQUIRE(dirs && workspace && (ndir == 1 || ndir == 2));
  OS2S_REQUIRE(B >= 1 && T >= 1 && H >= 8 && H % 8 == 0 && cell >= 0 && cell <= 2);
  const size_t per_dir = os2s_rnn_fwd_workspace_bytes(B, H);
  if (workspace_bytes < per_dir * ndir) return OS2S_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (hipMemsetAsync(workspace, 0, per_dir * ndir, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  RnnStepArgs a;
  a.cell = cell; a.B = B; a.T = T; a.H = H; a.G = rnn_gates(cell);
  a.forget_bias = forget_bias; a.lens = lens;
  bf16_t* h16[2][2];
  for (int d = 0; d < ndir; ++d) {
    const os2s_rnn_dir_fwd_t& s = dirs[d];
    OS2S_REQUIRE(s.gx && s.wh && s.y && s.ldy >= H && s.ldy % 4 == 0);
    char* ws = (char*)workspace + per_dir * d;
    h16[d][0] = (bf16_t*)ws; ws += (size_t)B * H * 2;
    h16[d][1] = (bf16_t*)ws; ws += (size_t)B * H * 2;
    RnnDirFwd& k = a.d[d];
    k.gx = (const bf16_t*)s.gx; k.wh = (const bf16_t*)s.wh; k.bh = s.bh;
    k.h32 = (float*)ws; ws += (size_t)B * H * 4;
    k.c32 = (float*)ws;
    k.y = (bf16_t*)s.y; k.ldy = s.ldy; k.gates = (bf16_t*)s.gates; k.c_seq = s.c_seq;
    k.reverse = s.reverse;
    if (lens) {   // outputs past the sequence ends are zero (rows are ldy apart)
      if (hipMemset2DAsync(s.y, (size_t)s.ldy * 2, 0, (size_t)H * 2, (size_t)B * T, stream) != hipSuccess)
        return OS2S_ERR_LAUNCH;
    }
  }
  if (cell == kGruCudnn && gru_xcd_supported(B, T, H, ndir)) {
    // ONE launch for all T steps: weights stationary in registers, hidden state exchanged through the
    // L2 of the XCD a direction lives on (rnn_xcd.hip)
    float* h32p[2];
    void* xws[2];
    for (int d = 0; d < ndir; ++d) {
      h32p[d] = a.d[d].h32;
      xws[d] = (char*)workspace + per_dir * d + rnn_fwd_state_bytes(B, H);
    }
    int* flags = (int*)((char*)workspace + per_dir * 0 + rnn_fwd_state_bytes(B, H) + os2s_gru_xcd_workspace_bytes(B, H));
    return launch_gru_xcd_fwd(stream, ndir, dirs, h32p, xws, flags, lens, B, T, H);
  }
  if (ndir == 1) a.d[1] = a.d[0];
  dim3 grid(ceil_div(H, 8), ceil_div(B, 32), ndir);
  for (int s = 0; s < T; ++s) {
    a.step = s;
    for (int d = 0; d < ndir; ++d) {
      a.d[d].h_prev16 = h16[d][s & 1];
      a.d[d].h_next16 = h16[d][(s & 1) ^ 1];
    }
    if (a.G == 3) { OS2S_LAUNCH(rnn_step_fwd_kernel<3>, grid, dim3(64 * kRnnWaves), 0, stream, a); }
    else { OS2S_LAUNCH(rnn_step_fwd_kernel<4>, grid, dim3(64 * kRnnWaves), 0, stream, a); }
  }
  return OS2S_OK;
}

extern "C" int os2s_rnn_layer_fwd(os2s_stream_t stream_, int cell, const uint16_t* gx,
                                  const uint16_t* wh, const float* bh, const int32_t* lens, int B,
                                  int T, int H, int reverse, float forget_bias, uint16_t* y,
                                  long long ldy, uint16_t* gates, float* c_seq, void* workspace,
                                  size_t workspace_bytes) {
  os2s_rnn_dir_fwd_t d;
  d.gx = gx; d.wh = wh; d.bh = bh; d.y = y; d.ldy = ldy; d.gates = gates; d.c_seq = c_seq;
  d.reverse = reverse;
  return os2s_rnn_layer_fwd_multi(stream_, cell, 1, &d, lens, B, T, H, forget_bias, workspace,
                                  workspace_bytes);
}

// workspace per direction: dg[2][B,G*H] bf16 + dh_carry[B,H] + dc_carry[B,H] fp32 (+ the exchange
// buffer and the flags of the persistent GRU kernel)
static size_t rnn_bwd_state_bytes(int B, int H) { return (size_t)B * H * (2 * 4 * 2 + 4 + 4) + 256; }
extern "C" size_t os2s_rnn_bwd_workspace_bytes(int B, int H) {
  return rnn_bwd_state_bytes(B, H) + os2s_gru_xcd_bwd_workspace_bytes(B, H) + 256;
}

extern "C" int os2s_rnn_layer_bwd_multi(os2s_stream_t stream_, int cell, int ndir,
                                        const os2s_rnn_dir_bwd_t* dirs, const int32_t* lens,
                                        int B, int T, int H, float forget_bias, void* workspace,
                                        size_t workspace_bytes) {
  OS2S_REQUIRE(dirs && workspace && (ndir == 1 || ndir == 2));
  OS2S_REQUIRE(B >= 1 && T >= 1 && H % 8 == 0 && cell >= 0 && cell <= 2);
  const size_t per_dir = os2s_rnn_bwd_workspace_bytes(B, H);
  if (workspace_bytes < per_dir * ndir) return OS2S_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int G = rnn_gates(cell);
  if (hipMemsetAsync(workspace, 0, per_dir * ndir, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  RnnBwdArgs a;
  a.cell = cell; a.B = B; a.T = T; a.H = H; a.G = G; a.lens = lens; a.forget_bias = forget_bias;
  bf16_t* dg[2][2];
  for (int d = 0; d < ndir; ++d) {
    const os2s_rnn_dir_bwd_t& s = dirs[d];
    OS2S_REQUIRE(s.whT && s.dy && s.y && s.gates && s.dgx);
    if (cell != kGruCudnn) OS2S_REQUIRE(s.c_seq);
    char* ws = (char*)workspace + per_dir * d;
    dg[d][0] = (bf16_t*)ws; ws += (size_t)B * 4 * H * 2;
    dg[d][1] = (bf16_t*)ws; ws += (size_t)B * 4 * H * 2;
    RnnDirBwd& k = a.d[d];
    k.whT = (const bf16_t*)s.whT; k.dy = (const bf16_t*)s.dy; k.lddy = s.lddy; k.ldy = s.ldy;
    k.gates = (const bf16_t*)s.gates; k.c_seq = s.c_seq; k.y = (const bf16_t*)s.y;
    k.dgx = (bf16_t*)s.dgx; k.dgr = (bf16_t*)s.dgr;
    k.dh_carry = (float*)ws; ws += (size_t)B * H * 4;
    k.dc_carry = (float*)ws;
    k.reverse = s.reverse;
    // gate gradients of frames past the sequence ends are zero
    if (hipMemsetAsync(s.dgx, 0, (size_t)B * T * G * H * 2, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
    if (s.dgr && hipMemsetAsync(s.dgr, 0, (size_t)B * T * G * H * 2, stream) != hipSuccess)
      return OS2S_ERR_LAUNCH;
  }
  if (cell == kGruCudnn && gru_xcd_bwd_supported(B, T, H, ndir)) {
    void* xws[2];
    for (int d = 0; d < ndir; ++d) xws[d] = (char*)workspace + per_dir * d + rnn_bwd_state_bytes(B, H);
    int* flags = (int*)((char*)workspace + rnn_bwd_state_bytes(B, H) + os2s_gru_xcd_bwd_workspace_bytes(B, H));
    return launch_gru_xcd_bwd(stream, ndir, dirs, xws, flags, lens, B, T, H);
  }
  if (ndir == 1) a.d[1] = a.d[0];
  const bool rows8 = ceil_div(H, 32) * ceil_div(B, 32) * ndir < 128;
  dim3 grid(ceil_div(H, rows8 ? 8 : 32), ceil_div(B, 32), ndir);
  for (int s = T - 1; s >= 0; --s) {
    a.step = s;
    a.first = (s == T - 1);
    const int par = (T - 1 - s) & 1;
    for (int d = 0; d < ndir; ++d) {
      a.d[d].dg_next = dg[d][par];
      a.d[d].dg_cur = dg[d][par ^ 1];
    }
    if (G == 3 && rows8) { OS2S_LAUNCH((rnn_step_bwd_kernel<3, 8>), grid, dim3(64 * kRnnWaves), 0, stream, a); }
    else if (G == 3) { OS2S_LAUNCH((rnn_step_bwd_kernel<3, 32>), grid, dim3(64 * kRnnWaves), 0, stream, a); }
    else if (rows8) { OS2S_LAUNCH((rnn_step_bwd_kernel<4, 8>), grid, dim3(64 * kRnnWaves), 0, stream, a); }
    else { OS2S_LAUNCH((rnn_step_bwd_kernel<4, 32>), grid, dim3(64 * kRnnWaves), 0, stream, a); }
  }
  return OS2S_OK;
}

extern "C" int os2s_rnn_layer_bwd(os2s_stream_t stream_, int cell, const uint16_t* whT,
                                  const int32_t* lens, const uint16_t* dy, long long lddy,
                                  const uint16_t* y, long long ldy, const uint16_t* gates,
                                  const float* c_seq, int B, int T, int H,
                                  int reverse, float forget_bias, uint16_t* dgx, uint16_t* dgr,
                                  void* workspace, size_t workspace_bytes) {
  os2s_rnn_dir_bwd_t d;
  d.whT = whT; d.dy = dy; d.lddy = lddy; d.y = y; d.ldy = ldy; d.gates = gates; d.c_seq = c_seq;
  d.dgx = dgx; d.dgr = dgr; d.reverse = reverse;
  return os2s_rnn_layer_bwd_multi(stream_, cell, 1, &d, lens, B, T, H, forget_bias, workspace,
                                  workspace_bytes);
}
