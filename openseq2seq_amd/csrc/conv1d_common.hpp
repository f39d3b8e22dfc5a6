// Shared by the implicit-GEMM convolution kernels (conv1d_igemm.hip) and the plain GEMM
// (gemm_pp.hip): kernel argument block, LDS-DMA helper and the fused epilogue.
#pragma once
#include <type_traits>

#include "os2s_common.hpp"

namespace os2s {

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const int32_t* in_len;
  const int32_t* out_len;   // rows t >= out_len[b] of the OUTPUT are never read by the caller
  const float* bias;
  float* stats;
  int B, Tin, Tout, Cin, Cout, K, stride, dil, padL;
  long long x_sb, x_st, y_sb, y_st;
  int out_f32, accumulate;
  int mtiles_per_b, MT, MT8, NT, nchunks, R, Rpad;
  // fused epilogue: y = residual + dropout(act(acc + bias))
  int act;                      // 0 none, 1 relu, 3 relu capped at 20
  float keep_prob;              // 1 = no dropout
  unsigned long long seed;
  const bf16_t* residual;       // same layout/strides as y (bf16 output only) or null
  // backward of act + dropout fused into the data-gradient GEMM that produces dy (gemm_pp.hip,
  // os2s_gemm_nt_mask_ws): y = (mask_ref > 0) ? acc * mask_scale : 0, mask_ref = the SAVED forward
  // output relu(.)-then-dropout of the layer (same layout/strides as y); `stats` then holds the
  // column sums of the masked gradient (= the bias gradient partials). Null = off.
  const bf16_t* mask_ref;
  float mask_scale;
  // with mask_ref: stats[.., 1, c] = sum over the window's rows of y[row, c] * stat_ref[row, c]
  // instead of the sum of squares (stat_ref in the layout of y) — the BatchNorm backward partials
  // sum(dz), sum(dz * conv_out) of the PRODUCING layer when this launch is the data gradient that
  // finalises its output gradient (os2s_conv1d_dgrad_bnact_ws). Null = sums of squares.
  const bf16_t* stat_ref;
  // ping-pong kernel only: split-unit workspace (fp32 partial tiles + one ticket per split unit)
  float* ws_slabs;
  int* ws_cnt;
  int ws_nslabs, ncu;
  int force_split;              // experiment hook: > 0 forces the tail split factor
  unsigned long long* dbg;      // experiment hook: slot time stamps [4 wg][2 waves][48 steps][9]
  int dbg_fixed_w;              // experiment hook: every step reads the weight tile of step 0
  // ping-pong convolution only: tile shape of the launch. -1 = chosen on the device from the live-window
  // count (pp_choose_tile), 0 = two windows x 256 columns, 2 / 3 = two / three windows x 128 columns.
  // pp_ok2 / pp_ok3: the narrow tiles fit this layer (LDS, K), pp_cost = fitted microseconds per
  // 64-deep step of the three tiles (the device-side choice is a pure function of these and in_len).
  int pp_tile = 0, pp_ok2 = 0, pp_ok3 = 0;
  float pp_c256 = 1.19f, pp_c2 = 0.90f, pp_c3 = 1.06f;
  int pp_prio = 0;              // experiment: the loading wave of a ping-pong slot runs at s_setprio 2
  float pp_dgrad_pen = 1.f;     // factor on the narrow tiles' cost in data-gradient launches (out_len given)
  // lockstep kernel: 0 = the workgroups an XCD holds at one time share a WEIGHT column tile (K > 1: the weights
  // are the big operand), 1 = they are the column tiles of the same ROW tiles (1x1 convolutions: the whole weight
  // matrix sits in every XCD's L2, the activations are what would be fetched once per column tile)
  int tile_order = 0;
};

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The out tile of all NWIN windows is staged through LDS in one pass when it fits next to the
// BatchNorm scratch (the 256 x 256 tiles: 135 KB + 16 KB of the 160 KB), else window by window.
constexpr size_t kEpiSplitBytes = 140 * 1024;
template <int BM, int BN, int NWIN, int NTHR>
constexpr size_t conv_epilogue_lds_bytes() {
  constexpr size_t OP = BN * 2 + 16;
  constexpr int EW = (NWIN * BM * OP > kEpiSplitBytes) ? 1 : NWIN;
  constexpr int RG = (NTHR / (BN / 2)) > 0 ? (NTHR / (BN / 2)) : 1;
  return (size_t)EW * BM * OP + (size_t)EW * RG * BN * 2 * 4;
}

// Group table of the grouped 1x1 launches (conv1d_igemm_grouped_kernel, conv1x1_pp_kernel): every
// group has its own input, weights, output, BatchNorm partials and channel counts; tile_begin =
// first workgroup (igemm) / sum of the previous groups' 256-column tiles (ping-pong kernel).
constexpr int kMaxConvGroups = 16;
struct ConvGroup {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  float* stats;
  int Cin, Cout, accumulate, tile_begin;
  // row strides (elements) when x / y are channel slices of wider tensors; 0 = contiguous (Cin / Cout).
  // Ping-pong kernel only (os2s_conv1x1_cat_fwd); batch stride = rows per sample x row stride
  int x_st = 0, y_st = 0;
};
struct ConvGroupTable {
  int ngroups, total_tiles;
  ConvGroup g[kMaxConvGroups];
};
// gemm_pp.hip: the groups on the 256 x 256 ping-pong tile (OS2S_ERR_UNSUPPORTED outside its envelope)
int launch_conv1x1_pp(hipStream_t stream, ConvArgs a, ConvGroupTable gt);

// Epilogue shared by the tile kernels: fused bias / ReLU / dropout / residual, bf16 pack, LDS
// transpose to full 16-B row stores, per-channel (sum, sum^2) partials for BatchNorm.
template <int BM, int BN, int WM, int WN, int NWIN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p,
                                              f32x16 (&acc)[(BN / WN) / 32][((BM * NWIN) / WM) / 32],
                                              char* smem, int tid, int lane, int wid,
                                              const int (&wmid)[NWIN], int n0,
                                              const int (&wb)[NWIN], const int (&wt0)[NWIN]) {
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int WTM = (BM * NWIN) / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  const int wm = wid / WN, wn = wid % WN;
  // STRADDLE: the rows of a wave cross a window boundary (three windows over four row groups of the
  // 128-column ping-pong tile): window, sample and time are then taken per 32-row fragment
  constexpr bool STRADDLE = (BM % WTM) != 0;
  static_assert(!STRADDLE || (size_t)NWIN * BM * (BN * 2 + 16) <= kEpiSplitBytes,
                "a straddling wave layout needs all windows staged in one pass");
  const int my_win = (wm * WTM) / BM;
  const int row_in_win = (wm * WTM) % BM;
  const int my_b = (NWIN == 1 || my_win == 0) ? wb[0] : wb[NWIN - 1];
  const int my_t0 = (NWIN == 1 || my_win == 0) ? wt0[0] : wt0[NWIN - 1];
  // wmid[w] = window id (row of the BN partial sums) or -1 for a window slot with no work
  const int my_mid = (NWIN == 1 || my_win == 0) ? wmid[0] : wmid[NWIN - 1];
  const int l31 = lane & 31, lhi = lane >> 5;
  if (p.out_f32) {
    // small/rare path (FC logits): scattered fp32 stores straight from registers
    // (not reachable with a straddling layout: the ping-pong launchers refuse out_f32)
    const int b = my_b, t0 = my_t0;
    const int valid_rows = (my_mid >= 0) ? min(BM, p.Tout - t0) : 0;
    float* const yb = reinterpret_cast<float*>(p.y) + (long long)b * p.y_sb;
#pragma unroll
    for (int in = 0; in < NI; ++in)
#pragma unroll
      for (int im = 0; im < MI; ++im) {
        const int tt = row_in_win + im * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int cc = n0 + wn * WTN + in * 32 + 8 * (e >> 2) + 4 * lhi + (e & 3);
          if (tt < valid_rows && cc < p.Cout) {
            float v = acc[in][im][e];
            if (p.bias) v += p.bias[cc];
            float* dst = yb + (long long)(t0 + tt) * p.y_st + cc;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    return;
  }

#ifdef OS2S_EPI_STAMPS   // development aid: cycle stamps of the epilogue phases (tools/conv1x1_phases.py)
  unsigned long long* const est = (p.dbg && blockIdx.x < 2048) ? p.dbg + (size_t)blockIdx.x * 24 + 8 : nullptr;
  int esi = 0;
#define OS2S_EPI_STAMP() do { if (est && tid == 0 && esi < 16) est[esi] = __builtin_readcyclecounter(); ++esi; } while (0)
#else
#define OS2S_EPI_STAMP() do {} while (0)
#endif
  constexpr int OP = BN * 2 + 16;  // out-tile pitch in bytes
  // The out tile is staged through LDS (coalesced 16-B row stores + the BN partial sums). Wide
  // tiles do not fit all windows at once: stage EW windows per pass.
  constexpr bool EPI_SPLIT = (size_t)NWIN * BM * OP > kEpiSplitBytes;
  constexpr int EW = EPI_SPLIT ? 1 : NWIN;
  char* const ot = smem;
#pragma unroll
  for (int w0 = 0; w0 < NWIN; w0 += EW) {
  OS2S_EPI_STAMP();
  __syncthreads();                 // staging buffers / previous pass are no longer read
  OS2S_EPI_STAMP();
  // accumulators -> bf16 out tile in LDS. The dropout variant is a separate instantiation behind a
  // UNIFORM branch: written as `if (p.keep_prob < 1.f)` inside the loop the compiler if-converts
  // it and evaluates the 64-bit hash of dropout_bits8 for every 4 values even when nothing is
  // dropped (measured: 26k of the 54k cycles of a 256 x 256 epilogue, ~12 us per staging pass).
  auto stage_tile = [&](auto DROP) {
  unsigned long long seed = p.seed;
  // the hash must not be speculated above the uniform branch either (it is pure arithmetic, LLVM
  // hoists it): tie it to a volatile asm that only executes on the dropout path
  if constexpr (decltype(DROP)::value) asm volatile("" : "+s"(seed));
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int im = 0; im < MI; ++im) {
      const int tt = wm * WTM + im * 32 + l31 - w0 * BM;   // row in this pass's out tile
      const int b = my_b, t0 = my_t0 - (my_win - w0) * BM;   // t0 + tt = time of this row
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = wn * WTN + in * 32 + 8 * g + 4 * lhi;
        float v0 = acc[in][im][4 * g + 0], v1 = acc[in][im][4 * g + 1];
        float v2 = acc[in][im][4 * g + 2], v3 = acc[in][im][4 * g + 3];
        if (p.bias) {
          const int gc = n0 + cc;
          if (gc + 3 < p.Cout) {
            v0 += p.bias[gc]; v1 += p.bias[gc + 1]; v2 += p.bias[gc + 2]; v3 += p.bias[gc + 3];
          }
        }
        if (p.act == 1) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        } else if (p.act == 3) {      // min(relu(x), 20)
          v0 = fminf(fmaxf(v0, 0.f), 20.f); v1 = fminf(fmaxf(v1, 0.f), 20.f);
          v2 = fminf(fmaxf(v2, 0.f), 20.f); v3 = fminf(fmaxf(v3, 0.f), 20.f);
        }
        if constexpr (decltype(DROP)::value) {
          // same (seed, element index / 8) convention as the elementwise kernels
          const long long e0 = ((long long)b * p.Tout + t0 + tt) * p.Cout + n0 + cc;
          const uint32_t bits = dropout_bits8(seed, (unsigned long long)(e0 >> 3), p.keep_prob) >>
                                (uint32_t)(e0 & 7);
          const float ik = 1.f / p.keep_prob;
          v0 = (bits & 1u) ? v0 * ik : 0.f;
          v1 = (bits & 2u) ? v1 * ik : 0.f;
          v2 = (bits & 4u) ? v2 * ik : 0.f;
          v3 = (bits & 8u) ? v3 * ik : 0.f;
        }
        u32x2 pk;
        pk[0] = pack2bf(v0, v1);
        pk[1] = pack2bf(v2, v3);
        *reinterpret_cast<u32x2*>(ot + tt * OP + cc * 2) = pk;
      }
    }
  };
  if constexpr (STRADDLE) {
    // (b, t0) of a row are per fragment here; the launchers keep dropout off these tiles (pp_ok3)
    stage_tile(std::false_type{});
  } else if (my_win >= w0 && my_win < w0 + EW) {
    if (__builtin_amdgcn_readfirstlane(p.keep_prob < 1.f ? 1 : 0)) stage_tile(std::true_type{});
    else stage_tile(std::false_type{});
  }
  OS2S_EPI_STAMP();
  __syncthreads();
  OS2S_EPI_STAMP();

#pragma unroll
  for (int w = w0; w < w0 + EW; ++w) {
  const int b = wb[w], t0 = wt0[w];
  const int valid_rows = (wmid[w] >= 0) ? min(BM, p.Tout - t0) : 0;
  const char* const otw = ot + (w - w0) * BM * OP;
  bf16_t* const yb = reinterpret_cast<bf16_t*>(p.y) + (long long)b * p.y_sb;
  constexpr int NQ = (BM * (BN / 8) + NTHR - 1) / NTHR;     // 16-B chunks per thread
  static_assert((BM * (BN / 8)) % NTHR == 0, "whole number of chunks per thread");
  // Two phases — every read of the thread (LDS tile, residual rows, previous output when
  // accumulating), then the adds and the stores — and a branch-free path for tiles that lie
  // completely inside the output: interleaved, the compiler keeps load -> wait -> store order
  // (the stores may alias the next load), and a bounds test around a store puts an
  // s_waitcnt vmcnt(0) in front of it; both made the data-gradient launches (accumulate = 1)
  // pay one memory round trip per 16 bytes.
  const bool inside = __builtin_amdgcn_readfirstlane((valid_rows == BM && n0 + BN <= p.Cout) ? 1 : 0);
  u32x4 v[NQ], ro[NQ], ao[NQ];
  if (inside) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * NTHR;
      const int row = q / (BN / 8), c8 = q - row * (BN / 8);
      v[i] = *reinterpret_cast<const u32x4*>(otw + row * OP + c8 * 16);
      const long long off = (long long)(t0 + row) * p.y_st + n0 + c8 * 8;
      if (p.residual) ro[i] = *reinterpret_cast<const u32x4*>(p.residual + (long long)b * p.y_sb + off);
      if (p.accumulate) ao[i] = *reinterpret_cast<const u32x4*>(yb + off);
      if (p.mask_ref) ro[i] = *reinterpret_cast<const u32x4*>(p.mask_ref + (long long)b * p.y_sb + off);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * NTHR;
      const int row = q / (BN / 8), c8 = q - row * (BN / 8);
      if (p.residual && !p.mask_ref) {      // (mask_ref excludes residual: checked on the host)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[i][e] = pack2bf(bflo(v[i][e]) + bflo(ro[i][e]), bfhi(v[i][e]) + bfhi(ro[i][e]));
      }
      if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[i][e] = pack2bf(bflo(v[i][e]) + bflo(ao[i][e]), bfhi(v[i][e]) + bfhi(ao[i][e]));
      }
      if (p.mask_ref) {          // activation (+ dropout) backward of the producing layer on the SUM
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[i][e] = pack2bf(bflo(ro[i][e]) > 0.f ? bflo(v[i][e]) * p.mask_scale : 0.f,
                            bfhi(ro[i][e]) > 0.f ? bfhi(v[i][e]) * p.mask_scale : 0.f);
        if (p.stats) *reinterpret_cast<u32x4*>(const_cast<char*>(otw) + row * OP + c8 * 16) = v[i];
      }
      *reinterpret_cast<u32x4*>(yb + (long long)(t0 + row) * p.y_st + n0 + c8 * 8) = v[i];
    }
  } else {
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * NTHR;
      const int row = q / (BN / 8), c8 = q - row * (BN / 8);
      const int gc = n0 + c8 * 8;
      if (row < valid_rows && gc < p.Cout) {
        u32x4 x = *reinterpret_cast<const u32x4*>(otw + row * OP + c8 * 16);
        bf16_t* dst = yb + (long long)(t0 + row) * p.y_st + gc;
        if (p.residual && !p.mask_ref) {
          const u32x4 o = *reinterpret_cast<const u32x4*>(
              p.residual + (long long)b * p.y_sb + (long long)(t0 + row) * p.y_st + gc);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = pack2bf(bflo(x[e]) + bflo(o[e]), bfhi(x[e]) + bfhi(o[e]));
        }
        if (p.accumulate) {
          const u32x4 o = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = pack2bf(bflo(x[e]) + bflo(o[e]), bfhi(x[e]) + bfhi(o[e]));
        }
        if (p.mask_ref) {
          const u32x4 o = *reinterpret_cast<const u32x4*>(
              p.mask_ref + (long long)b * p.y_sb + (long long)(t0 + row) * p.y_st + gc);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            x[e] = pack2bf(bflo(o[e]) > 0.f ? bflo(x[e]) * p.mask_scale : 0.f,
                           bfhi(o[e]) > 0.f ? bfhi(x[e]) * p.mask_scale : 0.f);
          if (p.stats) *reinterpret_cast<u32x4*>(const_cast<char*>(otw) + row * OP + c8 * 16) = x;
        }
        *reinterpret_cast<u32x4*>(dst) = x;
      }
    }
  }
  }

  OS2S_EPI_STAMP();
  if (p.stats && p.mask_ref) __syncthreads();   // the masked values were written back to the LDS tile
  if (p.stats) {
    constexpr int CP = BN / 2;       // column pairs
    constexpr int RG = (NTHR / CP) > 0 ? (NTHR / CP) : 1;    // row groups
    constexpr int RPT = (BM + RG - 1) / RG;                  // rows per thread
    const int cp = tid % CP, rg = tid / CP;
    // every window of the pass has its own scratch [RG][BN][2]: one barrier for all of them
#pragma unroll
    for (int w = w0; w < w0 + EW; ++w) {
      if (wmid[w] < 0) continue;
      const int valid_rows = min(BM, p.Tout - wt0[w]);
      const char* const otw = ot + (w - w0) * BM * OP;
      float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
      if (rg < RG) {
        // constant trip count + predicate: the LDS reads issue back to back (a `row < valid_rows`
        // loop bound made every read wait for the previous one)
        uint32_t vv[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int row = rg + i * RG;
          vv[i] = row < valid_rows ? *reinterpret_cast<const uint32_t*>(otw + row * OP + cp * 4) : 0u;
        }
        if (p.stat_ref) {
          // second statistic = sum of (value x reference): the reference pair of this thread's columns,
          // 4 bytes per row (a wave reads 256 contiguous bytes of a row)
          const bool cok = n0 + cp * 2 < p.Cout;
          const bf16_t* const rp = p.stat_ref + (long long)wb[w] * p.y_sb + (long long)wt0[w] * p.y_st + n0 + cp * 2;
          uint32_t rr[RPT];
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const int row = rg + i * RG;
            rr[i] = (cok && row < valid_rows) ? *reinterpret_cast<const uint32_t*>(rp + (long long)row * p.y_st) : 0u;
          }
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const float a = bflo(vv[i]), bb = bfhi(vv[i]);
            s0 += a; q0 += a * bflo(rr[i]);
            s1 += bb; q1 += bb * bfhi(rr[i]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const float a = bflo(vv[i]), bb = bfhi(vv[i]);
            s0 += a; q0 += a * a;
            s1 += bb; q1 += bb * bb;
          }
        }
        float* sc = reinterpret_cast<float*>(smem + EW * BM * OP) + (size_t)(w - w0) * RG * BN * 2;
        sc[(rg * BN + cp * 2 + 0) * 2 + 0] = s0;
        sc[(rg * BN + cp * 2 + 0) * 2 + 1] = q0;
        sc[(rg * BN + cp * 2 + 1) * 2 + 0] = s1;
        sc[(rg * BN + cp * 2 + 1) * 2 + 1] = q1;
      }
    }
    OS2S_EPI_STAMP();
    __syncthreads();
    OS2S_EPI_STAMP();
#pragma unroll
    for (int w = w0; w < w0 + EW; ++w) {
      if (wmid[w] < 0) continue;
      const float* sc = reinterpret_cast<const float*>(smem + EW * BM * OP) + (size_t)(w - w0) * RG * BN * 2;
      for (int c = tid; c < BN; c += NTHR) {
        float s = 0.f, qq = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          s += sc[(g * BN + c) * 2 + 0];
          qq += sc[(g * BN + c) * 2 + 1];
        }
        const int gc = n0 + c;
        if (gc < p.Cout) {
          p.stats[((long long)wmid[w] * 2 + 0) * p.Cout + gc] = s;
          p.stats[((long long)wmid[w] * 2 + 1) * p.Cout + gc] = qq;
        }
      }
    }
  }
  }
}

}  // namespace os2s
