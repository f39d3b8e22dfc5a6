// Implicit-GEMM 1-D convolution (channels-last, bf16 in, fp32 accumulate) on
// the gfx950 matrix cores. One kernel serves:
//   * forward conv1d of the TDNN/Jasper blocks   (tf.layers.conv1d, no bias —
//     open_seq2seq/parts/cnns/conv_blocks.py:195-206), including the fused
//     sequence mask on the conv INPUT (tdnn_encoder.py:185-186,204-205);
//   * 1x1 residual projections (conv_blocks.py:78-85) and plain GEMMs (K = 1);
//   * data-gradient (dgrad) of a stride-1 conv: the same kernel run on dY with
//     the tap-flipped / transposed weight copy and padL' = (K-1)*dil - padL.
//
//   y[b,t,co] = sum_k sum_ci x[b, t*stride + k*dil - padL, ci] * w[k][co][ci]
//
// Design (MI355X-first):
//   * tile = BM time steps of ONE batch item x BN output channels; the input
//     time WINDOW (BM-1)*stride + (K-1)*dil + 1 rows x 64 channels is staged in
//     LDS once per 64-channel chunk and re-used by all K taps (tap k just reads
//     rows shifted by k*dil) — activations are fetched from L2/HBM once per tile,
//     not K times; only the [BN x 64] weight tile streams per tap.
//   * all global->LDS traffic is LDS-DMA (global_load_lds, 16 B/lane); rows out
//     of range / past the sequence length are sourced from a zero page, so
//     padding and the sequence mask cost nothing.
//   * LDS images are 128-B rows with a 16-B-slot XOR swizzle (slot ^= (row>>1)&7)
//     applied on the DMA *source* address and on the ds_read_b128 side, which is
//     conflict-free for the 16-lane service groups of ds_read_b128.
//   * MFMA 32x32x16 bf16, operands swapped (A = weights, B = activations) so each
//     lane ends up with 4 consecutive output channels of one time row; the
//     epilogue goes through LDS to emit full 16-B/lane coalesced row stores and
//     the per-channel (sum, sum^2) partials BatchNorm needs — the conv output
//     is not re-read to get batch statistics.
//   * double-buffered LDS; the DMA of step s+1 is in flight during the MFMAs of
//     step s; one barrier per step; 2 workgroups per CU.
//   * XCD-aware block order: the blocks resident on one XCD at a time share the
//     same weight n-tile, so the weight stream is an L2 hit for all but one.
#include "os2s_common.hpp"
#include "conv1d_common.hpp"
#include "os2s_split_reduce.hpp"
#include <algorithm>
#include <string>
#include <array>
#include <type_traits>
#include <map>
#include <mutex>
#include <vector>

namespace os2s {

// BM = rows of ONE time window (one batch item); a workgroup processes NWIN consecutive
// windows (possibly of different batch items) against the same weight stream, so the
// M extent of a block is NWIN*BM while the tile granularity in time stays BM.
template <int BM, int BN, int WM, int WN, int NWIN, bool XSINGLE>
__device__ __forceinline__ void conv_tile_body(const ConvArgs& p, const int bid) {
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int WTM = (BM * NWIN) / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be 32-aligned");
  static_assert(BM % WTM == 0, "a wave must not straddle two windows");
  static_assert((BN * 8) % (NW * 64) == 0, "weight tile DMA split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int my_win = (wm * WTM) / BM;            // window this wave computes
  const int row_in_win = (wm * WTM) % BM;

  // ---- block -> tiles (XCD-aware: see header) ------------------------------
  const int xcd = bid & 7, loc = bid >> 3;
  const int n_idx = p.tile_order ? loc % p.NT : loc / p.MT8;
  const int m_first = ((p.tile_order ? loc / p.NT : loc - n_idx * p.MT8) * 8 + xcd) * NWIN;
  if (m_first >= p.MT) return;
  const int n0 = n_idx * BN;
  int wb[NWIN], wt0[NWIN], wlen[NWIN], wwin[NWIN];
#pragma unroll
  for (int w = 0; w < NWIN; ++w) {
    const int m = m_first + w;
    const bool live = m < p.MT;
    wb[w] = live ? m / p.mtiles_per_b : 0;
    wt0[w] = live ? (m - wb[w] * p.mtiles_per_b) * BM : 0;
    int len_b = p.Tin;
    if (p.in_len) {
      int l = p.in_len[wb[w]];
      len_b = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
    }
    wlen[w] = live ? len_b : 0;                   // dead window: every row reads as zero
    wwin[w] = wt0[w] * p.stride - p.padL;
  }
  // Exact-zero shortcuts (ragged batches are padded to the longest utterance):
  //  * every row of the input window lies past in_len -> the accumulators stay zero, the
  //    MFMA loop is skipped (the epilogue still stores the zero tile + zero BN partials);
  //  * the whole output tile lies past out_len -> nobody reads it, nothing is done at all.
  bool dead_in = true;
#pragma unroll
  for (int w = 0; w < NWIN; ++w) dead_in = dead_in && (wwin[w] >= wlen[w] || m_first + w >= p.MT);
  if (p.out_len) {
    bool dead_out = true;
#pragma unroll
    for (int w = 0; w < NWIN; ++w)
      dead_out = dead_out && (m_first + w >= p.MT || wt0[w] >= p.out_len[wb[w]]);
    if (dead_out) return;
  }
  static_assert(NWIN == 1 || NWIN == 2, "window selects below are written for <= 2 windows");
  // select by compare (a runtime-indexed register array would be demoted to scratch)
  const int my_b = (NWIN == 1 || my_win == 0) ? wb[0] : wb[NWIN - 1];
  const int my_t0 = (NWIN == 1 || my_win == 0) ? wt0[0] : wt0[NWIN - 1];
  const int win_bytes = p.Rpad * 128;
  const int xbuf_bytes = NWIN * win_bytes;
  char* const xbuf0 = smem;
  char* const wbuf0 = smem + (XSINGLE ? 1 : 2) * xbuf_bytes;
  const char* const zero = reinterpret_cast<const char*>(g_zero_page);

  auto stage_x = [&](int c, char* dst) {
    const int npieces = p.Rpad * 8;  // multiple of 64
#pragma unroll
    for (int w = 0; w < NWIN; ++w) {
      const bf16_t* const xb = p.x + (long long)wb[w] * p.x_sb;
      for (int base = wid * 64; base < npieces; base += NW * 64) {
        const int q = base + lane;
        const int r = q >> 3, jj = q & 7;
        const int j = jj ^ ((r >> 1) & 7);
        const int tin = wwin[w] + r;
        const int ch = c * 64 + j * 8;
        const bool ok = (r < p.R) && (tin >= 0) && (tin < wlen[w]) && (ch < p.Cin);
        const void* src = ok ? (const void*)(xb + (long long)tin * p.x_st + ch)
                             : (const void*)(zero + jj * 16);
        dma16(src, dst + w * win_bytes + base * 16);
      }
    }
  };
  auto stage_w = [&](int c, int k, char* dst) {
#pragma unroll
    for (int it = 0; it < (BN * 8) / (NW * 64); ++it) {
      const int base = (it * NW + wid) * 64;
      const int q = base + lane;
      const int row = q >> 3, jj = q & 7;
      const int j = jj ^ ((row >> 1) & 7);
      int n = n0 + row;
      n = n < p.Cout ? n : p.Cout - 1;
      const int ch = c * 64 + j * 8;
      const void* src =
          (ch < p.Cin)
              ? (const void*)(p.w + ((long long)k * p.Cout + n) * p.Cin + ch)
              : (const void*)(zero + jj * 16);
      dma16(src, dst + base * 16);
    }
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int im = 0; im < MI; ++im)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[in][im][e] = 0.f;

  const int nsteps = dead_in ? 0 : p.nchunks * p.K;
  if (!dead_in) {
    stage_x(0, xbuf0);
    stage_w(0, 0, wbuf0);
  }
  int c = 0, k = 0;
  const int l31 = lane & 31, lhi = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    if (XSINGLE && k == 0 && step > 0) {
      // single X buffer: everyone is done with the previous chunk's window -> refill it
      __syncthreads();
      stage_x(c, xbuf0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int kn = k + 1, cn = c;
    if (kn == p.K) { kn = 0; cn = c + 1; }
    if (step + 1 < nsteps) {
      if (!XSINGLE && kn == 0) stage_x(cn, xbuf0 + (cn & 1) * xbuf_bytes);
      stage_w(cn, kn, wbuf0 + ((step + 1) & 1) * (BN * 128));
    }
    const char* const xs = xbuf0 + (XSINGLE ? 0 : (c & 1) * xbuf_bytes) + my_win * win_bytes;
    const char* const ws = wbuf0 + (step & 1) * (BN * 128);
    const int rbase = (row_in_win + l31) * p.stride + k * p.dil;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int j = kk * 2 + lhi;
      bf16x8 xa[MI], wb[NI];
#pragma unroll
      for (int im = 0; im < MI; ++im) {
        const int r = rbase + im * 32 * p.stride;
        xa[im] = *reinterpret_cast<const bf16x8*>(xs + r * 128 + ((j ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int in = 0; in < NI; ++in) {
        const int n = wn * WTN + in * 32 + l31;
        wb[in] = *reinterpret_cast<const bf16x8*>(ws + n * 128 + ((j ^ ((n >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int in = 0; in < NI; ++in)
#pragma unroll
        for (int im = 0; im < MI; ++im)
          acc[in][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[in], xa[im], acc[in][im], 0, 0, 0);
    }
    c = cn;
    k = kn;
  }

  int wmid[NWIN];
#pragma unroll
  for (int w = 0; w < NWIN; ++w) wmid[w] = (m_first + w < p.MT) ? m_first + w : -1;
  conv_epilogue<BM, BN, WM, WN, NWIN>(p, acc, smem, tid, lane, wid, wmid, n0, wb, wt0);
}

template <int BM, int BN, int WM, int WN, int NWIN, bool XSINGLE = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN <= 4) ? (XSINGLE ? 3 : 2) : 1) void conv1d_igemm_kernel(
    ConvArgs p) {
  conv_tile_body<BM, BN, WM, WN, NWIN, XSINGLE>(p, blockIdx.x);
}

// Grouped launch: up to kMaxConvGroups independent 1x1 convolutions (same batch geometry and
// lengths; own input, weights, output, BatchNorm partials, channel counts) in ONE grid — the
// dense-residual 1x1 branches of a Jasper block end (conv_blocks.py:78-85: up to 10 per block,
// 55 per pass) and their data gradients. Each is a few microseconds of matrix work; launched
// one by one they cost ~45 us apiece.

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv1d_igemm_grouped_kernel(ConvArgs p, ConvGroupTable gt) {
  const int bid = blockIdx.x;
  if (bid >= gt.total_tiles) return;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < kMaxConvGroups; ++i)
    if (i < gt.ngroups && bid >= gt.g[i].tile_begin) gi = i;
  // select the group's fields without indexing the table by a runtime value in registers
  ConvGroup g = gt.g[0];
#pragma unroll
  for (int i = 1; i < kMaxConvGroups; ++i)
    if (i == gi) g = gt.g[i];
  p.x = g.x; p.w = g.w; p.y = g.y; p.stats = g.stats;
  p.Cin = g.Cin; p.Cout = g.Cout; p.accumulate = g.accumulate;
  p.x_sb = (long long)p.Tin * g.Cin; p.x_st = g.Cin;
  p.y_sb = (long long)p.Tout * g.Cout; p.y_st = g.Cout;
  p.NT = (g.Cout + BN - 1) / BN;
  p.nchunks = (g.Cin + 63) / 64;
  conv_tile_body<BM, BN, WM, WN, 1, false>(p, bid - g.tile_begin);
}


// ---------------------------------------------------------------------------------------------
// Ping-pong kernel for the stride-1 layers with K >= 5 (the bulk of the Jasper FLOPs).
//
// Tile: two 128-row time windows x BN = 256 output channels, 8 waves. Waves 0-3 ("group A")
// compute window 0, waves 4-7 ("group B") window 1; wave w and wave w+4 sit on the same SIMD.
// Both groups stream the SAME weight tile, so a workgroup reads the weights once per 256 rows.
//
// The loop is cut into barrier-delimited slots in which one wave of every SIMD issues MFMAs
// (COMPUTE) while its partner fetches the fragments of its next item from LDS (LOAD) — the
// matrix pipe never waits for ds_read latency, address arithmetic or a barrier skew of the
// wave that feeds it. An item is (64-channel step, 64-row half of the wave's 128 x 64 tile):
// 16 MFMA 32x32x16 = 512 matrix-pipe cycles per slot.
//
//   slot        4s         4s+1        4s+2        4s+3
//   group A   LOAD(2s)   COMP(2s)   LOAD(2s+1)  COMP(2s+1)
//   group B   COMP(2s-1) LOAD(2s)   COMP(2s)    LOAD(2s+1)
//
// LDS: X windows double-buffered per 64-channel chunk (re-used by all K taps), weight tile
// [256 x 64] double-buffered per step. The weight fragments of a step are read once (LOAD of
// the even item) and kept in registers, so a weight buffer is busy in 2 slots out of 8. ALL DMA
// is issued by the wave that is in its odd LOAD slot (measured: a DMA instruction issued between
// MFMAs costs the matrix pipe ~45 cycles): LOAD(2s+1) first drains what LOAD(2s-1) issued
// (vmcnt(0) after a whole step of flight time — nothing to count), then issues the weight tile
// of step s+2 and one instruction of the X window of chunk c+1 (during the first steps of chunk
// c; needs K > number of such instructions per wave).
//
// Work distribution (what the kernel is really about — the loop above runs at ~1.4 PFLOP/s per
// busy CU, an unbalanced launch wastes half of it):
//   * live windows only. Every wave derives the live-window list of the ragged batch from
//     in_len / out_len with one wave scan (B <= 64): window pairs are formed over the
//     COMPACTED list, so no workgroup is half padding and no workgroup is dispatched for
//     padding at all. Dead windows of a forward call get their zero rows / zero BatchNorm
//     partials from a few store-only workgroups at the end of the grid.
//   * unit = (window pair, n-tile), enumerated so that consecutive workgroups (= the 8 XCDs
//     round-robin) hold DIFFERENT pairs and the n-tiles of one pair follow each other on the
//     same XCD: the X windows are fetched into one L2 only, every weight n-tile stream is
//     shared by the workgroups of an XCD that run in lockstep.
//   * U units on G CUs: the first floor(U/G)*G units run whole; the remaining r = U mod G units
//     are split f ways over the input-channel chunks (f chosen per launch, on the device, from
//     a fixed cost model) so that the tail costs ~1/f of a round instead of a full one. The
//     pieces of a unit leave fp32 partial tiles in a workspace; the piece that arrives last
//     (agent-scope release / acquire around one atomic ticket) sums them in piece order —
//     deterministic — and runs the epilogue. No spinning, so no co-residency assumption.
// ---------------------------------------------------------------------------------------------
constexpr int kPpBN = 256;
constexpr int kPpZeroWin = 2;                      // dead windows per store-only workgroup
#ifndef OS2S_PP_PRIO
#define OS2S_PP_PRIO 0                             // experiment: s_setprio(1) around the MFMA runs
#endif

__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// Store-only role of the ping-pong launches (forward calls): workgroup z of the tail of the grid writes
// the zero rows (or the residual rows) and the zero BatchNorm partials of kPpZeroWin dead windows.
__device__ __forceinline__ void pp_zero_role(const ConvArgs& p, const int nw, const int z, const int tid) {
  constexpr int BM = 128;
  if (p.out_len || p.out_f32 || z >= (p.MT + kPpZeroWin - 1) / kPpZeroWin) return;
  for (int m = z * kPpZeroWin; m < (z + 1) * kPpZeroWin && m < p.MT; ++m) {
    const int b = m / p.mtiles_per_b, j = m - b * p.mtiles_per_b;
    if (j < __shfl(nw, b, 64)) continue;
    const int t0 = j * BM, rows = min(BM, p.Tout - t0);
    if (!p.accumulate) {
      bf16_t* const yb = reinterpret_cast<bf16_t*>(p.y) + (long long)b * p.y_sb;
      const int c8n = p.Cout >> 3;
      const u32x4 zv = {0u, 0u, 0u, 0u};
      for (int e = tid; e < rows * c8n; e += 512) {
        const int row = e / c8n, c8 = e - row * c8n;
        u32x4 v = zv;
        if (p.residual)
          v = *reinterpret_cast<const u32x4*>(p.residual + (long long)b * p.y_sb +
                                              (long long)(t0 + row) * p.y_st + c8 * 8);
        *reinterpret_cast<u32x4*>(yb + (long long)(t0 + row) * p.y_st + c8 * 8) = v;
      }
    }
    if (p.stats)
      for (int e = tid; e < 2 * p.Cout; e += 512) p.stats[(long long)m * 2 * p.Cout + e] = 0.f;
  }
}

// Tail-split factor of the 256-column tile (same decision in every workgroup: pure function of the launch)
__host__ __device__ __forceinline__ int pp256_split(const ConvArgs& p, const int U, const int S) {
  const int G = p.ncu;
  const int q = U / G, r = U - q * G;
  int f = 1;
  if (p.force_split > 0 && p.ws_slabs) {
    f = p.force_split;
    while (f > 1 && (f > p.nchunks || r * f > p.ws_nslabs)) --f;
    if (r == 0) f = 1;
  } else if (r > 0 && p.ws_slabs) {
    // a round of whole units takes S steps x 1.18 us (tools/bench_conv_split.py)
    f = split_factor(r, G, p.pp_c256 * S, p.nchunks < 8 ? p.nchunks : 8, p.ws_nslabs);
  }
  return f;
}

// Tile shape of a ping-pong launch, chosen on the device (the live-window count L of a ragged batch is
// only known there) from a fixed cost model — every workgroup evaluates the same pure function of the
// launch arguments and in_len / out_len. Candidates: 0 = two windows x 256 columns (tail of the
// launch split along the input channels), 2 / 3 = two / three windows x 128 columns (ppn_body).
// The model is rounds x steps x microseconds per step of the tile (+ the fitted tail-split terms):
// what decides is how many equal units a layer gives — Jasper's 384 / 512 / 640-channel layers on the
// bench batch are 146 / 146 / 219 units of 256 columns on 256 CUs, but 219 / 196 / 245 narrow ones.
__host__ __device__ __forceinline__ int pp_choose_tile(const ConvArgs& p, const int L) {
  if (p.pp_tile >= 0) return p.pp_tile;
  const int G = p.ncu, S = p.nchunks * p.K;
  float best;
  {
    const int U = ((L + 1) >> 1) * p.NT;
    const int q = U / G, r = U - q * G;
    const float round_us = p.pp_c256 * S + 12.f;
    best = q * round_us;
    if (r > 0) {
      const int f = pp256_split(p, U, S);
      best += f > 1 ? (float)((r * f + G - 1) / G) * round_us / f + 20.f + 1.5f * f + 0.17f * (r * f) : round_us;
    }
  }
  int tile = 0;
  const int NT128 = (p.Cout + 127) >> 7;
  // data-gradient launches share the chip with the weight-gradient stream: what counts there is CU time,
  // not the length of the launch alone — the narrow tiles' cost carries a penalty factor
  const float pen = p.out_len ? p.pp_dgrad_pen : 1.f;
  if (p.pp_ok2) {
    const int U = ((L + 1) >> 1) * NT128;
    const float t = pen * (float)((U + G - 1) / G) * (p.pp_c2 * S + 9.f);
    if (t < 0.97f * best) { best = t; tile = 2; }
  }
  if (p.pp_ok3) {
    const int U = ((L + 2) / 3) * NT128;
    const float t = pen * (float)((U + G - 1) / G) * (p.pp_c3 * S + 10.f);
    if (t < 0.97f * best) { best = t; tile = 3; }
  }
  return tile;
}

template <bool DBG>
__device__ __forceinline__ void pp256_body(const ConvArgs& p, const int L, const int scan, const int nw, char* smem);
template <int NWIN, bool DBG>
__device__ __forceinline__ void ppn_body(const ConvArgs& p, const int L, const int scan, const int nw, char* smem);

// live windows per sample (one value per lane), inclusive scan over the batch in `scan`
__device__ __forceinline__ int pp_live_windows(const ConvArgs& p, const int lane, int& scan) {
  constexpr int BM = 128;
  int nw = 0;
  if (lane < p.B) {
    if (p.out_len) {
      // data-gradient call: rows >= out_len are never read by the caller
      int lo = p.out_len[lane];
      lo = lo < 0 ? 0 : (lo < p.Tout ? lo : p.Tout);
      nw = (lo + BM - 1) / BM;
    } else {
      int len = p.Tin;
      if (p.in_len) {
        const int l = p.in_len[lane];
        len = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
      }
      nw = len > 0 ? (len + p.padL + BM - 1) / BM : 0;   // windows with a real input row
    }
    nw = nw < p.mtiles_per_b ? nw : p.mtiles_per_b;
  }
  scan = nw;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(scan, o, 64);
    if (lane >= o) scan += t;
  }
  return nw;
}

// The two ping-pong kernels of a launch whose tile is chosen on the device (pp_tile < 0) are enqueued
// back to back; every workgroup of both evaluates pp_choose_tile and the kernel that was not chosen
// exits at once (one global load + a wave scan per workgroup, but a grid of several hundred of them:
// 7.5 - 8.7 us per null launch in profiles/r05_jasper_kernel_stats.csv, 0.35 ms of a Jasper step — which
// is why a caller that holds the sequence lengths on the host hands them over, os2s_conv1d_set_host_lens:
// the same function is then evaluated at launch time and only the chosen kernel is enqueued). They are
// separate kernels, not
// one kernel with three bodies: fused, the register allocator re-read kernel arguments inside the
// main loops (s_load + s_waitcnt lgkmcnt(0) in every LOAD slot of the 256-column body).
// DBG: per-slot s_memtime stamps of waves 0 and 4 of workgroups 0..3 (tools/pp_timeline.py)
template <bool DBG>
__global__ __launch_bounds__(512, 2) void conv1d_pp_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int scan;
  const int nw = pp_live_windows(p, threadIdx.x & 63, scan);
  const int L = __builtin_amdgcn_readlane(scan, 63);
  if (__builtin_amdgcn_readfirstlane(pp_choose_tile(p, L)) != 0) return;
  pp256_body<DBG>(p, L, scan, nw, smem);
}

template <bool DBG>
__global__ __launch_bounds__(512, 2) void conv1d_ppn_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int scan;
  const int nw = pp_live_windows(p, threadIdx.x & 63, scan);
  const int L = __builtin_amdgcn_readlane(scan, 63);
  const int tile = __builtin_amdgcn_readfirstlane(pp_choose_tile(p, L));
  if (tile == 2) ppn_body<2, DBG>(p, L, scan, nw, smem);
  else if (tile == 3) ppn_body<3, DBG>(p, L, scan, nw, smem);
}

template <bool DBG>
__device__ __forceinline__ void pp256_body(const ConvArgs& p, const int L, const int scan, const int nw, char* smem) {
  constexpr int BM = 128, BN = kPpBN, NWIN = 2, WM = 2, WN = 4;
  constexpr int MI = 4, NI = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;
  const int P = (L + 1) >> 1, U = P * p.NT, G = p.ncu;
  const int S = p.nchunks * p.K;
  const int r = U - (U / G) * G;
  const int f = pp256_split(p, U, S);
  const int nfull = f > 1 ? U - r : U;
  const int nwork = nfull + (f > 1 ? r * f : 0);
  const int bid = blockIdx.x;

  if (bid >= nwork) {
    pp_zero_role(p, nw, (int)gridDim.x - 1 - bid, tid);
    return;
  }

  // ---- work role: (unit rank, piece) -> (window pair, n-tile, chunk range) --------------------
  int rank = bid, piece = 0, npiece = 1;
  if (bid >= nfull) {
    const int i = bid - nfull;
    rank = nfull + i / f;
    piece = i - (i / f) * f;
    npiece = f;
  }
  int xcd, loc;
  {
    const int P8 = (P + 7) >> 3, rem = P - 8 * (P8 - 1), base = (P8 - 1) * p.NT * 8;
    if (rank < base) { xcd = rank & 7; loc = rank >> 3; }
    else { const int i = rank - base; loc = (P8 - 1) * p.NT + i / rem; xcd = i - (i / rem) * rem; }
  }
  const int pair = (loc / p.NT) * 8 + xcd;
  const int n_idx = loc - (loc / p.NT) * p.NT;
  const int n0 = n_idx * BN;
  int wb[NWIN], wt0[NWIN], wlen[NWIN], wwin[NWIN], wmid[NWIN];
#pragma unroll
  for (int w = 0; w < NWIN; ++w) {
    const int i = 2 * pair + w;
    const bool live = i < L;
    const int b = live ? __builtin_popcountll(__ballot(scan <= i)) : 0;
    const int before = b > 0 ? __shfl(scan, b - 1, 64) : 0;
    wb[w] = b;
    wt0[w] = live ? (i - before) * BM : 0;
    int len_b = p.Tin;
    if (p.in_len) {
      const int l = p.in_len[b];
      len_b = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
    }
    wlen[w] = live ? len_b : 0;                     // dead slot: every row reads as zero
    wwin[w] = wt0[w] - p.padL;                      // stride 1
    wmid[w] = live ? b * p.mtiles_per_b + (i - before) : -1;
  }
  const int c_beg = __builtin_amdgcn_readfirstlane(piece * p.nchunks / npiece);
  const int c_end = __builtin_amdgcn_readfirstlane((piece + 1) * p.nchunks / npiece);

  const int win_bytes = p.Rpad * 128;
  const int xbuf_bytes = NWIN * win_bytes;
  char* const xbuf0 = smem;
  char* const wbuf0 = smem + 2 * xbuf_bytes;
  char* const dump = wbuf0 + 2 * BN * 128;          // 1 KB landing zone of padding DMAs
  const int rb_per_win = p.Rpad >> 3;               // DMA instructions (8 rows each) per window
  const int nxi = (rb_per_win + 3) >> 2;            // X DMA instructions per wave per chunk

  // All global->LDS traffic goes through buffer descriptors (LDS-DMA, 16 B per lane): the
  // per-lane byte offset is fixed for the whole kernel, the chunk / tap position sits in the
  // scalar offset, and the hardware range check supplies the zeros — rows before the sequence
  // start (negative offset = huge unsigned), rows past in_len (num_records = len rows) and the
  // dead window slot (num_records = 0) — so a DMA costs two scalar and one vector instruction.
  // A wave stages rows of its OWN group's window only: one descriptor, no selects in the loop.
  __amdgpu_buffer_rsrc_t xrs;
  {
    const int b_own = grp ? wb[1] : wb[0];
    const int len_own = grp ? wlen[1] : wlen[0];
    const bf16_t* xb = p.x + (long long)b_own * p.x_sb;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)xb);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)xb >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(len_own * (int)p.x_st * 2);
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes,
                                            0x00020000);
  }
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w, 0, (int)((long long)p.K * p.Cout * p.Cin * 2), 0x00020000);
  // one DMA instruction = 8 rows x 128 B of an X window. Instruction n (n < nxi) of a chunk
  // issued by wave (grp, wq) covers row group rb = wq + 4 n of the group's window; every wave
  // issues exactly nxi per chunk (the surplus ones read out of range and land in `dump`).
  // per-lane byte offset inside the 8-row group: row (lane >> 3), 16-B slot jj ^ swizzle(row);
  // the swizzle term (row >> 1) & 7 of row rb*8 + (lane >> 3) flips bit 2 for odd rb
  const int xlane = (lane >> 3) * (int)p.x_st * 2 + (((lane & 7) ^ (lane >> 4)) << 4);
  const int xrow0 = __builtin_amdgcn_readfirstlane((grp ? wwin[1] : wwin[0]) * (int)p.x_st * 2);
  const int xgrp_bytes = __builtin_amdgcn_readfirstlane(8 * (int)p.x_st * 2);
  auto stage_x = [&](int c, int n) {
    const int rb = (wid & 3) + 4 * n;
    const bool real = rb < rb_per_win;
    char* dst = real ? xbuf0 + (c & 1) * xbuf_bytes + grp * win_bytes + rb * 1024 : dump;
    // rows before the sequence start give a negative (= huge unsigned) offset -> zeros
    const int srow = real ? xrow0 + rb * xgrp_bytes : (int)0x80000000;
    const int vo = (xlane ^ ((rb & 1) << 6)) + srow;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16,
                                             vo, __builtin_amdgcn_readfirstlane(c * 128), 0, 0);
  };
  // weight tile rows handled by this lane (4 DMA instructions of 8 rows per wave and step)
  int wsrc[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    const int row = (pi * 8 + wid) * 8 + (lane >> 3), jj = lane & 7;
    const int j = jj ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;
    wsrc[pi] = (n * p.Cin + j * 8) * 2;
  }
  const int w_kstride = __builtin_amdgcn_readfirstlane(p.Cout * p.Cin * 2);
  auto stage_w = [&](int c, int k, int par) {
    const int soff = __builtin_amdgcn_readfirstlane((DBG && p.dbg_fixed_w) ? 0 : k * w_kstride + c * 128);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(wbuf0 + par * (BN * 128) + (pi * 8 + wid) * 1024),
          16, wsrc[pi], soff, 0, 0);
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int im = 0; im < MI; ++im)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[in][im][e] = 0.f;

  const int nsteps = (c_end - c_beg) * p.K;
  {
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int n = 0; n < nxi; ++n) stage_x(c_beg, n);
    stage_w(c_beg, 0, 0);
    if (nsteps > 1) stage_w(p.K > 1 ? c_beg : c_beg + 1, p.K > 1 ? 1 : 0, 1);
    // fragment offsets of the weight tile: row = wn*64 + in*32 + l31, fixed for the kernel
    int woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      woff[kk] = (wn * 64 + l31) * 128 + (((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp) pp_barrier();                          // group B runs one slot behind group A

    int c = c_beg, k = 0;   // step s
    int cw = c_beg, kw = 0; // step s + 2
    for (int i = 0; i < 2; ++i) { if (++kw == p.K) { kw = 0; ++cw; } }

    unsigned long long* const tl = reinterpret_cast<unsigned long long*>(dump + 1024);
    const bool rec = DBG && p.dbg && bid < 4 && lane == 0 && (wid & 3) == 0;
    auto stamp = [&](int s, int i) {
      if (DBG && rec && s < 48) tl[(grp * 48 + s) * 9 + i] = __builtin_readcyclecounter();
    };
    // LDS byte addresses of this lane's X fragments (k-step kk, rows l31 + 32 i of the window)
    // for the CURRENT step; refreshed for the next step in the odd LOAD slot, which has slack
    const char* xad[4];
    auto set_xad = [&](int cc, int kk_tap) {
      const int r0 = l31 + kk_tap * p.dil;
      const int m = (r0 >> 1) & 7;
      const char* const xrow = xbuf0 + (cc & 1) * xbuf_bytes + grp * win_bytes + r0 * 128;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xad[kk] = xrow + (((kk * 2 + lhi) ^ m) << 4);
    };
    set_xad(c, k);
    auto step = [&](auto PAR, int s) {
      constexpr int PB = decltype(PAR)::value;
      const char* const ws = wbuf0 + PB * (BN * 128);
      bf16x8 wf[NI][4], xf[2][4];
      // ---- LOAD(2s): weight fragments of the step + X fragments of rows 0..63 -----------------
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int in = 0; in < NI; ++in)
          wf[in][kk] = *reinterpret_cast<const bf16x8*>(ws + woff[kk] + in * 4096);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          xf[i2][kk] = *reinterpret_cast<const bf16x8*>(xad[kk] + i2 * 4096);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(s, 0);
      pp_barrier();
      stamp(s, 1);
      // ---- COMPUTE(2s): nothing but MFMAs — any other instruction of this wave (a DMA issue
      //      costs ~45 cycles) would come straight out of the matrix pipe's time
      if (OS2S_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < NI; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            acc[in][i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk],
                                                                  acc[in][i2], 0, 0, 0);
      if (OS2S_PP_PRIO) __builtin_amdgcn_s_setprio(0);
      stamp(s, 2);
      pp_barrier();
      stamp(s, 3);
      // ---- LOAD(2s+1): X fragments of rows 64..127, then the DMA traffic of the step: drain what
      //      the previous odd slot issued (a whole step of flight time), issue the weight tile of
      //      step s+2 (its buffer was last read in LOAD(2s)) and one X instruction of the next
      //      chunk, and prepare the fragment addresses of step s+1
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          xf[i2][kk] = *reinterpret_cast<const bf16x8*>(xad[kk] + (2 + i2) * 4096);
      if (DBG) stamp(s, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(s, 5);
      if (s + 2 < nsteps) stage_w(cw, kw, PB);
      if (k < nxi && c + 1 < c_end) stage_x(c + 1, k);
      if (++k == p.K) { k = 0; ++c; }
      if (++kw == p.K) { kw = 0; ++cw; }
      set_xad(c, k);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();
      stamp(s, 6);
      // ---- COMPUTE(2s+1)
      if (OS2S_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < NI; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            acc[in][2 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk],
                                                                      acc[in][2 + i2], 0, 0, 0);
      if (OS2S_PP_PRIO) __builtin_amdgcn_s_setprio(0);
      stamp(s, 7);
      pp_barrier();
      stamp(s, 8);
    };
    for (int s = 0; s < nsteps; s += 2) {
      step(std::integral_constant<int, 0>{}, s);
      if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
    }
    if (!grp) pp_barrier();
    if (DBG && p.dbg && bid < 4) {
      __syncthreads();
      for (int i = tid; i < 2 * 48 * 9; i += 512) p.dbg[bid * 2 * 48 * 9 + i] = tl[i];
    }
  }

  if (npiece > 1) {
    // ---- split unit: publish the partial tile, take a ticket; the last arriver reduces ---------
    const int sidx = rank - nfull;
    auto at = [&](int v) -> f32x16& { return acc[v >> 2][v & 3]; };
    if (!split_publish_and_reduce(at, p.ws_slabs + (size_t)sidx * f * kSplitSlabFloats,
                                  p.ws_cnt + sidx, piece, f, smem, tid))
      return;
  }
  conv_epilogue<BM, BN, WM, WN, NWIN>(p, acc, smem, tid, lane, wid, wmid, n0, wb, wt0);
}

// ---------------------------------------------------------------------------------------------
// Narrow ping-pong tile: NWIN (2 or 3) live 128-row windows x 128 output channels per workgroup.
//
// Why it exists: the unit of the 256-column tile is (window pair, 256 columns) — a ragged Jasper batch
// of ~146 live windows gives 146 units for the 384 / 512-channel layers and 219 for the 640 / 768-channel
// ones on 256 CUs, and the 384 / 640 / 896-channel layers pay for half a tile of dead columns. Pieces of
// a reduction split cost a 256 KB fp32 partial tile each (measured: slower, DESIGN round 3). The narrow
// tiles own DISJOINT outputs instead: 3 windows x 128 columns = 0.75 of a unit (196 / 245 units for
// 512 / 640 channels), 2 x 128 = 0.5 (219 units for 384 channels, 511 = two full rounds for 896).
//
// Wave layout: 8 waves = 4 row groups (wr) x 2 column halves (wc); wave (wr, wc) owns the 32-row
// fragments wr*NWIN .. wr*NWIN + NWIN-1 of the 4*NWIN fragments of the tile (a fragment lies in ONE
// window; with three windows a wave straddles two) x 64 columns: NWIN x 2 accumulator tiles, 96 / 64
// registers. Waves 0-3 (row groups 0, 1) are group A, waves 4-7 group B; wave w and w + 4 share a SIMD.
//
// Two slots per 64-deep step (the 256-column kernel has four): in one slot a wave fetches ALL fragments
// of its next step (8 weight + 4*NWIN window reads), drains the LDS-DMA it issued one step ago, issues
// the DMA of the step two ahead; in the other it issues its 8*NWIN MFMAs (768 / 512 matrix-pipe
// cycles) while its SIMD partner loads. The weight tile [128 x 64] is read by both groups one slot
// apart and refilled two steps ahead: ring of THREE (16 KB each). X windows: double-buffered per
// 64-channel chunk as in the 256-column kernel, one instruction per wave and step during the first
// taps of the previous chunk. A window image is R = 128 + (K-1)*dil rows; when R is 4 mod 8 the last
// 8-row DMA instruction of a window runs with lanes 32-63 masked off, so three windows of K = 21
// (148 rows) fit next to the weight ring (162 816 of the 163 840 bytes).
// No reduction split: these tiles are chosen when they give whole rounds of equal units.
// ---------------------------------------------------------------------------------------------
constexpr int kPpnDbgSteps = 24;                     // DBG: recorded steps (9 stamps each, waves 0 and 4)
template <int NWIN, bool DBG>
__device__ __forceinline__ void ppn_body(const ConvArgs& p, const int L, const int scan, const int nw, char* smem) {
  constexpr int BM = 128, BN = 128, WM = 4, WN = 2, MI = NWIN, NI = 2;
  constexpr int WTILE = BN * 128;                     // one weight tile: 128 rows x 64 k (bf16)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = wid >> 1, wc = wid & 1;
  const int P = (L + NWIN - 1) / NWIN;
  const int NT = (p.Cout + BN - 1) / BN;
  const int nwork = P * NT;
  const int bid = blockIdx.x;
  if (bid >= nwork) {
    pp_zero_role(p, nw, (int)gridDim.x - 1 - bid, tid);
    return;
  }
  // ---- unit rank -> (window group, n-tile): consecutive workgroups (= the XCDs round-robin) hold
  // different window groups, the n-tiles of one group follow each other on one XCD ---------------
  int xcd, loc;
  {
    const int P8 = (P + 7) >> 3, rem = P - 8 * (P8 - 1), base = (P8 - 1) * NT * 8;
    if (bid < base) { xcd = bid & 7; loc = bid >> 3; }
    else { const int i = bid - base; loc = (P8 - 1) * NT + i / rem; xcd = i - (i / rem) * rem; }
  }
  const int wgrp = (loc / NT) * 8 + xcd;
  const int n_idx = loc - (loc / NT) * NT;
  const int n0 = n_idx * BN;
  int wb[NWIN], wt0[NWIN], wmid[NWIN];
  // per-window buffer descriptor and first-row offset of the X staging. Always indexed by a
  // COMPILE-TIME window number (stage_x is instantiated per window): a runtime index would put the
  // arrays into scratch and the descriptor into a waterfall loop
  // per-window words of the X buffer descriptors (base, bytes) and first-row offsets, as NAMED scalars: the
  // descriptor of an instruction is assembled from arithmetic selects of these (see prep_x)
  unsigned xlo0 = 0, xlo1 = 0, xlo2 = 0, xhi0 = 0, xhi1 = 0, xhi2 = 0;
  int xby0 = 0, xby1 = 0, xby2 = 0, xr0a = 0, xr0b = 0, xr0c = 0;
#pragma unroll
  for (int w = 0; w < NWIN; ++w) {
    const int i = NWIN * wgrp + w;
    const bool live = i < L;
    const int b = live ? __builtin_popcountll(__ballot(scan <= i)) : 0;
    const int before = b > 0 ? __shfl(scan, b - 1, 64) : 0;
    wb[w] = b;
    wt0[w] = live ? (i - before) * BM : 0;
    int len_b = p.Tin;
    if (p.in_len) {
      const int l = p.in_len[b];
      len_b = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
    }
    wmid[w] = live ? b * p.mtiles_per_b + (i - before) : -1;
    const bf16_t* xb = p.x + (long long)b * p.x_sb;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)xb);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)xb >> 32));
    // dead slot: num_records = 0, every row reads as zero
    const int bytes = __builtin_amdgcn_readfirstlane((live ? len_b : 0) * (int)p.x_st * 2);
    const int row0 = __builtin_amdgcn_readfirstlane((wt0[w] - p.padL) * (int)p.x_st * 2);   // stride 1
    if (w == 0) { xlo0 = lo; xhi0 = hi; xby0 = bytes; xr0a = row0; }
    if (w == 1) { xlo1 = lo; xhi1 = hi; xby1 = bytes; xr0b = row0; }
    if (w == 2) { xlo2 = lo; xhi2 = hi; xby2 = bytes; xr0c = row0; }
  }

  // ---- LDS: X windows [2 chunks][NWIN][win_bytes] | weight ring [3][16 KB] ---------------------
  const int rbw = (p.R + 7) >> 3;                     // 8-row DMA instructions per window
  const bool half = (p.R & 7) != 0 && (p.R & 7) <= 4; // the last one covers 4 rows only
  const int win_bytes = (half ? rbw * 8 - 4 : rbw * 8) * 128;
  const int xbuf_bytes = NWIN * win_bytes;
  char* const xbuf0 = smem;
  char* const wbuf0 = smem + 2 * xbuf_bytes;
  const int npw = (rbw + 7) >> 3;                     // X instructions per wave per window and chunk
  const int nxi = NWIN * npw;                         // ... per wave and chunk (launcher: K > nxi)

  // X DMA: instruction n = m * NWIN + w of a chunk issued by wave wid covers row group rb = wid + 8 m of
  // window w (8 rows x 128 B); per-lane byte offset inside the group: row (lane >> 3), 16-B slot
  // jj ^ swizzle(row), swizzle term (row >> 1) & 7 flips bit 2 for odd rb. Rows before the sequence
  // start give a negative (= huge unsigned) offset, rows past in_len lie past num_records: zeros.
  const int xlane = (lane >> 3) * (int)p.x_st * 2 + (((lane & 7) ^ (lane >> 4)) << 4);
  const int xgrp_bytes = __builtin_amdgcn_readfirstlane(8 * (int)p.x_st * 2);
  // An X instruction is PREPARED (window, row group, LDS destination, per-lane offset, half mask) in
  // one place and ISSUED in another: in the main loop the preparation rides between the MFMAs of the
  // wave's COMPUTE slot, the LOAD slot only carries s_mov m0 + buffer_load (see the slot budget below).
  // Plain scalars, no struct / array: hipcc turns those into scratch accesses whose vmcnt wait in the
  // COMPUTE slot would also wait for the LDS-DMA in flight.
  int x_dst = 0, x_soff = 0, x_vo = 0, x_by = 0;
  unsigned x_lo = 0, x_hi = 0;
  bool x_keep = false;                                // per lane: this lane takes part in the instruction
  auto prep_x = [&](int c, int n, bool on) __attribute__((always_inline)) {   // n = 0 .. nxi-1: window n % NWIN, row group wid + 8 (n / NWIN)
    const int m = n / NWIN;
    const int w = n - m * NWIN;
    const int rb = wid + 8 * m;
    // the last 8-row group of a 4-mod-8 window image is written by lanes 0-31 only
    x_keep = on && rb < rbw && !(half && rb == rbw - 1 && lane >= 32);
    x_dst = (c & 1) * xbuf_bytes + w * win_bytes + rb * 1024;
    x_soff = c * 128;
    // (arithmetic, not a select of the captured variables: hipcc turns that into a select of their
    // ADDRESSES and keeps the whole closure in scratch)
    const int s0 = w == 0, s1 = w == 1, s2 = w >= 2;
    x_lo = xlo0 * s0 + xlo1 * s1 + xlo2 * s2;
    x_hi = xhi0 * s0 + xhi1 * s1 + xhi2 * s2;
    x_by = xby0 * s0 + xby1 * s1 + xby2 * s2;
    const int r0 = xr0a * s0 + xr0b * s1 + xr0c * s2;
    x_vo = (xlane ^ ((rb & 1) << 6)) + r0 + rb * xgrp_bytes;
  };
  auto issue_x = [&]() __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(x_hi) << 32) |
                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(x_lo)),
        0, __builtin_amdgcn_readfirstlane(x_by), 0x00020000);
    if (x_keep)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(xbuf0 + x_dst), 16, x_vo,
                                               __builtin_amdgcn_readfirstlane(x_soff), 0, 0);
  };
  auto stage_x = [&](int c, int n) __attribute__((always_inline)) { prep_x(c, n, true); issue_x(); };
  // weight tile rows handled by this lane (2 DMA instructions of 8 rows per wave and step)
  // the weight descriptor is rebuilt per step from three values parked in VGPRs (opaque to the
  // compiler): left to itself it re-reads p.w from the kernel-argument segment in every LOAD slot
  // (SGPRs are short here) and the s_waitcnt lgkmcnt(0) behind that s_load also waits for the 20
  // fragment reads just issued, before the first DMA instruction of the slot can go out
  unsigned wlo_v = (unsigned)(unsigned long long)p.w, whi_v = (unsigned)((unsigned long long)p.w >> 32);
  unsigned wbytes_v = (unsigned)((long long)p.K * p.Cout * p.Cin * 2);
  asm volatile("" : "+v"(wlo_v), "+v"(whi_v), "+v"(wbytes_v));
  int wsrc[2];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    const int row = (pi * 8 + wid) * 8 + (lane >> 3), jj = lane & 7;
    const int j = jj ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;
    wsrc[pi] = (n * p.Cin + j * 8) * 2;
  }
  const int w_kstride = __builtin_amdgcn_readfirstlane(p.Cout * p.Cin * 2);
  auto stage_w_at = [&](int soff_in, int buf) __attribute__((always_inline)) {
    const int soff = __builtin_amdgcn_readfirstlane(soff_in);
    // (readfirstlane returns int: without the unsigned casts a low word with bit 31 set sign-extends
    // over the high word of the base address)
    const unsigned wlo = (unsigned)__builtin_amdgcn_readfirstlane(wlo_v);
    const unsigned whi = (unsigned)__builtin_amdgcn_readfirstlane(whi_v);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)whi << 32) | (unsigned long long)wlo), 0,
        (int)__builtin_amdgcn_readfirstlane(wbytes_v), 0x00020000);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(wbuf0 + buf * WTILE + (pi * 8 + wid) * 1024), 16,
          wsrc[pi], soff, 0, 0);
  };
  auto stage_w = [&](int c, int k, int buf) __attribute__((always_inline)) { stage_w_at(k * w_kstride + c * 128, buf); };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int im = 0; im < MI; ++im)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[in][im][e] = 0.f;

  const int nsteps = p.nchunks * p.K;
  {
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int n = 0; n < nxi; ++n) stage_x(0, n);
    stage_w(0, 0, 0);
    if (nsteps > 1) stage_w(p.K > 1 ? 0 : 1, p.K > 1 ? 1 : 0, 1);
    // weight fragment offsets: row = wc*64 + in*32 + l31 of the tile, fixed for the kernel
    int woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      woff[kk] = (wc * 64 + l31) * 128 + (((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4);
    // X fragment im = rows (wr*MI + im)*32 .. +31 of the tile: window (f >> 2), rows (f & 3)*32 of it
    int ximg[MI];
#pragma unroll
    for (int im = 0; im < MI; ++im) {
      const int f = wr * MI + im;
      ximg[im] = (f >> 2) * win_bytes + ((f & 3) * 32) * 128;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp) pp_barrier();                            // group B runs one slot behind group A

    int c = 0, k = 0;     // step s
    int cw = 0, kw = 0;   // step s + 2
    for (int i = 0; i < 2; ++i) { if (++kw == p.K) { kw = 0; ++cw; } }
    // LDS byte addresses of this lane's X fragments of the CURRENT step (refreshed at the end of LOAD)
    const char* xad[MI][4];
    auto set_xad = [&](int cc, int kk_tap) __attribute__((always_inline)) {
      const int r0 = l31 + kk_tap * p.dil;            // row inside the 32-row fragment's window image
      const int m = (r0 >> 1) & 7;                    // ((f&3)*32 + r0) >> 1 & 7 == (r0 >> 1) & 7
      const char* const xrow = xbuf0 + (cc & 1) * xbuf_bytes + r0 * 128;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int sw = ((kk * 2 + lhi) ^ m) << 4;
#pragma unroll
        for (int im = 0; im < MI; ++im) xad[im][kk] = xrow + ximg[im] + sw;
      }
    };
    set_xad(0, 0);
    // DBG: s_memtime stamps of waves 0 and 4 of workgroups 0..3 (tools/pp_timeline.py): 0 first read batch
    // issued, 1 DMA drained, 2 DMA issued, 3 second read batch issued, 4 arithmetic done, 5 reads
    // landed, 6 barrier passed (MFMAs start), 7 MFMAs issued, 8 barrier passed
    unsigned long long* const tl = reinterpret_cast<unsigned long long*>(wbuf0 + 3 * WTILE);
    const bool rec = DBG && p.dbg && bid < 4 && lane == 0 && (wid & 3) == 0;
    auto stamp = [&](int s, int i) __attribute__((always_inline)) {
      if (DBG && rec && s >= 8 && s < 8 + kPpnDbgSteps) tl[(grp * kPpnDbgSteps + s - 8) * 9 + i] = __builtin_readcyclecounter();
    };
    // Slot budget. While a wave's SIMD partner issues its MFMAs back to back, the loading wave gets
    // roughly one instruction out per MFMA (ppn timeline, tools/ppn_timeline.py): a LOAD slot of ~100
    // instructions (20 fragment reads, DMA issue with its address arithmetic, the (c, k) counters, 12
    // fragment addresses of the next step) ran 920-1060 cycles next to 768 cycles of MFMAs. So the LOAD
    // slot carries only what must be there — the reads, the drain, s_mov m0 + buffer_load per DMA — and
    // everything that PREPARES the next LOAD slot (counters, DMA operands, fragment addresses) is issued
    // by the computing wave between its own MFMAs (5 issue slots per 32-cycle MFMA are free there).
    // pp_prio: the loading wave raises its priority for the slot (its few instructions go out at once).
    int w_soff = kw * w_kstride + cw * 128;           // weight DMA of the NEXT load slot (step s + 2)
    prep_x(1, 0, 1 < p.nchunks);                      // X DMA of the next load slot
    auto step = [&](auto WBUF, int s) __attribute__((always_inline)) {
      constexpr int WB = decltype(WBUF)::value;
      const char* const ws = wbuf0 + WB * WTILE;
      bf16x8 wf[NI][4], xf[MI][4];
      if (DBG) {                                      // (the timing-only modes skip the reads)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int in = 0; in < NI; ++in) wf[in][kk] = bf16x8{};
#pragma unroll
          for (int im = 0; im < MI; ++im) xf[im][kk] = bf16x8{};
        }
      }
      // ---- LOAD(s)
      if (p.pp_prio) __builtin_amdgcn_s_setprio(2);
      const bool rd = !(DBG && (p.dbg_fixed_w & 4)) || s < 2;
      if (rd) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int in = 0; in < NI; ++in)
            wf[in][kk] = *reinterpret_cast<const bf16x8*>(ws + woff[kk] + in * 4096);
          xf[0][kk] = *reinterpret_cast<const bf16x8*>(xad[0][kk]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(s, 0);
      // drain what this wave issued in LOAD(s-1) (a whole step of flight time), then issue the weight
      // tile of step s+2 (its ring slot was last read in the other group's LOAD(s-1)) and one X
      // instruction of the next chunk; the barrier at the end of this slot publishes the drained tiles
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(s, 1);
      if (!(DBG && (p.dbg_fixed_w & 2))) {
        if (s + 2 < nsteps) stage_w_at(w_soff, (WB + 2) % 3);
        issue_x();
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(s, 2);
      if (rd) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int im = 1; im < MI; ++im)
            xf[im][kk] = *reinterpret_cast<const bf16x8*>(xad[im][kk]);
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(s, 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(s, 5);
      if (p.pp_prio) __builtin_amdgcn_s_setprio(0);
      pp_barrier();
      stamp(s, 6);
      // ---- COMPUTE(s): the MFMAs, and between them the preparation of LOAD(s+1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < NI; ++in)
#pragma unroll
          for (int im = 0; im < MI; ++im)
            acc[in][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[im][kk], acc[in][im], 0, 0, 0);
      if (++k == p.K) { k = 0; ++c; }
      if (++kw == p.K) { kw = 0; ++cw; }
      w_soff = kw * w_kstride + cw * 128;
      prep_x(c + 1, k, k < nxi && c + 1 < p.nchunks);
      set_xad(c, k);
      // one MFMA, then up to two vector and three scalar instructions of the preparation, and so on
#pragma unroll
      for (int i = 0; i < MI * NI * 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
      }
      if (DBG) __builtin_amdgcn_sched_barrier(0);
      stamp(s, 7);
      pp_barrier();
      stamp(s, 8);
    };
    for (int s = 0; s < nsteps; s += 3) {
      step(std::integral_constant<int, 0>{}, s);
      if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
      if (s + 2 < nsteps) step(std::integral_constant<int, 2>{}, s + 2);
    }
    if (!grp) pp_barrier();
    if (DBG && p.dbg && bid < 4) {
      __syncthreads();
      for (int i = tid; i < 2 * kPpnDbgSteps * 9; i += 512) p.dbg[bid * 2 * kPpnDbgSteps * 9 + i] = tl[i];
    }
  }
  conv_epilogue<BM, BN, WM, WN, NWIN>(p, acc, smem, tid, lane, wid, wmid, n0, wb, wt0);
}

constexpr int kConvBM = 128, kConvBN = 128;

template <int BM, int BN, int WM, int WN, int NWIN, bool XSINGLE = false>
static int launch_conv(hipStream_t stream, ConvArgs& a) {
  constexpr int NTHR = WM * WN * 64;
  a.mtiles_per_b = ceil_div(a.Tout, BM);
  a.MT = a.B * a.mtiles_per_b;
  a.MT8 = ceil_div(ceil_div(a.MT, NWIN), 8);
  a.NT = ceil_div(a.Cout, BN);
  a.nchunks = ceil_div(a.Cin, 64);
  a.R = (BM - 1) * a.stride + (a.K - 1) * a.dil + 1;
  a.Rpad = ceil_div(a.R, 8) * 8;
  size_t main_bytes = (size_t)(XSINGLE ? 1 : 2) * NWIN * a.Rpad * 128 + (size_t)2 * BN * 128;
  size_t epi_bytes = conv_epilogue_lds_bytes<BM, BN, NWIN, NTHR>();
  size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  if (smem > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_igemm_kernel<BM, BN, WM, WN, NWIN, XSINGLE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  const int grid = a.MT8 * 8 * a.NT;
  OS2S_LAUNCH((conv1d_igemm_kernel<BM, BN, WM, WN, NWIN, XSINGLE>), dim3(grid), dim3(NTHR), smem,
              stream, a);
  return OS2S_OK;
}

// Ping-pong kernel launcher; OS2S_ERR_UNSUPPORTED when the shape is outside its envelope
// (stride 1, Cin a multiple of 64, K long enough to spread the X prefetch, B <= 64, LDS budget).
// workspace = [1024 int32 tickets, zero on entry and on exit][fp32 partial tiles]; without one
// the tail of the launch is not split.
static int g_pp_prio = 0;
// fitted us per 64-deep step of the 2x256, 2x128, 3x128 tile (tools/bench_conv_shapes.py on the bench batch:
// 0.261 / 0.232 ms for 640 channels K = 21, 0.091 / 0.066 for 384 K = 13, 0.611 / 0.724 for 896 K = 29) and the
// data-gradient penalty: >= 100 keeps the narrow tiles out of data-gradient launches altogether — next to the
// weight-gradient stream what counts is CU time, and a narrow tile needs 10-19 % MORE of it per output
// (measured: Jasper step 40.6 -> 42.0 ms with the narrow tiles in backward, 39.16 -> 39.05 forward only)
static float g_pp_cost[4] = {1.19f, 0.90f, 1.06f, 1000.f};
// host copy of the sequence lengths of the forward launches that follow (os2s_conv1d_set_host_lens)
static std::vector<int32_t> g_host_lens;
static bool g_host_lens_set = false;

// LDS bytes of the narrow tile's main loop (ppn_body): X windows double-buffered + weight ring of 3
static size_t ppn_main_bytes(int nwin, int R) {
  const int rbw = (R + 7) / 8;
  const bool half = (R & 7) != 0 && (R & 7) <= 4;
  const size_t win = (size_t)(half ? rbw * 8 - 4 : rbw * 8) * 128;
  return 2 * nwin * win + (size_t)3 * 128 * 128;
}

// tile: -1 = chosen on the device (pp_choose_tile), 0 / 2 / 3 = forced (2 / 3 fall back to 0 when the
// narrow tile does not fit the layer)
static int launch_conv_pp(hipStream_t stream, ConvArgs& a, void* workspace, size_t workspace_bytes, int tile) {
  constexpr int BM = 128, BN = kPpBN, NWIN = 2, NTHR = 512;
  if (a.stride != 1 || a.Cin % 64 != 0 || a.out_f32 || a.B > 64) return OS2S_ERR_UNSUPPORTED;
  a.mtiles_per_b = ceil_div(a.Tout, BM);
  a.MT = a.B * a.mtiles_per_b;
  a.MT8 = 0;
  a.NT = ceil_div(a.Cout, BN);
  a.nchunks = a.Cin / 64;
  a.R = (BM - 1) + (a.K - 1) * a.dil + 1;
  a.Rpad = ceil_div(a.R, 8) * 8;
  const int nxi = (a.Rpad / 8 + 3) / 4;
  if (a.K <= nxi) return OS2S_ERR_UNSUPPORTED;
  const bool dbg_any = a.dbg != nullptr || a.dbg_fixed_w;
  const bool dbg_n = dbg_any && (tile == 2 || tile == 3);     // stamps of the narrow tile (forced)
  const bool dbg = dbg_any && !dbg_n;
  const size_t dbg_n_bytes = dbg_n ? (size_t)2 * kPpnDbgSteps * 9 * 8 : 0;
  const size_t main_bytes = (size_t)4 * a.Rpad * 128 + (size_t)2 * BN * 128 + 1024 + (dbg ? 2 * 48 * 9 * 8 : 0);
  const size_t epi_bytes = conv_epilogue_lds_bytes<BM, BN, NWIN, NTHR>();
  const size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  if (smem > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  // ---- narrow tiles (ppn_body): K must spread the X prefetch of a chunk, LDS must hold the images;
  // no dropout in the epilogue of the three-window tile (conv_epilogue, STRADDLE)
  const int npw = (ceil_div(a.R, 8) + 7) / 8;
  size_t smem_n = 0;
  a.pp_ok2 = a.pp_ok3 = 0;
  const bool no_narrow = tile < 0 && a.out_len != nullptr && g_pp_cost[3] >= 100.f;
  if (!dbg && a.Cout >= 128 && tile != 0 && !no_narrow) {
    const size_t m2 = ppn_main_bytes(2, a.R) + dbg_n_bytes, e2 = conv_epilogue_lds_bytes<BM, 128, 2, NTHR>();
    const size_t m3 = ppn_main_bytes(3, a.R) + dbg_n_bytes, e3 = conv_epilogue_lds_bytes<BM, 128, 3, NTHR>();
    // (a cost of 1e6 or more takes a narrow tile out of the candidates: A/B runs)
    if (tile != 3 && a.K > 2 * npw && m2 <= 160 * 1024 && e2 <= 160 * 1024 && (tile == 2 || g_pp_cost[1] < 1e6f)) a.pp_ok2 = 1;
    if (tile != 2 && a.K > 3 * npw && m3 <= 160 * 1024 && e3 <= 160 * 1024 && a.keep_prob >= 1.f &&
        (tile == 3 || g_pp_cost[2] < 1e6f)) a.pp_ok3 = 1;
    if (a.pp_ok2) smem_n = std::max(smem_n, std::max(m2, e2));
    if (a.pp_ok3) smem_n = std::max(smem_n, std::max(m3, e3));
  }
  if ((tile == 2 && !a.pp_ok2) || (tile == 3 && !a.pp_ok3)) tile = 0;
  a.pp_tile = tile;
  a.pp_c256 = g_pp_cost[0]; a.pp_c2 = g_pp_cost[1]; a.pp_c3 = g_pp_cost[2]; a.pp_dgrad_pen = g_pp_cost[3];
  a.pp_prio = g_pp_prio;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_pp_kernel<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute((const void*)conv1d_pp_kernel<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute((const void*)conv1d_ppn_kernel<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute((const void*)conv1d_ppn_kernel<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n;
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  a.ncu = ncu;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0;
  const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
  if (workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
    a.ws_cnt = reinterpret_cast<int*>(workspace);
    a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
    size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
    const size_t cap = (size_t)3 * ncu;
    a.ws_nslabs = (int)(n < cap ? n : cap);
  }
  // a forward launch whose lengths the host knows (no mask at all, or the caller's host copy): the tile
  // choice is evaluated here — the same pure function the device evaluates — and ONE kernel is enqueued
  if (tile < 0 && !a.out_len && (a.pp_ok2 || a.pp_ok3) &&
      (a.in_len == nullptr || (g_host_lens_set && (int)g_host_lens.size() == a.B))) {
    int L = 0;
    for (int b = 0; b < a.B; ++b) {
      int len = a.Tin;
      if (a.in_len) len = std::min(std::max(g_host_lens[b], 0), a.Tin);
      const int nw = len > 0 ? (len + a.padL + BM - 1) / BM : 0;
      L += std::min(nw, a.mtiles_per_b);
    }
    tile = a.pp_tile = pp_choose_tile(a, L);
  }
  // upper bound of each grid (the live-window count — and with it the tile, unless the host knew the
  // lengths — is only known on the device): every unit of the padded batch + the pieces of a split tail +
  // the store-only workgroups; surplus workgroups exit at once
  const int nzero = a.out_len ? 0 : ceil_div(a.MT, kPpZeroWin);
  if (tile <= 0) {
    const int grid = ceil_div(a.MT, NWIN) * a.NT + a.ws_nslabs + nzero;
    if (dbg) {
      OS2S_LAUNCH(conv1d_pp_kernel<true>, dim3(grid), dim3(NTHR), smem, stream, a);
    } else {
      OS2S_LAUNCH(conv1d_pp_kernel<false>, dim3(grid), dim3(NTHR), smem, stream, a);
    }
  }
  if (tile != 0 && (a.pp_ok2 || a.pp_ok3)) {
    const int grid = ceil_div(a.MT, 2) * ceil_div(a.Cout, 128) + nzero;
    if (dbg_n) {
      OS2S_LAUNCH(conv1d_ppn_kernel<true>, dim3(grid), dim3(NTHR), smem_n, stream, a);
    } else {
      OS2S_LAUNCH(conv1d_ppn_kernel<false>, dim3(grid), dim3(NTHR), smem_n, stream, a);
    }
  }
  return OS2S_OK;
}

}  // namespace os2s

// Tile choice is a fixed function of the problem shape (no timing, no hidden state):
//   * ping-pong kernel (256 x 256 per workgroup, balanced over the live windows) whenever the
//     shape is inside its envelope and the layer is at least 320 channels wide (the 256-channel
//     layers: a whole unit is shorter than the fixed costs of its workgroup, 0.057 vs 0.054 ms);
//   * otherwise the lockstep kernel: 256 x 256 tile for wide 1x1 / short-K layers with enough
//     rows to fill the chip, else the 128 x 128 tile (single-buffered X window for K >= 8).
// os2s_set_option("conv1d.variant", v >= 0) forces a tile for experiments and tests:
//   0 = 128x128, X window double-buffered   3 = 128x128, X window single-buffered when K >= 8
//   5 = 256x256 lockstep                   10 = ping-pong (tile chosen on the device)
//   12 / 13 = ping-pong, 2 / 3 windows x 128 columns   14 = ping-pong, 2 windows x 256 columns
// narrowest layer the ping-pong kernels take (conv1d.pp_min_cout; the environment variable OS2S_PP_MIN_COUT is
// read ONCE, at the first launch — no per-launch environment or map lookups in the launch path)
static int g_pp_min_cout = -1;
static int pp_min_cout() {
  if (g_pp_min_cout < 0) {
    const char* v = getenv("OS2S_PP_MIN_COUT");
    g_pp_min_cout = v ? atoi(v) : 320;
  }
  return g_pp_min_cout;
}
static int g_conv_variant = -1;
static int g_conv_split = -1;
static unsigned long long* g_conv_dbg = nullptr;
static int g_conv_fixed_w = 0;
extern "C" int os2s_conv1d_set_host_lens(const int32_t* lens, int B) {
  if (lens == nullptr || B <= 0) {
    os2s::g_host_lens_set = false;
    os2s::g_host_lens.clear();
    return OS2S_OK;
  }
  os2s::g_host_lens.assign(lens, lens + B);
  os2s::g_host_lens_set = true;
  return OS2S_OK;
}
// Named options of the convolution launchers (os2s_set_option; nothing here is read per launch):
//   conv1d.variant            >= 0 forces a tile (list above), -1 = by shape
//   conv1d.split              > 0 forces the tail split factor of the ping-pong kernel, -1 = cost model
//   conv1d.pp_cost_256 / .pp_cost_2x128 / .pp_cost_3x128   fitted microseconds per 64-deep step of the three
//       ping-pong tiles — the constants of the device-side tile choice (tools/bench_conv_shapes.py refits
//       them); a cost >= 1e6 removes a narrow tile from the candidates
//   conv1d.pp_dgrad_penalty   factor on the narrow tiles' cost in data-gradient launches (out_len given)
//   conv1d.pp_prio            1: the loading wave of a narrow-tile slot runs at s_setprio 2
//   conv1d.pp_min_cout        narrowest layer (output channels) the ping-pong kernels take (default 320)
//   conv1x1.variant           the 1x1 launches: 0 / 1 = lockstep 128x128 tile, 2 = 256x256 ping-pong tile,
//                             -1 (default) = ping-pong for a single K = 1 layer of >= pp_min_cout output channels,
//                             lockstep for the grouped launches
static os2s::OptionReg r_variant("conv1d.variant", [](double v) { g_conv_variant = (int)v; });
static os2s::OptionReg r_split("conv1d.split", [](double v) { g_conv_split = (int)v; });
static os2s::OptionReg r_c256("conv1d.pp_cost_256", [](double v) { os2s::g_pp_cost[0] = (float)v; });
static os2s::OptionReg r_c2("conv1d.pp_cost_2x128", [](double v) { os2s::g_pp_cost[1] = (float)v; });
static os2s::OptionReg r_c3("conv1d.pp_cost_3x128", [](double v) { os2s::g_pp_cost[2] = (float)v; });
static os2s::OptionReg r_pen("conv1d.pp_dgrad_penalty", [](double v) { os2s::g_pp_cost[3] = (float)v; });
static os2s::OptionReg r_prio("conv1d.pp_prio", [](double v) { os2s::g_pp_prio = (int)v; });
static os2s::OptionReg r_min("conv1d.pp_min_cout", [](double v) { g_pp_min_cout = (int)v; });
// debug stamps "conv1d" (tools/pp_timeline.py, tools/ppn_timeline.py): device buffer of slot time stamps;
// mode (256-column tile) = every step streams the weight tile of step 0 (always an L2 hit); narrow
// tiles: bit 1 = no DMA issue in the loop, bit 2 = no fragment reads in the loop (timing only)
static os2s::StampReg r_stamps("conv1d", [](void* stamps, int mode) {
  g_conv_dbg = (unsigned long long*)stamps;
  g_conv_fixed_w = mode;
});

static int g_conv1x1_variant = -1;
static os2s::OptionReg r_1x1("conv1x1.variant", [](double v) { g_conv1x1_variant = (int)v; });
static int g_conv1x1_order = 1;
static os2s::OptionReg r_1x1o("conv1x1.order", [](double v) { g_conv1x1_order = v != 0.0; });

extern "C" int os2s_conv1d_num_mtiles(int B, int Tout) {
  return B * os2s::ceil_div(Tout, os2s::kConvBM);
}

extern "C" size_t os2s_conv1d_workspace_bytes(void) {
  return os2s::kSplitTicketBytes + (size_t)3 * 256 * os2s::kSplitSlabFloats * 4;
}

static int conv1d_fwd_impl(os2s_stream_t stream, const uint16_t* x, const uint16_t* w, void* y,
                           const int32_t* in_len, const float* bias, float* stats, int B,
                           int Tin, int Cin, int Cout, int K, int stride, int dil, int padL,
                           int Tout, long long y_stride_b, long long y_stride_t, int out_f32,
                           int accumulate, int act, float keep_prob, unsigned long long seed,
                           const uint16_t* residual, const int32_t* out_len, void* workspace,
                           size_t workspace_bytes, const uint16_t* mask_ref = nullptr, float mask_scale = 1.f,
                           const uint16_t* stat_ref = nullptr) {
  using namespace os2s;
  if (mask_ref) OS2S_REQUIRE(!residual && !out_f32 && act == 0 && keep_prob == 1.f && !bias);
  if (stat_ref) OS2S_REQUIRE(mask_ref && stats);
  OS2S_REQUIRE(act == 0 || act == 1 || act == 3);
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  if (out_f32) OS2S_REQUIRE(act == 0 && keep_prob == 1.f && residual == nullptr);
  OS2S_REQUIRE(x && w && y);
  OS2S_REQUIRE(B >= 0 && Tin >= 1 && Tout >= 1 && Cin >= 8 && Cout >= 1 && K >= 1);
  OS2S_REQUIRE(stride >= 1 && dil >= 1);
  OS2S_REQUIRE(Cin % 8 == 0);
  if (!out_f32) OS2S_REQUIRE(Cout % 8 == 0 && y_stride_t % 8 == 0 && y_stride_b % 8 == 0);
  if (B == 0) return OS2S_OK;
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.in_len = in_len; a.out_len = out_len; a.bias = bias; a.stats = stats;
  a.B = B; a.Tin = Tin; a.Tout = Tout; a.Cin = Cin; a.Cout = Cout; a.K = K;
  a.stride = stride; a.dil = dil; a.padL = padL;
  a.x_sb = (long long)Tin * Cin; a.x_st = Cin;
  a.y_sb = y_stride_b; a.y_st = y_stride_t;
  a.out_f32 = out_f32; a.accumulate = accumulate;
  a.act = act; a.keep_prob = keep_prob; a.seed = seed; a.residual = residual;
  a.mask_ref = mask_ref; a.mask_scale = mask_scale; a.stat_ref = stat_ref;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256;
  a.force_split = g_conv_split;
  a.dbg = g_conv_dbg; a.dbg_fixed_w = g_conv_fixed_w;
  hipStream_t st = (hipStream_t)stream;
  int v = g_conv_variant;
  if (v < 0) {
    v = K >= 8 ? 3 : 0;
    if (Cout >= 512 && (long long)B * ceil_div(Tout, kConvBM) >= 256) v = 5;
    if (Cout >= pp_min_cout()) v = 10;
  }
  // K = 1: the lockstep tile's life is its 4 - 16 steps of exposed load latency (1.1 - 2 us each, one step of
  // prefetch: 33.7 us at 512 -> 512 channels, 26 752 rows); the ping-pong tile's rings hide it (26.3 us; 1024 ->
  // 1024: 93 -> 68 us; QuartzNet step -0.8 ms, profiles/r06_quartznet_pointwise_ab.log)
  if ((v == 10 || v == 11) && K == 1 && g_conv1x1_variant != 0 && g_conv1x1_variant != 1 && residual == nullptr && Cout >= 256 &&
      y_stride_t == Cout && y_stride_b == (long long)Tout * Cout) {
    // a wide 1x1 layer is the one-group case of the grouped ping-pong launch
    ConvGroupTable gt;
    gt.ngroups = 1;
    gt.g[0].x = x; gt.g[0].w = w; gt.g[0].y = y; gt.g[0].stats = stats;
    gt.g[0].Cin = Cin; gt.g[0].Cout = Cout; gt.g[0].accumulate = accumulate ? 1 : 0; gt.g[0].tile_begin = 0;
    const int rc = launch_conv1x1_pp(st, a, gt);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  if (v == 11) v = 10;
  if (v == 10 || (v >= 12 && v <= 14)) {
    // the dense-residual / accumulate epilogue variants are all supported by the ping-pong kernel;
    // 10: tile chosen on the device, 12 / 13: two / three windows x 128 columns, 14: two windows x 256
    const int rc = launch_conv_pp(st, a, workspace, workspace_bytes, v == 10 ? -1 : (v == 14 ? 0 : v - 10));
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
    v = K >= 8 ? 3 : 0;
    if (g_conv_variant < 0 && Cout >= 512 && (long long)B * ceil_div(Tout, kConvBM) >= 256) v = 5;
  }
  if (v == 5) {
    const int rc = launch_conv<kConvBM, 256, 2, 4, 2, false>(st, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
    v = K >= 8 ? 3 : 0;
  }
  if (v == 3 && K >= 8) return launch_conv<kConvBM, kConvBN, 2, 2, 1, true>(st, a);
  return launch_conv<kConvBM, kConvBN, 2, 2, 1>(st, a);
}

extern "C" int os2s_conv1d_fwd_ws(os2s_stream_t stream, const uint16_t* x, const uint16_t* w,
                                  void* y, const int32_t* in_len, const float* bias,
                                  float* stats, int B, int Tin, int Cin, int Cout, int K,
                                  int stride, int dil, int padL, int Tout,
                                  long long y_stride_b, long long y_stride_t, int out_f32,
                                  int accumulate, int act, float keep_prob,
                                  unsigned long long seed, const uint16_t* residual,
                                  const int32_t* out_len, void* workspace,
                                  size_t workspace_bytes) {
  return conv1d_fwd_impl(stream, x, w, y, in_len, bias, stats, B, Tin, Cin, Cout, K, stride, dil,
                         padL, Tout, y_stride_b, y_stride_t, out_f32, accumulate, act, keep_prob,
                         seed, residual, out_len, workspace, workspace_bytes);
}

// The data gradient of a convolution whose INPUT is the output of a conv + BatchNorm + ReLU (+ dropout)
// layer (parts/cnns/conv_blocks.py:170-232), as the LAST contribution to that output's gradient:
//   dx (+)= conv(dy, flipped transposed weights);  dz = (mask_ref > 0) ? dx * mask_scale : 0  -> dx
//   stats[window, 0, c] = sum_rows dz,  stats[window, 1, c] = sum_rows dz * stat_ref
// mask_ref = that layer's saved output (zero where the ReLU or the dropout mask was off), mask_scale =
// 1 / keep_prob, stat_ref = its convolution output: the partial sums of its BatchNorm backward come out of
// this launch and its own reduction pass (os2s_bn_act_bwd_reduce) is not needed. stats must be ZERO on
// entry (windows past out_len are not visited). Layouts of mask_ref / stat_ref: as dx.
extern "C" int os2s_conv1d_dgrad_bnact_ws(os2s_stream_t stream, const uint16_t* dy, const uint16_t* wT, void* dx,
                                          float* stats, int B, int Tin, int Cin, int Cout, int K, int dil,
                                          int padL, int Tout, int accumulate, const int32_t* out_len,
                                          const uint16_t* mask_ref, float mask_scale, const uint16_t* stat_ref,
                                          void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(mask_ref && stat_ref && stats);
  return conv1d_fwd_impl(stream, dy, wT, dx, nullptr, nullptr, stats, B, Tin, Cin, Cout, K, 1, dil, padL, Tout,
                         (long long)Tout * Cout, Cout, 0, accumulate, 0, 1.f, 0ull, nullptr, out_len, workspace,
                         workspace_bytes, mask_ref, mask_scale, stat_ref);
}

extern "C" int os2s_conv1d_fwd_ex(os2s_stream_t stream, const uint16_t* x, const uint16_t* w,
                                  void* y, const int32_t* in_len, const float* bias,
                                  float* stats, int B, int Tin, int Cin, int Cout, int K,
                                  int stride, int dil, int padL, int Tout,
                                  long long y_stride_b, long long y_stride_t, int out_f32,
                                  int accumulate, int act, float keep_prob,
                                  unsigned long long seed, const uint16_t* residual,
                                  const int32_t* out_len) {
  return conv1d_fwd_impl(stream, x, w, y, in_len, bias, stats, B, Tin, Cin, Cout, K, stride, dil,
                         padL, Tout, y_stride_b, y_stride_t, out_f32, accumulate, act, keep_prob,
                         seed, residual, out_len, nullptr, 0);
}

extern "C" int os2s_conv1d_fwd(os2s_stream_t stream, const uint16_t* x, const uint16_t* w,
                               void* y, const int32_t* in_len, const float* bias, float* stats,
                               int B, int Tin, int Cin, int Cout, int K, int stride, int dil,
                               int padL, int Tout, long long y_stride_b, long long y_stride_t,
                               int out_f32, int accumulate) {
  return conv1d_fwd_impl(stream, x, w, y, in_len, bias, stats, B, Tin, Cin, Cout, K, stride, dil,
                         padL, Tout, y_stride_b, y_stride_t, out_f32, accumulate, 0, 1.f, 0,
                         nullptr, nullptr, nullptr, 0);
}

// Grouped 1x1 convolutions (see conv1d_igemm_grouped_kernel). groups is a HOST array.
static int conv1x1_fwd_grouped_impl(os2s_stream_t stream, const os2s_conv_group_t* groups,
                                    int ngroups, const int32_t* in_len,
                                    const int32_t* out_len, int B, int T, int out_f32) {
  using namespace os2s;
  OS2S_REQUIRE(groups && ngroups >= 1 && ngroups <= kMaxConvGroups && B >= 0 && T >= 1);
  if (B == 0) return OS2S_OK;
  constexpr int BM = 128, BN = 128;
  ConvArgs a;
  a.x = nullptr; a.w = nullptr; a.y = nullptr; a.stats = nullptr;
  a.in_len = in_len; a.out_len = out_len; a.bias = nullptr;
  a.B = B; a.Tin = T; a.Tout = T; a.Cin = 0; a.Cout = 0; a.K = 1;
  a.stride = 1; a.dil = 1; a.padL = 0;
  a.x_sb = 0; a.x_st = 0; a.y_sb = 0; a.y_st = 0;
  a.out_f32 = out_f32 ? 1 : 0; a.accumulate = 0; a.act = 0; a.keep_prob = 1.f; a.seed = 0; a.residual = nullptr;
  a.mask_ref = nullptr; a.mask_scale = 1.f; a.stat_ref = nullptr;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = -1;
  a.dbg = g_conv_dbg; a.dbg_fixed_w = 0;     // experiment hook (conv1x1_pp_kernel phase stamps)
  a.mtiles_per_b = ceil_div(T, BM);
  a.MT = B * a.mtiles_per_b;
  a.MT8 = ceil_div(a.MT, 8);
  a.NT = 0; a.nchunks = 0;
  a.R = BM; a.Rpad = BM;
  a.tile_order = g_conv1x1_order;
  ConvGroupTable gt;
  gt.ngroups = ngroups;
  int tiles = 0;
  for (int i = 0; i < ngroups; ++i) {
    const os2s_conv_group_t& g = groups[i];
    OS2S_REQUIRE(g.x && g.w && g.y && g.Cin >= 8 && g.Cout >= 8 && g.Cin % 8 == 0 && g.Cout % 8 == 0);
    if (out_f32) OS2S_REQUIRE(g.stats == nullptr);
    gt.g[i].x = g.x; gt.g[i].w = g.w; gt.g[i].y = g.y; gt.g[i].stats = g.stats;
    gt.g[i].Cin = g.Cin; gt.g[i].Cout = g.Cout; gt.g[i].accumulate = g.accumulate ? 1 : 0;
    gt.g[i].tile_begin = tiles;
    tiles += a.MT8 * 8 * ceil_div(g.Cout, BN);
  }
  for (int i = ngroups; i < kMaxConvGroups; ++i) gt.g[i] = gt.g[0];
  gt.total_tiles = tiles;
  // The 256 x 256 ping-pong tile (conv1x1_pp_kernel) is available behind os2s_set_option("conv1x1.variant", 2)
  // only: a 1x1 unit is 4-12 steps of matrix work followed by 128 KB of output, one workgroup per CU
  // cannot overlap the two (measured: epilogue 25 us of a 40 us unit; Jasper step +1 ms), while two
  // or three lockstep workgroups per CU do.
  if (g_conv1x1_variant == 2 && !out_f32) {
    const int rc = launch_conv1x1_pp((hipStream_t)stream, a, gt);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  const size_t main_bytes = (size_t)2 * a.Rpad * 128 + (size_t)2 * BN * 128;
  constexpr size_t kOP = BN * 2 + 16;
  const size_t epi_bytes = conv_epilogue_lds_bytes<BM, BN, 1, 256>();
  const size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_igemm_grouped_kernel<128, 128, 2, 2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH((conv1d_igemm_grouped_kernel<128, 128, 2, 2>), dim3(tiles), dim3(256), smem,
              (hipStream_t)stream, a, gt);
  return OS2S_OK;
}

extern "C" int os2s_conv1x1_fwd_grouped(os2s_stream_t stream, const os2s_conv_group_t* groups,
                                        int ngroups, const int32_t* in_len,
                                        const int32_t* out_len, int B, int T) {
  return conv1x1_fwd_grouped_impl(stream, groups, ngroups, in_len, out_len, B, T, 0);
}

extern "C" int os2s_conv1x1_fwd_grouped_ex(os2s_stream_t stream, const os2s_conv_group_t* groups,
                                           int ngroups, const int32_t* in_len,
                                           const int32_t* out_len, int B, int T, int out_f32) {
  return conv1x1_fwd_grouped_impl(stream, groups, ngroups, in_len, out_len, B, T, out_f32);
}

// 1x1 convolution between channel slices of wider tensors (the dense-residual sum and its data gradients,
// csrc/dense_residual.hip): the one-group case of the 256 x 256 ping-pong launch over the live windows — these
// products have 4 ... 80 steps of 64 channels, the regime that tile is built for — else the lockstep tile.
extern "C" int os2s_conv1x1_cat_fwd(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                                    const uint16_t* w, uint16_t* y, long long y_row_stride, const int32_t* in_len,
                                    const int32_t* out_len, const float* bias, int B, int T, int Cin, int Cout,
                                    int accumulate, void* workspace, size_t workspace_bytes) {
  using namespace os2s;
  OS2S_REQUIRE(x && w && y && B >= 0 && T >= 1 && Cin >= 8 && Cout >= 8 && Cin % 8 == 0 && Cout % 8 == 0);
  OS2S_REQUIRE(x_row_stride >= Cin && y_row_stride >= Cout && x_row_stride % 8 == 0 && y_row_stride % 8 == 0);
  OS2S_REQUIRE(x_row_stride < (1ll << 30) && y_row_stride < (1ll << 30));
  if (B == 0) return OS2S_OK;
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.in_len = in_len; a.out_len = out_len; a.bias = bias; a.stats = nullptr;
  a.B = B; a.Tin = T; a.Tout = T; a.Cin = Cin; a.Cout = Cout; a.K = 1; a.stride = 1; a.dil = 1; a.padL = 0;
  a.x_sb = (long long)T * x_row_stride; a.x_st = x_row_stride;
  a.y_sb = (long long)T * y_row_stride; a.y_st = y_row_stride;
  a.out_f32 = 0; a.accumulate = accumulate ? 1 : 0;
  a.act = 0; a.keep_prob = 1.f; a.seed = 0; a.residual = nullptr;
  a.mask_ref = nullptr; a.mask_scale = 1.f; a.stat_ref = nullptr;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256;
  a.force_split = -1; a.dbg = nullptr; a.dbg_fixed_w = 0;
  ConvGroupTable gt;
  gt.ngroups = 1;
  gt.g[0].x = x; gt.g[0].w = w; gt.g[0].y = y; gt.g[0].stats = nullptr;
  gt.g[0].Cin = Cin; gt.g[0].Cout = Cout; gt.g[0].accumulate = a.accumulate; gt.g[0].tile_begin = 0;
  gt.g[0].x_st = (int)x_row_stride; gt.g[0].y_st = (int)y_row_stride;
  int rc = launch_conv1x1_pp((hipStream_t)stream, a, gt);
  if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  (void)workspace; (void)workspace_bytes;
  if (Cout >= 512 && (long long)B * ceil_div(T, kConvBM) >= 256) {
    rc = launch_conv<kConvBM, 256, 2, 4, 2, false>((hipStream_t)stream, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  return launch_conv<kConvBM, kConvBN, 2, 2, 1>((hipStream_t)stream, a);
}
