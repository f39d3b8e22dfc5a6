// Weight-gradient of the channels-last 1-D convolution on the gfx950 matrix cores.
//
//   dW[k][co][ci] (+)= sum_b sum_t dY[b,t,co] * X[b, t*stride + k*dil - padL, ci]
//
// (the gradient TF derives for tf.layers.conv1d in conv_bn_actv /
//  conv_bn_res_bn_actv, open_seq2seq/parts/cnns/conv_blocks.py:195-206; X rows
//  past in_len[b] are zero because the encoder masks every conv input,
//  tdnn_encoder.py:185-186,204-205). Output is fp32 — the mixed-precision
//  wrapper casts every gradient to fp32 before anything else touches it
//  (optimizers/mp_wrapper.py:79), so we never materialise a bf16 gradient.
//
// GEMM view per tap: M = Cout, N = Cin, reduction over (b, t). Both operands
// have the reduction index (time) as their ROW index in memory ([t][c], c
// contiguous), the opposite of what an MFMA fragment wants (8 consecutive
// reduction elements per lane), so fragments are fetched with the CDNA4 LDS
// transpose read ds_read_b64_tr_b16 from row-major [64 t][128 c] LDS tiles.
//   * tiles arrive by LDS-DMA (16 B/lane); zero page for padding / masked rows;
//   * 32-byte-unit XOR swizzle (unit ^= (row & 3) << 1) on DMA source + read
//     side makes every tr read hit 8 distinct units = all 64 banks;
//   * one workgroup computes TAPS (=2) adjacent taps from ONE dY tile and one
//     (64 + dil)-row X window: halves L2->LDS bytes per FLOP;
//   * grid = (co tile, ci tile, batch split) x tap pairs; tap pairs of one tile
//     are adjacent on one XCD so dY/X tiles are L2 hits for all but the first;
//   * batch splits are combined with fp32 atomics only when a layer is too
//     small to fill the chip otherwise.
#include <stdlib.h>

#include <cstdlib>
#include "os2s_common.hpp"
#include "os2s_split_reduce.hpp"
#include <mutex>

namespace os2s {

struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  const int32_t* in_len;
  int B, Tin, Tout, Cin, Cout, K, stride, dil, padL;
  int NCO, NCI, NTP, NSPLIT, steps_per_split, use_atomic;
  int xrows, xrows_pad;  // X window rows per 64-step (and padded to x4)
  int xbuf_bytes, steptab_bytes;   // ping-pong kernel: X ring slot size, step table size
  long long x_ld;        // x row stride in elements (>= Cin; channel slice of a wider tensor)
  int accumulate;
  // ping-pong kernel: split-unit workspace (os2s_split_reduce.hpp)
  float* ws_slabs;
  int* ws_cnt;
  int ws_nslabs, ncu, force_split;
  int xcd_order;             // 1: consecutive ranks (the tap quads / ci tiles of one tile) share an XCD (wgrad_xcd_rank)
  // conv1d_wgrad_pp_kernel, grouped launch (os2s_conv1d_wgrad_grouped_ws): NG > 1 layers of ONE shape over one batch;
  // units are ranked group-major, unit -> (group, co tile, ci tile, tap quad); x / dy / dw of group g below
  int NG;
  const bf16_t* gx[8];
  const bf16_t* gdy[8];
  float* gdw[8];
  unsigned long long* dbg;   // experiment hook: slot time stamps [4 wg][2 waves][48 steps][10]
  int dbg_mode;              // experiment hook (DBG kernel): 1 no dY DMA, 2 no X DMA, 4 frozen cursor
};

// Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"; a wrong guess costs speed,
// never results). Consecutive ranks of the ping-pong / one-wave kernels are the tap quads (or ci tiles) of ONE
// (co, ci) tile: they stream the same dY / X rows at the same time, so they belong behind the same L2. Bijective on
// [0, n): XCD x owns the contiguous ranks [start(x), start(x + 1)).
__device__ __forceinline__ int wgrad_xcd_rank(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ void dma16w(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 lds_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)p);
}

// COT = output-channel extent of the block tile (128: 4 waves, 2 workgroups/CU;
// 256: 8 waves, 1 workgroup/CU, 0.74x the L2->LDS bytes per FLOP — the 128 tile runs at the
// 64 B/clk/CU L2 port limit on the big layers).
// One (co tile, ci tile, reduction split) unit x tap group of the lockstep kernel.
template <int TAPS, int COT>
__device__ __forceinline__ void wgrad_lockstep_unit(const WgradArgs& p, const int unit, const int tp, char* smem) {
  constexpr int BT = 64;  // reduction rows per step
  constexpr int NW = COT / 32, NTHR = NW * 64, NSUB = COT / 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;  // wave tile 64 (co) x 64 (ci)
  const int ysub = wm >> 1, wmi = wm & 1;  // 128-channel dY sub-image, 64-row half inside it
  const int split = unit / (p.NCO * p.NCI);
  const int rem = unit - split * (p.NCO * p.NCI);
  const int co0 = (rem / p.NCI) * COT, ci0 = (rem % p.NCI) * 128;
  const int k0 = tp * TAPS;
  const int ntaps = min(TAPS, p.K - k0);

  const int ybuf_bytes = NSUB * BT * 256;
  const int xbuf_bytes = p.xrows_pad * 256;
  char* const ybuf0 = smem;
  char* const xbuf0 = smem + 2 * ybuf_bytes;
  const char* const zero = reinterpret_cast<const char*>(g_zero_page);

  // The reduction index space is (batch item, 64-row time chunk). X rows past in_len[b] are
  // exact zeros (masked conv inputs), so chunks whose whole X window lies past the sequence
  // end contribute nothing and are not visited: only the LIVE steps are cut into NSPLIT
  // contiguous ranges (every workgroup derives the same enumeration from in_len).
  const int tchunks = (p.Tout + BT - 1) / BT;
  auto nlive = [&](int b) -> int {
    if (!p.in_len) return tchunks;
    const int l = min(max(p.in_len[b], 0), p.Tin);
    if (l <= 0) return 0;
    return min(tchunks, (l + p.padL + BT * p.stride - 1) / (BT * p.stride));
  };
  int total_live = 0;
  for (int b = 0; b < p.B; ++b) total_live += nlive(b);
  const int sps = (total_live + p.NSPLIT - 1) / p.NSPLIT;
  const int s_begin = split * sps;
  const int s_end = min(total_live, s_begin + sps);
  const int nsteps = max(0, s_end - s_begin);
  int cur_b = 0, cur_c = 0;   // cursor of the NEXT step to stage
  {
    int acc_steps = 0;
    while (cur_b < p.B && acc_steps + nlive(cur_b) <= s_begin) { acc_steps += nlive(cur_b); ++cur_b; }
    cur_c = s_begin - acc_steps;
  }

  auto stage = [&](int step, int buf) {
    const int b = min(cur_b, p.B - 1);
    const int t0 = cur_c * BT;
    ++cur_c;
    if (cur_c >= nlive(b)) {
      cur_c = 0;
      ++cur_b;
      while (cur_b < p.B && nlive(cur_b) == 0) ++cur_b;
    }
    int len_b = p.Tin;
    if (p.in_len) {
      int l = p.in_len[b];
      len_b = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
    }
    // dY tile: 64 rows x 16 pieces
    char* yd = ybuf0 + buf * ybuf_bytes;
    const bf16_t* dyb = p.dy + (long long)b * p.Tout * p.Cout;
#pragma unroll
    for (int it = 0; it < (NSUB * 16) / NW; ++it) {
      const int base = (it * NW + wid) * 64;
      const int q = base + lane;
      const int sub = q >> 10, row = (q & 1023) >> 4, ps = q & 15;
      const int u = (ps >> 1) ^ ((row & 3) << 1);
      const int ch = co0 + sub * 128 + ((u << 1) | (ps & 1)) * 8;
      const int t = t0 + row;
      const bool ok = (t < p.Tout) && (ch < p.Cout);
      const void* src = ok ? (const void*)(dyb + (long long)t * p.Cout + ch)
                           : (const void*)(zero + ps * 16);
      dma16w(src, yd + base * 16);
    }
    // X window: xrows_pad rows x 16 pieces; LDS row r <-> input time tin0 + r
    char* xd = xbuf0 + buf * xbuf_bytes;
    const bf16_t* xb = p.x + (long long)b * p.Tin * p.x_ld;
    const int tin0 = t0 * p.stride + k0 * p.dil - p.padL;
    const int npieces = p.xrows_pad * 16;
    for (int base = wid * 64; base < npieces; base += NTHR) {
      const int q = base + lane;
      const int row = q >> 4, ps = q & 15;
      const int u = (ps >> 1) ^ ((row & 3) << 1);
      const int ch = ci0 + ((u << 1) | (ps & 1)) * 8;
      const int tin = tin0 + row;
      const bool ok = (row < p.xrows) && (tin >= 0) && (tin < len_b) && (ch < p.Cin);
      const void* src = ok ? (const void*)(xb + (long long)tin * p.x_ld + ch)
                           : (const void*)(zero + ps * 16);
      dma16w(src, xd + base * 16);
    }
  };

  f32x16 acc[TAPS][2][2];
#pragma unroll
  for (int a = 0; a < TAPS; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;

  // per-lane constants of the transpose reads
  const int g16 = (lane >> 4) & 1;   // which 16-column block of the 32-wide MFMA tile
  const int i16 = lane & 15;
  const int rsub = i16 >> 2;         // row within the 4-row block this lane addresses
  const int csub = (i16 & 3) * 8;    // byte offset of its 4-element chunk
  const int lhi = lane >> 5;

  if (nsteps > 0) stage(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (step + 1 < nsteps) stage(step + 1, (step + 1) & 1);
    const char* const ys = ybuf0 + (step & 1) * ybuf_bytes + ysub * (BT * 256);
    const char* const xs = xbuf0 + (step & 1) * xbuf_bytes;
    // Software pipeline inside the wave: the transpose reads of k-slice kk+1 are issued before
    // the MFMAs of slice kk (the compiler otherwise waits for each group of reads right before
    // the MFMAs that consume it, leaving the matrix pipe idle for an LDS round trip per group).
    struct Frags { bf16x8 af[2]; bf16x8 bfr[TAPS][2]; };
    auto load_frags = [&](int kk) {
      Frags f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = ((wmi * 64 + i * 32) >> 4) + g16;  // logical 32-B unit (16 channels)
        const int r0 = kk * 16 + lhi * 8 + rsub;
        const int r1 = r0 + 4;
        const bf16x4 lo = lds_tr(ys + r0 * 256 + ((u ^ ((r0 & 3) << 1)) << 5) + csub);
        const bf16x4 hi = lds_tr(ys + r1 * 256 + ((u ^ ((r1 & 3) << 1)) << 5) + csub);
        f.af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < TAPS; ++a)   // a tap past K (last group of an odd K) reads valid LDS rows; its tile is not stored
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int u = ((wn * 64 + j * 32) >> 4) + g16;
          const int r0 = (kk * 16 + lhi * 8 + rsub) * p.stride + a * p.dil;
          const int r1 = r0 + 4 * p.stride;
          const bf16x4 lo = lds_tr(xs + r0 * 256 + ((u ^ ((r0 & 3) << 1)) << 5) + csub);
          const bf16x4 hi = lds_tr(xs + r1 * 256 + ((u ^ ((r1 & 3) << 1)) << 5) + csub);
          f.bfr[a][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      return f;
    };
    Frags cur = load_frags(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      Frags nxt;
      if (kk < 3) nxt = load_frags(kk + 1);
#pragma unroll
      for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[a][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.af[i], cur.bfr[a][j], acc[a][i][j], 0, 0, 0);
      if (kk < 3) cur = nxt;
    }
  }

  // ---- epilogue: fp32 stores / atomics, ci contiguous across lanes ---------
  const int l31 = lane & 31;
#pragma unroll
  for (int a = 0; a < TAPS; ++a) {
    if (a < ntaps) {
      float* const dwk = p.dw + (long long)(k0 + a) * p.Cout * p.Cin;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ci = ci0 + wn * 64 + j * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            if (co < p.Cout && ci < p.Cin) {
              float* dst = dwk + (long long)co * p.Cin + ci;
              if (p.use_atomic)
                __hip_atomic_fetch_add(dst, acc[a][i][j][e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
              else
                *dst = acc[a][i][j][e];
            }
          }
        }
    }
  }
}

template <int TAPS, int COT>
__global__ __launch_bounds__(COT * 2, (COT == 128) ? 2 : 1) void conv1d_wgrad_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int unit = (loc / p.NTP) * 8 + xcd;
  const int tp = loc - (loc / p.NTP) * p.NTP;
  if (unit >= p.NCO * p.NCI * p.NSPLIT) return;
  wgrad_lockstep_unit<TAPS, COT>(p, unit, tp, smem);
}

// Up to kMaxWgradGroups independent K = 1 weight gradients over the same ragged batch (B, T,
// in_len) in ONE grid: the dense-residual 1x1 branches of a Jasper block end (conv_blocks.py:78-85:
// up to 10 branches of 2-36 output tiles each). One launch per branch put 12-36 tiles on 256 CUs
// and cut the reduction 14-40 ways to fill the chip, every piece ending in fp32 atomics on the
// same dW; together the branches of a block are 50-216 tiles, so the reduction is cut a few ways
// at most (not at all for the wide blocks: one owner per dW element, deterministic).
constexpr int kMaxWgradGroups = 16;
struct WgradGroup {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  long long x_ld;
  int Cin, Cout, NCI, unit_begin;
};
struct WgradGroupTable {
  int ngroups, total_units;
  WgradGroup g[kMaxWgradGroups];
};

__global__ __launch_bounds__(256, 2) void conv1d_wgrad_grouped_kernel(WgradArgs p, WgradGroupTable gt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  if (bid >= gt.total_units) return;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < kMaxWgradGroups; ++i)
    if (i < gt.ngroups && bid >= gt.g[i].unit_begin) gi = i;
  WgradGroup g = gt.g[0];
#pragma unroll
  for (int i = 1; i < kMaxWgradGroups; ++i)
    if (i == gi) g = gt.g[i];
  p.x = g.x; p.dy = g.dy; p.dw = g.dw; p.x_ld = g.x_ld;
  p.Cin = g.Cin; p.Cout = g.Cout;
  p.NCO = (g.Cout + 127) / 128; p.NCI = g.NCI;
  wgrad_lockstep_unit<1, 128>(p, bid - g.unit_begin, 0, smem);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong weight-gradient kernel (stride 1, K >= 2): same slot structure as conv1d_pp_kernel.
//
// Workgroup tile: 128 output channels x 128 input channels x FOUR adjacent taps, 8 waves, from
// one dY tile [64 t][128 co] and one (64 + 3 dil)-row X window [t][128 ci] per reduction step
// (34 KB of LDS-DMA per 32 MFMAs per wave — the DMA issue of the loading wave, ~75 cycles per
// 1-KB instruction, is what bounds the odd slot; a 256 x 128 x 2-tap tile needs 49 KB).
// Waves 0-3 ("group A") accumulate taps k0, k0+1, waves 4-7 ("group B") taps k0+2, k0+3. A wave
// owns 64 co x 64 ci of each of its two taps (128 accumulator registers); an item is one 64-row
// reduction step of one tap = 16 MFMA 32x32x16. Wave w and wave w+4 share a SIMD: in every
// barrier-delimited slot one of them issues its 16 MFMAs while the other fetches fragments
// (transpose reads, ds_read_b64_tr_b16: the reduction index is the ROW index of both operands
// in memory) and issues the LDS-DMA of the step two ahead.
//
//   LOAD(2s)   : dY fragments of step s (kept for both taps) + X fragments of the first tap
//   LOAD(2s+1) : X fragments of the second tap; drain the DMA issued one step ago; issue step
//                s+2 (X into a ring of 3 — the other group still reads the X window of step s in
//                ITS odd slot — dY into a ring of 2: dY is only read in even slots)
//
// The reduction runs over the LIVE 64-row chunks of the ragged batch only (rows past in_len are
// zero conv inputs). Units (co tile, ci tile, tap quad) that do not fill the last round of
// workgroups are cut f ways along the reduction and reduced deterministically by the last
// arriver (os2s_split_reduce.hpp): no fp32 atomics, dW written once by one owner.
// ---------------------------------------------------------------------------------------------
// Transpose read as inline assembly (32-bit LDS byte address + immediate offset). hipcc (ROCm
// 7.2) drains vmcnt(0) in front of the first LDS read it can see after an LDS-DMA issue, which
// here would wait for the tiles of step s+2 one COMPUTE slot after they were requested; these
// reads only ever touch tiles whose DMA was drained explicitly (vmcnt(0) + barrier, above).
// The compiler does not count them in lgkmcnt either: every LOAD slot ends with an explicit
// s_waitcnt lgkmcnt(0) followed by a scheduling barrier before the first use.
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr_asm(unsigned addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// one bf16x8 MFMA operand = rows r0 and r0+4 (1024 B apart) of k-slice KK (4096 B apart)
template <int KK>
__device__ __forceinline__ bf16x8 lds_frag(unsigned addr) {
  const bf16x4 lo = lds_tr_asm<KK * 4096>(addr);
  const bf16x4 hi = lds_tr_asm<KK * 4096 + 1024>(addr);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void wpp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int kWppTaps = 4;
// descriptor words + X row offset of one staged step (conv1d_wgrad_pp_kernel: prep / issue)
struct WppStaged {
  unsigned ylo, yhi, ynr, xlo, xhi, xnr;
  int xro;
};

// DBG: per-slot s_memtime stamps of waves 0 and 4 of workgroups 0..3 (tools/pp_timeline.py)
template <bool DBG>
__global__ __launch_bounds__(512, 2) void conv1d_wgrad_pp_kernel(WgradArgs p) {
  constexpr int BT = 64, COT = 128, CIT = 128;
  constexpr int YBUF = BT * 256;                           // dY tile [64 rows][128 ch]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3;
  const int wm = wq >> 1, wn = wq & 1;                     // co 64-half, ci 64-half

  // ---- live 64-row chunks per sample (one value per lane, B <= 64), inclusive scan -----------
  const int tchunks = (p.Tout + BT - 1) / BT;
  int nl = 0, len_l = 0;
  if (lane < p.B) {
    len_l = p.Tin;
    if (p.in_len) {
      const int l = p.in_len[lane];
      len_l = l < 0 ? 0 : (l < p.Tin ? l : p.Tin);
    }
    nl = len_l > 0 ? (len_l + p.padL + BT - 1) / BT : 0;
    nl = nl < tchunks ? nl : tchunks;
  }
  int scan = nl;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(scan, o, 64);
    if (lane >= o) scan += t;
  }
  const int total_live = __builtin_amdgcn_readlane(scan, 63);

  // ---- block -> (unit, piece) -------------------------------------------------------------
  const int U = p.NG * p.NCO * p.NCI * p.NTP, G = p.ncu;
  const int q = U / G, r = U - q * G;
  int f = 1;
  {
    int fmax = total_live / 8;                             // >= 8 steps per piece
    fmax = fmax > 16 ? 16 : fmax;
    if (p.force_split > 0 && p.ws_slabs) {
      f = p.force_split < fmax ? p.force_split : (fmax > 1 ? fmax : 1);
      while (f > 1 && r * f > p.ws_nslabs) --f;
      if (r == 0) f = 1;
    } else if (r > 0 && p.ws_slabs) {
      f = split_factor(r, G, 1.1f * total_live, fmax, p.ws_nslabs);
    }
  }
  const int nfull = f > 1 ? U - r : U;
  const int nwork = nfull + (f > 1 ? r * f : 0);
  if ((int)blockIdx.x >= nwork) return;
  const int bid = p.xcd_order ? wgrad_xcd_rank(blockIdx.x, nwork) : (int)blockIdx.x;
  int rank = bid, piece = 0, npiece = 1;
  if (bid >= nfull) {
    const int i = bid - nfull;
    rank = nfull + i / f;
    piece = i - (i / f) * f;
    npiece = f;
  }
  // tap quads of one (co, ci) tile are adjacent ranks: they stream the same dY / X rows
  const int tp = rank % p.NTP;
  int rem = rank / p.NTP;
  if (p.NG > 1) {                                          // grouped launch: same-shape layers, group-major ranks
    const int per = p.NCO * p.NCI;
    const int g = __builtin_amdgcn_readfirstlane(rem / per);    // wave-uniform: scalar loads from the arguments
    rem -= g * per;
    // (selected by compare: a runtime index would move the argument table to private memory)
    const bf16_t* sx = p.gx[0]; const bf16_t* sdy = p.gdy[0]; float* sdw = p.gdw[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if (i == g) { sx = p.gx[i]; sdy = p.gdy[i]; sdw = p.gdw[i]; }
    p.x = sx; p.dy = sdy; p.dw = sdw;
  }
  const int co0 = (rem / p.NCI) * COT, ci0 = (rem % p.NCI) * CIT;
  const int k0 = tp * kWppTaps;
  const int sps = (total_live + npiece - 1) / npiece;
  const int s_begin = __builtin_amdgcn_readfirstlane(min(piece * sps, total_live));
  const int s_end = __builtin_amdgcn_readfirstlane(min(total_live, s_begin + sps));
  const int nsteps = s_end - s_begin;

  // LDS: dY ring (2 x 16 KB) | X ring (3 x xbuf_bytes, each rounded up to whole 8-wave DMA
  // rounds so that no wave needs a predicate for its first two X instructions) | step table
  const int xbuf_bytes = p.xbuf_bytes;
  char* const ybuf0 = smem;
  char* const xbuf0 = smem + 2 * YBUF;
  int* const steptab = reinterpret_cast<int*>(xbuf0 + 3 * xbuf_bytes);

  // ---- step table: entry i = (sample, 64-row chunk, in_len) of reduction step s_begin + i ------
  // built once by the lanes that own a sample; the loop then needs no cursor logic, no branch
  // and no global scalar load: one LDS dword per step, read a step ahead.
  {
    const int excl = scan - nl;
    for (int c = 0; c < tchunks; ++c) {
      const int idx = excl + c - s_begin;
      if (lane < p.B && c < nl && idx >= 0 && idx < nsteps && wid == 0)
        steptab[idx] = lane | (c << 8) | (len_l << 16);
    }
  }

  // ---- DMA: per-lane byte offsets are fixed for the kernel ------------------------------------
  // dY tile: 64 rows x 16 pieces of 16 B = 16 instructions, 2 per wave. LDS piece q of a 256-B
  // row holds the channel block given by the 32-B-unit XOR swizzle (as the tr reads expect).
  int yv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int qq = (it * 8 + wid) * 64 + lane;
    const int row = qq >> 4, ps = qq & 15;
    const int u = (ps >> 1) ^ ((row & 3) << 1);
    const int ch = co0 + ((u << 1) | (ps & 1)) * 8;
    yv[it] = ch < p.Cout ? (row * p.Cout + ch) * 2 : (int)0x80000000;
  }
  // X window: xrows rows x 16 pieces, 3 instructions per wave (24 x 4 rows >= 64 + 3 dil rows);
  // lanes past the window / past Cin carry an offset that stays out of range after the row
  // offset of the step is added (|row offset| < 2^30, see the launcher)
  int xv[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int qq = (n * 8 + wid) * 64 + lane;
    const int row = qq >> 4, ps = qq & 15;
    const int u = (ps >> 1) ^ ((row & 3) << 1);
    const int ch = ci0 + ((u << 1) | (ps & 1)) * 8;
    xv[n] = (ch < p.Cin && row < p.xrows) ? (row * (int)p.x_ld + ch) * 2 : (int)0x80000000;
  }
  const bool x3 = 16 + wid < ((p.xrows * 16 + 63) >> 6);   // does this wave own a third X instruction
  const unsigned long long dy_base = (unsigned long long)p.dy, x_base = (unsigned long long)p.x;
  const int ycol_bytes = __builtin_amdgcn_readfirstlane(p.Cout * 2);
  const int xcol_bytes = __builtin_amdgcn_readfirstlane((int)p.x_ld * 2);
  const unsigned long long xsample_bytes = (unsigned long long)p.Tin * (unsigned long long)p.x_ld * 2ull;
  // Staging the tiles of the step described by table entry `ent` is two things: `prep` — the two buffer
  // descriptors and the row offset of the X window, ~30 scalar instructions — and `issue`, the five LDS-DMA
  // instructions themselves. In the loop `prep` for step s+2 runs INSIDE the COMPUTE slot of step s, between
  // its MFMAs (round 5: tools/pp_timeline.py showed the LOAD slot that carried both at 908 cycles against 598
  // of the COMPUTE slot it is paired with — the partner waited 300 cycles per step at the barrier).
  // (the words are pinned to scalar registers where they are computed — an empty asm — or the compiler sinks
  // the whole computation back to its first use in the LOAD slot)
  auto prep = [&](int ent) __attribute__((always_inline)) -> WppStaged {
    const int b = ent & 0xff, t0 = ((ent >> 8) & 0xff) * BT, len_b = (int)((unsigned)ent >> 16);
    WppStaged st;
    // dY rows t0.. of sample b; num_records up to the sample end (rows >= Tout read as zeros)
    const unsigned long long yb = dy_base + (unsigned long long)(unsigned)(b * p.Tout + t0) * (unsigned)ycol_bytes;
    st.ylo = (unsigned)yb; st.yhi = (unsigned)(yb >> 32);
    st.ynr = (unsigned)((p.Tout - t0) * ycol_bytes);
    // X rows of sample b, num_records = in_len rows: rows < 0 (a negative offset) and rows >=
    // in_len read as zeros
    const unsigned long long xb = x_base + (unsigned long long)(unsigned)b * xsample_bytes;
    st.xlo = (unsigned)xb; st.xhi = (unsigned)(xb >> 32);
    st.xnr = (unsigned)(len_b * xcol_bytes);
    st.xro = (t0 + k0 * p.dil - p.padL) * xcol_bytes;
#if defined(__HIP_DEVICE_COMPILE__)      // ("s" is a scalar-register constraint of the device pass only)
    asm volatile("" : "+s"(st.ylo), "+s"(st.yhi), "+s"(st.ynr), "+s"(st.xlo), "+s"(st.xhi), "+s"(st.xnr), "+s"(st.xro));
#endif
    return st;
  };
  auto issue = [&](const WppStaged& st, int ybuf_idx, int xbuf_idx) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)st.yhi << 32) | st.ylo), 0, (int)st.ynr, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)st.xhi << 32) | st.xlo), 0, (int)st.xnr, 0x00020000);
    char* const yd = ybuf0 + ybuf_idx * YBUF;
    if (!(DBG && (p.dbg_mode & 1)))
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          yrs, (__attribute__((address_space(3))) void*)(yd + (it * 8 + wid) * 1024), 16, yv[it], 0, 0, 0);
    char* const xd = xbuf0 + xbuf_idx * xbuf_bytes;
    if (!(DBG && (p.dbg_mode & 2))) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            xrs, (__attribute__((address_space(3))) void*)(xd + (n * 8 + wid) * 1024), 16,
            xv[n] + st.xro, 0, 0, 0);
      if (x3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            xrs, (__attribute__((address_space(3))) void*)(xd + (16 + wid) * 1024), 16, xv[2] + st.xro, 0, 0, 0);
    }
  };
  auto stage = [&](int ent, int ybuf_idx, int xbuf_idx) __attribute__((always_inline)) {
    issue(prep(ent), ybuf_idx, xbuf_idx);
  };

  // taps past K (the last tap quad of K = 4n + 1 ... 4n + 3: every Jasper layer has K = 4n + 1) are
  // not computed: their fragment reads and MFMAs are skipped (wave-uniform), the wave keeps its DMA
  // duties and its barriers. A quad with one live tap then costs about a third less than a full one.
  const bool live0 = k0 + 2 * grp < p.K, live1 = k0 + 2 * grp + 1 < p.K;
  f32x16 acc[2][2][2];                                     // [tap of the pair][i][j]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[e][i][j][v] = 0.f;

  const int lhi = lane >> 5;
  if (nsteps > 0) {
    // per-lane constants of the transpose reads (relative to the tile base)
    const int g16 = (lane >> 4) & 1;   // which 16-column block of the 32-wide MFMA tile
    const int i16 = lane & 15;
    const int rsub = i16 >> 2;         // row within the 4-row block this lane addresses
    const int csub = (i16 & 3) * 8;    // byte offset of its 4-element chunk
    int ya[2], xa[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 4 * wm + 2 * i + g16;
      ya[i] = (lhi * 8 + rsub) * 256 + ((u ^ ((rsub & 3) << 1)) << 5) + csub;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r0 = lhi * 8 + rsub + (2 * grp + e) * p.dil;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = 4 * wn + 2 * j + g16;
        xa[e][j] = r0 * 256 + ((u ^ ((r0 & 3) << 1)) << 5) + csub;
      }
    }
    __syncthreads();                                       // step table complete
    stage(__builtin_amdgcn_readfirstlane(steptab[0]), 0, 0);
    if (nsteps > 1) stage(__builtin_amdgcn_readfirstlane(steptab[1]), 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp) wpp_barrier();                                // group B runs one slot behind group A

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned long long* const tl = reinterpret_cast<unsigned long long*>(xbuf0 + 3 * xbuf_bytes + p.steptab_bytes);
    const unsigned tab0 = lds0 + 2 * YBUF + 3 * xbuf_bytes;
    const bool rec = DBG && p.dbg && bid < 4 && lane == 0 && (wid & 3) == 0;
    auto stamp = [&](int s, int i) {
      if (DBG && rec && s < 48) tl[(grp * 48 + s) * 10 + i] = __builtin_readcyclecounter();
    };
    int xi = 0;                                            // ring slot of the current step's X
    for (int s = 0; s < nsteps; ++s) {
      const unsigned ys = lds0 + (s & 1) * YBUF;
      const unsigned xs = lds0 + 2 * YBUF + xi * xbuf_bytes;
      bf16x8 xf[2][4], yf[2][4];
      // ---- LOAD(2s): also the table entry of step s+2 (consumed in the odd slot)
      int ent_v;
      asm volatile("ds_read_b32 %0, %1" : "=v"(ent_v) : "v"(tab0 + (s + 2 < nsteps ? s + 2 : s) * 4) : "memory");
      if (live0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          yf[i][0] = lds_frag<0>(ys + ya[i]); yf[i][1] = lds_frag<1>(ys + ya[i]);
          yf[i][2] = lds_frag<2>(ys + ya[i]); yf[i][3] = lds_frag<3>(ys + ya[i]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xf[j][0] = lds_frag<0>(xs + xa[0][j]); xf[j][1] = lds_frag<1>(xs + xa[0][j]);
          xf[j][2] = lds_frag<2>(xs + xa[0][j]); xf[j][3] = lds_frag<3>(xs + xa[0][j]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(s, 0);
      wpp_barrier();
      stamp(s, 1);
      // ---- COMPUTE(2s): the MFMAs and, between them, the descriptors of step s+2 (its table entry landed
      //      with the lgkmcnt(0) that closed the LOAD slot)
      WppStaged nxt;
      if (live0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][kk], xf[j][kk], acc[0][i][j], 0, 0, 0);
        nxt = prep(__builtin_amdgcn_readfirstlane(ent_v));
        // one MFMA, then up to three scalar / one vector instruction of the preparation, and so on
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
      } else {
        nxt = prep(__builtin_amdgcn_readfirstlane(ent_v));
      }
      if (DBG) __builtin_amdgcn_sched_barrier(0);
      stamp(s, 2);
      wpp_barrier();
      stamp(s, 3);
      // ---- LOAD(2s+1): X fragments of the second tap; then drain the DMA issued one step ago
      //      and issue step s+2 (X slot (s+2) % 3 was last read one step ago, the dY slot in the
      //      even slots of this step)
      if (live1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xf[j][0] = lds_frag<0>(xs + xa[1][j]); xf[j][1] = lds_frag<1>(xs + xa[1][j]);
          xf[j][2] = lds_frag<2>(xs + xa[1][j]); xf[j][3] = lds_frag<3>(xs + xa[1][j]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(s, 4); }
      {
        int x2 = xi + 2;
        x2 = x2 >= 3 ? x2 - 3 : x2;
        if (s + 2 < nsteps) issue(nxt, s & 1, x2);
        xi = xi + 1 >= 3 ? 0 : xi + 1;
      }
      stamp(s, 5);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(s, 6);
      wpp_barrier();
      stamp(s, 7);
      // ---- COMPUTE(2s+1)
      if (live1) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[1][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][kk], xf[j][kk], acc[1][i][j], 0, 0, 0);
      }
      stamp(s, 8);
      wpp_barrier();
      stamp(s, 9);
    }
    if (!grp) wpp_barrier();
    if (DBG && p.dbg && bid < 4) {
      __syncthreads();
      for (int i = tid; i < 2 * 48 * 10; i += 512) p.dbg[bid * 2 * 48 * 10 + i] = tl[i];
    }
  }
  __syncthreads();

  if (npiece > 1) {
    const int sidx = rank - nfull;
    auto at = [&](int v) -> f32x16& { return acc[v >> 2][(v >> 1) & 1][v & 1]; };
    if (!split_publish_and_reduce(at, p.ws_slabs + (size_t)sidx * f * kSplitSlabFloats,
                                  p.ws_cnt + sidx, piece, f, smem, tid))
      return;
  }

  // ---- epilogue: the one owner of the tile writes dW (fp32), ci contiguous across lanes -------
  // accumulate: ALL previous values of a tap's 64 elements are loaded first, then added and stored
  // (written element by element the compiler must keep load -> wait -> store order because the
  // stores may alias the next load: 128 serial memory round trips per lane)
  // Tiles completely inside [Cout, Cin] (every Jasper layer) take a branch-free path: with a bounds
  // test around each store every store sits in its own basic block and gets a vmcnt(0) in front.
  const int l31 = lane & 31;
  const bool inside = __builtin_amdgcn_readfirstlane((co0 + 128 <= p.Cout && ci0 + 128 <= p.Cin) ? 1 : 0);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int tap = k0 + 2 * grp + e;
    if (tap < p.K) {
      float* const dwk = p.dw + (long long)tap * p.Cout * p.Cin;
      if (inside) {
        float* const base = dwk + (long long)(co0 + wm * 64 + 4 * lhi) * p.Cin + ci0 + wn * 64 + l31;
        float old[2][2][16];
        if (p.accumulate) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int v = 0; v < 16; ++v)
                old[i][j][v] = base[(long long)(i * 32 + (v & 3) + 8 * (v >> 2)) * p.Cin + j * 32];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int v = 0; v < 16; ++v) acc[e][i][j][v] += old[i][j][v];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
              base[(long long)(i * 32 + (v & 3) + 8 * (v >> 2)) * p.Cin + j * 32] = acc[e][i][j][v];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const int co = co0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lhi;
              if (co < p.Cout && ci < p.Cin) {
                float* dst = dwk + (long long)co * p.Cin + ci;
                float val = acc[e][i][j][v];
                if (p.accumulate) val += *dst;
                *dst = val;
              }
            }
          }
      }
    }
  }
}

#include "conv1d_wgrad_sw.hpp"

// ---------------------------------------------------------------------------------------------
// Ping-pong weight gradient of the K = 1 convolution = the weight gradient of tf.layers.Dense
// (Transformer projections, attention_layer.py:54-62 / ffn_layer.py:51-85 / embedding_layer.py:
// 90-105; the FC and 1x1 layers of the other models):
//
//   dW[co][ci] (+)= sum_b sum_{t < len_b} dY[b,t,co] * X[b,t,ci]
//
// a "TN" GEMM: the reduction index is the ROW index of both operands, so both fragments come
// from transpose reads as in conv1d_wgrad_pp_kernel. With one tap there is no X-window reuse;
// the tile is therefore the plain GEMM's 256 co x 256 ci (64 KB of LDS-DMA per 64-row step for
// 32 MFMAs per wave, the ratio of gemm_pp.hip) and the slot / ring structure is gemm_pp's:
//
//   group A (waves 0-3) owns co rows 0..127 of the tile, group B (waves 4-7) rows 128..255; a
//   wave owns 128 co x 64 ci (128 accumulator registers); an item = one 64-co half = 16 MFMA.
//   LOAD(2s)   : X fragments of step s (kept for both items) + dY fragments of the first item;
//                issue the group's OWN dY half [64 t][128 co] of step s+2 (ring of 3)
//   LOAD(2s+1) : dY fragments of the second item; drain all DMA but the 4 just issued (counted
//                vmcnt); issue the X tile [64 t][256 ci] of step s+2 (ring of 2: X is read in
//                even slots only)
//
// Every LOAD slot carries exactly 4 LDS-DMA instructions per wave. The reduction runs over the
// live 64-row chunks of the ragged batch; the (sample, chunk) of a step is derived from the
// per-lane scan of live chunks with one ballot (no table: the 160 KB of LDS are the two rings).
// Rows at or past len_b are outside the step's buffer descriptors (both operands) and read as
// zeros. The output is small against the reduction (a 1024 x 1024 Dense kernel is 16 tiles for
// 130 steps), so the units of the last partial round are cut up to 16 ways along the reduction
// and reduced by the last arriver (os2s_split_reduce.hpp): deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
// gt.ngroups > 0: up to kMaxWgradGroups independent problems over the same rows (B, T, in_len) in
// one grid — units are ranked over all groups (group g owns ranks [unit_begin, unit_begin + NCO*NCI)),
// p.NCO * p.NCI = the total: three 1024 x 1024 Dense weight gradients are 48 tiles cut 5 ways instead
// of three launches of 16 tiles cut 16 ways (or the lockstep kernel with its atomics).
__global__ __launch_bounds__(512, 2) void conv1d_wgrad1x1_pp_kernel(WgradArgs p, WgradGroupTable gt) {
  constexpr int BT = 64;
  constexpr int TILE = BT * 256 * 2;                       // [64 rows][256 ch] bf16 = 32 KB
  constexpr int HALF = TILE / 2;                           // one [64][128] sub-image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;

  // ---- live 64-row chunks per sample (one value per lane, B <= 64), inclusive scan -----------
  const int tchunks = (p.Tout + BT - 1) / BT;
  int nl = 0, len_l = 0;
  if (lane < p.B) {
    len_l = p.Tout;
    if (p.in_len) {
      const int l = p.in_len[lane];
      len_l = l < 0 ? 0 : (l < p.Tout ? l : p.Tout);
    }
    nl = (len_l + BT - 1) / BT;
    nl = nl < tchunks ? nl : tchunks;
  }
  int scan = nl;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(scan, o, 64);
    if (lane >= o) scan += t;
  }
  const int total_live = __builtin_amdgcn_readlane(scan, 63);

  // ---- block -> (unit, piece): as conv1d_wgrad_pp_kernel ----------------------------------------
  const int U = p.NCO * p.NCI, G = p.ncu;
  const int q = U / G, r = U - q * G;
  int f = 1;
  {
    int fmax = total_live / 8;                             // >= 8 steps per piece
    fmax = fmax > 16 ? 16 : fmax;
    if (p.force_split > 0 && p.ws_slabs) {
      f = p.force_split < fmax ? p.force_split : (fmax > 1 ? fmax : 1);
      while (f > 1 && r * f > p.ws_nslabs) --f;
      if (r == 0) f = 1;
    } else if (r > 0 && p.ws_slabs) {
      f = split_factor(r, G, 1.1f * total_live, fmax, p.ws_nslabs);
    }
  }
  const int nfull = f > 1 ? U - r : U;
  const int nwork = nfull + (f > 1 ? r * f : 0);
  if ((int)blockIdx.x >= nwork) return;
  const int bid = p.xcd_order ? wgrad_xcd_rank(blockIdx.x, nwork) : (int)blockIdx.x;
  int rank = bid, piece = 0, npiece = 1;
  if (bid >= nfull) {
    const int i = bid - nfull;
    rank = nfull + i / f;
    piece = i - (i / f) * f;
    npiece = f;
  }
  int grank = rank;                                       // rank inside its group
  if (gt.ngroups > 0) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < kMaxWgradGroups; ++i)
      if (i < gt.ngroups && rank >= gt.g[i].unit_begin) gi = i;
    WgradGroup g = gt.g[0];
#pragma unroll
    for (int i = 1; i < kMaxWgradGroups; ++i)
      if (i == gi) g = gt.g[i];
    p.x = g.x; p.dy = g.dy; p.dw = g.dw; p.x_ld = g.x_ld;
    p.Cin = g.Cin; p.Cout = g.Cout; p.NCI = g.NCI;
    grank = rank - g.unit_begin;
  }
  // the ci tiles of one co tile are adjacent ranks: they stream the same dY columns together
  const int co0 = (grank / p.NCI) * 256, ci0 = (grank % p.NCI) * 256;
  const int sps = (total_live + npiece - 1) / npiece;
  const int s_begin = __builtin_amdgcn_readfirstlane(min(piece * sps, total_live));
  const int s_end = __builtin_amdgcn_readfirstlane(min(total_live, s_begin + sps));
  const int nsteps = s_end - s_begin;

  // ---- DMA: per-lane byte offsets are fixed for the kernel --------------------------------------
  // an instruction moves 4 rows x 256 B of a [64][128] sub-image; LDS piece ps of a row holds the
  // channel block given by the 32-B-unit XOR swizzle (as the transpose reads expect)
  int yv[4], xv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    {
      const int qq = (it * 4 + wn) * 64 + lane;
      const int row = qq >> 4, ps = qq & 15;
      const int u = (ps >> 1) ^ ((row & 3) << 1);
      const int ch = co0 + grp * 128 + ((u << 1) | (ps & 1)) * 8;
      yv[it] = ch < p.Cout ? (row * p.Cout + ch) * 2 : (int)0x80000000;
    }
    {
      const int idx = it * 8 + wid;
      const int qq = (idx & 15) * 64 + lane;
      const int row = qq >> 4, ps = qq & 15;
      const int u = (ps >> 1) ^ ((row & 3) << 1);
      const int ch = ci0 + (idx >> 4) * 128 + ((u << 1) | (ps & 1)) * 8;
      xv[it] = ch < p.Cin ? (row * (int)p.x_ld + ch) * 2 : (int)0x80000000;
    }
  }
  const unsigned long long dy_base = (unsigned long long)p.dy, x_base = (unsigned long long)p.x;
  const int ycol_bytes = __builtin_amdgcn_readfirstlane(p.Cout * 2);
  const int xcol_bytes = __builtin_amdgcn_readfirstlane((int)p.x_ld * 2);
  // (sample, first row, length) of live step i: the samples whose inclusive scan is <= i lie before it
  auto entry = [&](int i, int& b, int& t0, int& len_b) {
    const unsigned long long m = __ballot(scan <= i);
    int bb = __builtin_popcountll(m);
    bb = bb < p.B ? bb : p.B - 1;
    b = __builtin_amdgcn_readfirstlane(bb);
    const int before = b > 0 ? __builtin_amdgcn_readlane(scan, b - 1) : 0;
    t0 = (i - before) * BT;
    len_b = __builtin_amdgcn_readlane(len_l, b);
  };
  auto stage_y = [&](int b, int t0, int len_b, int slot) {   // this group's dY half
    int rows = len_b - t0;
    rows = rows < 0 ? 0 : (rows > BT ? BT : rows);
    rows = __builtin_amdgcn_readfirstlane(rows);          // (a clamp compiles to v_med3: keep the descriptor scalar)
    const unsigned long long base =
        dy_base + ((unsigned long long)(unsigned)b * (unsigned)p.Tout + (unsigned)t0) * (unsigned)ycol_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * ycol_bytes, 0x00020000);
    char* const dst = smem + slot * TILE + grp * HALF;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs, (__attribute__((address_space(3))) void*)(dst + (it * 4 + wn) * 1024), 16, yv[it], 0, 0, 0);
  };
  auto stage_x = [&](int b, int t0, int len_b, int slot) {
    int rows = len_b - t0;
    rows = rows < 0 ? 0 : (rows > BT ? BT : rows);
    rows = __builtin_amdgcn_readfirstlane(rows);          // (a clamp compiles to v_med3: keep the descriptor scalar)
    const unsigned long long base =
        x_base + ((unsigned long long)(unsigned)b * (unsigned)p.Tin + (unsigned)t0) * (unsigned)xcol_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * xcol_bytes, 0x00020000);
    char* const dst = smem + 3 * TILE + slot * TILE;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs, (__attribute__((address_space(3))) void*)(dst + (it * 8 + wid) * 1024), 16, xv[it], 0, 0, 0);
  };

  f32x16 acc[2][2][2];                                     // [co half of the group][i][j]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[e][i][j][v] = 0.f;

  const int lhi = lane >> 5;
  if (nsteps > 0) {
    // per-lane constants of the transpose reads (relative to the sub-image base)
    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    const int rsub = i16 >> 2, csub = (i16 & 3) * 8;
    int ya[2][2], xa[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = 4 * e + 2 * i + g16;
        ya[e][i] = grp * HALF + (lhi * 8 + rsub) * 256 + ((u ^ ((rsub & 3) << 1)) << 5) + csub;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u = 4 * (wn & 1) + 2 * j + g16;
      xa[j] = (wn >> 1) * HALF + (lhi * 8 + rsub) * 256 + ((u ^ ((rsub & 3) << 1)) << 5) + csub;
    }
    {
      int b, t0, len_b;
      entry(s_begin, b, t0, len_b);
      stage_y(b, t0, len_b, 0);
      stage_x(b, t0, len_b, 0);
      // only step 0 has to land before the loop starts; the 8 instructions of step 1 are drained by
      // the counted wait of LOAD(1)
      if (nsteps > 1) {
        entry(s_begin + 1, b, t0, len_b);
        stage_y(b, t0, len_b, 1);
        stage_x(b, t0, len_b, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    wpp_barrier();                                         // (__syncthreads() would drain vmcnt(0) again)
    if (grp) wpp_barrier();                                // group B runs one slot behind group A

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    int yi = 0;                                            // ring slot of the current step's dY
    for (int s = 0; s < nsteps; ++s) {
      const unsigned ys = lds0 + yi * TILE;
      const unsigned xs = lds0 + 3 * TILE + (s & 1) * TILE;
      const bool more = s + 2 < nsteps;
      bf16x8 xf[2][4], yf[2][4];
      // ---- LOAD(2s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        xf[j][0] = lds_frag<0>(xs + xa[j]); xf[j][1] = lds_frag<1>(xs + xa[j]);
        xf[j][2] = lds_frag<2>(xs + xa[j]); xf[j][3] = lds_frag<3>(xs + xa[j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        yf[i][0] = lds_frag<0>(ys + ya[0][i]); yf[i][1] = lds_frag<1>(ys + ya[0][i]);
        yf[i][2] = lds_frag<2>(ys + ya[0][i]); yf[i][3] = lds_frag<3>(ys + ya[0][i]);
      }
      int b2 = 0, t2 = 0, l2 = 0;
      if (more) {
        entry(s_begin + s + 2, b2, t2, l2);
        int y2 = yi + 2;
        y2 = y2 >= 3 ? y2 - 3 : y2;
        stage_y(b2, t2, l2, y2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wpp_barrier();
      // ---- COMPUTE(2s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][kk], xf[j][kk], acc[0][i][j], 0, 0, 0);
      wpp_barrier();
      // ---- LOAD(2s+1): the tiles of step s+1 (issued a step ago) must have landed before the
      //      next barrier pair; only the 4 dY instructions of this step may stay in flight
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        yf[i][0] = lds_frag<0>(ys + ya[1][i]); yf[i][1] = lds_frag<1>(ys + ya[1][i]);
        yf[i][2] = lds_frag<2>(ys + ya[1][i]); yf[i][3] = lds_frag<3>(ys + ya[1][i]);
      }
      if (more) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        stage_x(b2, t2, l2, s & 1);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      yi = yi + 1 >= 3 ? 0 : yi + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wpp_barrier();
      // ---- COMPUTE(2s+1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[1][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][kk], xf[j][kk], acc[1][i][j], 0, 0, 0);
      wpp_barrier();
    }
    if (!grp) wpp_barrier();
  }
  __syncthreads();

  if (npiece > 1) {
    const int sidx = rank - nfull;
    auto at = [&](int v) -> f32x16& { return acc[v >> 2][(v >> 1) & 1][v & 1]; };
    if (!split_publish_and_reduce(at, p.ws_slabs + (size_t)sidx * f * kSplitSlabFloats,
                                  p.ws_cnt + sidx, piece, f, smem, tid))
      return;
  }

  // ---- epilogue: the one owner of the tile writes dW (fp32), ci contiguous across lanes; the
  //      accumulate path loads all 64 old values of a co half before the adds (see above) ---------
  const int l31 = lane & 31;
  const bool inside = __builtin_amdgcn_readfirstlane((co0 + 256 <= p.Cout && ci0 + 256 <= p.Cin) ? 1 : 0);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int cob = co0 + grp * 128 + e * 64, cib = ci0 + wn * 64;
    if (inside) {
      float* const base = p.dw + (long long)(cob + 4 * lhi) * p.Cin + cib + l31;
      float old[2][2][16];
      if (p.accumulate) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
              old[i][j][v] = base[(long long)(i * 32 + (v & 3) + 8 * (v >> 2)) * p.Cin + j * 32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[e][i][j][v] += old[i][j][v];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            base[(long long)(i * 32 + (v & 3) + 8 * (v >> 2)) * p.Cin + j * 32] = acc[e][i][j][v];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ci = cib + j * 32 + l31;
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int co = cob + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lhi;
            if (co < p.Cout && ci < p.Cin) {
              float* dst = p.dw + (long long)co * p.Cin + ci;
              float val = acc[e][i][j][v];
              if (p.accumulate) val += *dst;
              *dst = val;
            }
          }
        }
    }
  }
}

}  // namespace os2s

// accumulate = 0 : dW = grad;  accumulate = 1 : dW += grad.
// With a workspace (os2s_conv1d_workspace_bytes(), zero tickets, one per stream — the same
// contract as os2s_conv1d_fwd_ws) the ping-pong kernel spreads small layers over the chip by
// cutting the reduction; its result is deterministic and written by one owner per element.
// The lockstep kernel (stride > 1, K = 1, narrow layers) splits the batch with fp32 atomics when
// accumulate = 1 and the layer is too small to fill the chip otherwise.
static int conv1d_wgrad_impl(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                             const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                             int Tin, int Cin, int Cout, int K, int stride, int dil,
                             int padL, int Tout, int accumulate, void* workspace,
                             size_t workspace_bytes);

static int wgrad_pp_min_units() {   // experiment knob (A/B runs on one box); default 8
  static const int v = [] { const char* e = getenv("OS2S_WGRAD_PP_MIN_UNITS"); return e ? atoi(e) : 8; }();
  return v;
}
// K = 1: the 256 x 256 ping-pong tile needs a long reduction per output tile to pay for its
// pipeline fill and the split reduction (OS2S_WGRAD1X1_PP: 0 = never, 1 = whenever the shape
// allows, unset = by shape)
static bool wgrad1x1_pp_auto(int B, int T, int Cin, int Cout, bool have_ws) {
  static const int mode = [] { const char* e = getenv("OS2S_WGRAD1X1_PP"); return e ? atoi(e) : -1; }();
  if (mode == 0) return false;
  if (mode == 1) return true;
  // measured (tools/bench_dense_shapes.py, 8300 rows): 2048 x 1024 and larger beat the lockstep
  // kernel and hipBLASLt (0.073 vs 0.082 / 0.075 ms; 32768 x 1024: 0.465 vs 1.03 / 0.526 ms); a
  // 1024 x 1024 output is 16 tiles cut 16 ways and loses to the lockstep kernel (0.071 vs 0.054 ms),
  // as do the ragged B = 32 residual branches of Jasper below 1024 channels
  const long long rows = (long long)B * T;
  const int units = os2s::ceil_div(Cout, 256) * os2s::ceil_div(Cin, 256);
  return have_ws && rows >= 2048 && units >= 32;
}
static int g_wgrad_variant = -1;   // experiment / test hook: 0 = lockstep kernel, 1 = ping-pong, 2 = K = 1 ping-pong, 3 = one wave per SIMD
static int g_wgrad_split = -1;
static int g_wgrad_sw_ablate = 0;  // conv1d_wgrad.sw_ablate: only read by builds with -DOS2S_SW_ABLATE
static int g_wgrad_xcd = 1;        // conv1d_wgrad.xcd_order: 0 = rank = blockIdx.x (rounds 2 - 5), 1 = wgrad_xcd_rank
static unsigned long long* g_wgrad_dbg = nullptr;
static int g_wgrad_dbg_mode = 0;
// os2s_set_option: conv1d_wgrad.variant (0 = lockstep kernel, 1 = ping-pong, 2 = K = 1 ping-pong, -1 = by shape),
// conv1d_wgrad.split (> 0 forces the reduction split factor of the ping-pong kernels, -1 = cost model);
// debug stamps "conv1d_wgrad" (tools/pp_timeline.py): per-slot time stamps of the ping-pong kernel
static os2s::OptionReg r_wg_variant("conv1d_wgrad.variant", [](double v) { g_wgrad_variant = (int)v; });
static os2s::OptionReg r_wg_split("conv1d_wgrad.split", [](double v) { g_wgrad_split = (int)v; });
static os2s::OptionReg r_wg_abl("conv1d_wgrad.sw_ablate", [](double v) { g_wgrad_sw_ablate = (int)v; });
static os2s::OptionReg r_wg_xcd("conv1d_wgrad.xcd_order", [](double v) { g_wgrad_xcd = v != 0 ? 1 : 0; });
static os2s::StampReg r_wg_stamps("conv1d_wgrad", [](void* stamps, int mode) {
  g_wgrad_dbg = (unsigned long long*)stamps;
  g_wgrad_dbg_mode = mode;
});

extern "C" int os2s_conv1d_wgrad_ws(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                                    const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                                    int Tin, int Cin, int Cout, int K, int stride, int dil,
                                    int padL, int Tout, int accumulate, void* workspace,
                                    size_t workspace_bytes) {
  return conv1d_wgrad_impl(stream, x, x_row_stride, dy, dw, in_len, B, Tin, Cin, Cout, K, stride,
                           dil, padL, Tout, accumulate, workspace, workspace_bytes);
}

extern "C" int os2s_conv1d_wgrad_ex(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                                    const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                                    int Tin, int Cin, int Cout, int K, int stride, int dil,
                                    int padL, int Tout, int accumulate) {
  return conv1d_wgrad_impl(stream, x, x_row_stride, dy, dw, in_len, B, Tin, Cin, Cout, K, stride,
                           dil, padL, Tout, accumulate, nullptr, 0);
}

extern "C" int os2s_conv1d_wgrad(os2s_stream_t stream, const uint16_t* x,
                                 const uint16_t* dy, float* dw,
                                 const int32_t* in_len, int B, int Tin, int Cin,
                                 int Cout, int K, int stride, int dil, int padL,
                                 int Tout, int accumulate) {
  return conv1d_wgrad_impl(stream, x, Cin, dy, dw, in_len, B, Tin, Cin, Cout, K, stride, dil,
                           padL, Tout, accumulate, nullptr, 0);
}

// groups != nullptr (os2s_conv1d_wgrad_grouped_ws): ngroups layers of this one shape; only the ping-pong kernel takes
// them — returns OS2S_ERR_UNSUPPORTED when the shape is not its (the caller then launches the layers one by one)
static int conv1d_wgrad_impl_g(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                               const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                               int Tin, int Cin, int Cout, int K, int stride, int dil,
                               int padL, int Tout, int accumulate, void* workspace,
                               size_t workspace_bytes, const os2s_cwgrad_group_t* groups, int ngroups);

static int conv1d_wgrad_impl(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                             const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                             int Tin, int Cin, int Cout, int K, int stride, int dil,
                             int padL, int Tout, int accumulate, void* workspace,
                             size_t workspace_bytes) {
  return conv1d_wgrad_impl_g(stream, x, x_row_stride, dy, dw, in_len, B, Tin, Cin, Cout, K, stride, dil, padL, Tout,
                             accumulate, workspace, workspace_bytes, nullptr, 1);
}

// Up to 8 convolution layers of ONE shape (Cin, Cout, K, dilation, padding) over one batch (B, T, lengths) in one
// launch of the ping-pong weight-gradient kernel: the repeated sub-blocks of a Jasper block (conv_blocks.py:61-168:
// `repeat` x the same tf.layers.conv1d) are 12 - 150 units of work each on 256 CUs — alone each is cut up to 16
// ways along the reduction (fill, 256 KB slab per piece, one reducer per unit); together they fill the chip whole.
extern "C" int os2s_conv1d_wgrad_grouped_ws(os2s_stream_t stream, const os2s_cwgrad_group_t* groups, int ngroups,
                                            const int32_t* in_len, int B, int Tin, int Cin, int Cout, int K,
                                            int stride, int dil, int padL, int Tout, int accumulate,
                                            void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(groups && ngroups >= 1 && ngroups <= 8);
  for (int i = 0; i < ngroups; ++i) OS2S_REQUIRE(groups[i].x && groups[i].dy && groups[i].dw);
  for (int i = 1; i < ngroups; ++i) OS2S_REQUIRE(groups[i].x_row_stride == groups[0].x_row_stride);
  int rc = OS2S_ERR_UNSUPPORTED;
  if (ngroups > 1)
    rc = conv1d_wgrad_impl_g(stream, groups[0].x, groups[0].x_row_stride, groups[0].dy, groups[0].dw, in_len, B, Tin,
                             Cin, Cout, K, stride, dil, padL, Tout, accumulate, workspace, workspace_bytes, groups,
                             ngroups);
  if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  for (int i = 0; i < ngroups; ++i) {
    rc = conv1d_wgrad_impl(stream, groups[i].x, groups[i].x_row_stride, groups[i].dy, groups[i].dw, in_len, B, Tin,
                           Cin, Cout, K, stride, dil, padL, Tout, accumulate, workspace, workspace_bytes);
    if (rc != OS2S_OK) return rc;
  }
  return OS2S_OK;
}

static int conv1d_wgrad_impl_g(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                               const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                               int Tin, int Cin, int Cout, int K, int stride, int dil,
                               int padL, int Tout, int accumulate, void* workspace,
                               size_t workspace_bytes, const os2s_cwgrad_group_t* groups, int ngroups) {
  using namespace os2s;
  OS2S_REQUIRE(x_row_stride >= Cin && x_row_stride % 8 == 0);
  OS2S_REQUIRE(x && dy && dw);
  OS2S_REQUIRE(B >= 0 && Tin >= 1 && Tout >= 1 && Cin >= 8 && Cout >= 8 && K >= 1);
  OS2S_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && stride >= 1 && dil >= 1);
  if (B == 0) return OS2S_OK;
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.in_len = in_len;
  a.B = B; a.Tin = Tin; a.Tout = Tout; a.Cin = Cin; a.Cout = Cout; a.K = K;
  a.stride = stride; a.dil = dil; a.padL = padL; a.x_ld = x_row_stride;
  a.accumulate = accumulate ? 1 : 0;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = g_wgrad_split;
  a.dbg = g_wgrad_dbg; a.dbg_mode = g_wgrad_dbg_mode; a.xcd_order = g_wgrad_xcd; a.NG = 1;
  for (int i = 0; i < 8; ++i) { a.gx[i] = nullptr; a.gdy[i] = nullptr; a.gdw[i] = nullptr; }
  if (groups) {
    a.NG = ngroups;
    for (int i = 0; i < ngroups; ++i) { a.gx[i] = groups[i].x; a.gdy[i] = groups[i].dy; a.gdw[i] = groups[i].dw; }
  }

  // ---- ping-pong kernel ------------------------------------------------------------------
  const bool pp_shape = stride == 1 && K >= 2 && B <= 64 && Cout >= 128 && Cin >= 64 &&
                        Tin < 65536 && Tout <= 255 * 64 && 63 + 3 * dil + 1 <= 96 &&
                        (long long)Tin * x_row_stride * 2 < (1ll << 30) &&
                        (long long)Tin * x_row_stride * 2 < (1ll << 31) &&
                        (long long)Tout * Cout * 2 < (1ll << 31);
  // (co, ci, 4-tap) tiles of the ping-pong kernel; with the reduction split across workgroups even
  // the 256-channel layers (12 tiles) beat the lockstep kernel (0.076 vs 0.093 ms ragged)
  const int pp_units = ceil_div(Cout, 128) * ceil_div(Cin, 128) * ceil_div(K, kWppTaps);
  // ---- one wave per SIMD, 16 accumulator blocks per wave (conv1d_wgrad_sw.hpp; round 6): the same units ----
  const bool sw_shape = pp_shape && 63 + 3 * dil + 1 <= kSwXInstr * 16 &&     // the X window fits the 20 KB slot
                        Cout % 128 == 0 && Cin % 128 == 0;                      // whole tiles only
  // Opt-in (conv1d_wgrad.variant 3): measured against the ping-pong kernel on the 768 x 768 x 25 layer it needs 3 050
  // cycles per 64-row step at 2.11 GHz where the ping-pong kernel needs 2 400 at 1.75 GHz — 0.648 vs 0.612 ms
  // (profiles/r06_wgrad_sw_ablation.txt, DESIGN.md "Round-6 kernel work").
  if (sw_shape && g_wgrad_variant == 3 && !groups) {
    a.NCO = ceil_div(Cout, 128);
    a.NCI = ceil_div(Cin, 128);
    a.NTP = ceil_div(K, kWppTaps);
    a.NSPLIT = 1; a.steps_per_split = 0; a.use_atomic = 0;
    a.xrows = 63 + (kWppTaps - 1) * dil + 1;
    a.xrows_pad = ceil_div(a.xrows, 4) * 4;
    a.xbuf_bytes = kSwXBuf;
    a.steptab_bytes = ceil_div(B * ceil_div(Tout, 64) * 4, 16) * 16;
    const size_t smem = (size_t)kSwRing * (kSwYBuf + kSwXBuf) + a.steptab_bytes;
    if (smem <= 160 * 1024) {
      static std::once_flag once_sw;
      static hipError_t attr_rc = hipSuccess;
      static int ncu = 256;
      std::call_once(once_sw, [] {
        attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad_sw_kernel<0>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef OS2S_SW_ABLATE
        for (const void* k : {(const void*)conv1d_wgrad_sw_kernel<1>, (const void*)conv1d_wgrad_sw_kernel<2>,
                              (const void*)conv1d_wgrad_sw_kernel<4>, (const void*)conv1d_wgrad_sw_kernel<8>,
                              (const void*)conv1d_wgrad_sw_kernel<3>, (const void*)conv1d_wgrad_sw_kernel<5>})
          if (attr_rc == hipSuccess) attr_rc = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
          ncu = n;
      });
      if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
      a.ncu = ncu;
      const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
      if (workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
        a.ws_cnt = reinterpret_cast<int*>(workspace);
        a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
        size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
        const size_t cap = (size_t)3 * ncu;
        a.ws_nslabs = (int)(n < cap ? n : cap);
      }
      const int U = a.NCO * a.NCI * a.NTP;
      const int r = U % ncu;
      const int pieces = a.ws_nslabs < 16 * r ? a.ws_nslabs : 16 * r;
#ifdef OS2S_SW_ABLATE      // measurement build only (tools/sw_ablate.py): the stream with one ingredient removed
      switch (g_wgrad_sw_ablate) {
        case 1: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<1>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        case 2: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<2>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        case 3: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<3>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        case 4: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<4>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        case 5: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<5>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        case 8: OS2S_LAUNCH(conv1d_wgrad_sw_kernel<8>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a); return OS2S_OK;
        default: break;
      }
#endif
      OS2S_LAUNCH(conv1d_wgrad_sw_kernel<0>, dim3(U + pieces), dim3(256), smem, (hipStream_t)stream, a);
      return OS2S_OK;
    }
  }

  if (pp_shape && g_wgrad_variant != 0 && (groups || g_wgrad_variant == 1 || pp_units >= wgrad_pp_min_units())) {
    a.NCO = ceil_div(Cout, 128);
    a.NCI = ceil_div(Cin, 128);
    a.NTP = ceil_div(K, kWppTaps);
    a.NSPLIT = 1; a.steps_per_split = 0; a.use_atomic = 0;
    a.xrows = 63 + (kWppTaps - 1) * dil + 1;
    a.xrows_pad = ceil_div(a.xrows, 4) * 4;
    a.xbuf_bytes = 24 * 1024;                         // 3 DMA rounds of 8 waves x 1 KB
    a.steptab_bytes = ceil_div(B * ceil_div(Tout, 64) * 4, 16) * 16;
    const size_t smem = (size_t)2 * 64 * 256 + (size_t)3 * a.xbuf_bytes + a.steptab_bytes + (a.dbg ? 2 * 48 * 10 * 8 : 0);
    if (smem <= 160 * 1024) {
      static std::once_flag once;
      static hipError_t attr_rc = hipSuccess;
      static int ncu = 256;
      std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad_pp_kernel<false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_rc == hipSuccess)
          attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad_pp_kernel<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
          ncu = n;
      });
      if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
      a.ncu = ncu;
      const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
      if (workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
        a.ws_cnt = reinterpret_cast<int*>(workspace);
        a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
        size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
        const size_t cap = (size_t)3 * ncu;
        a.ws_nslabs = (int)(n < cap ? n : cap);
      }
      // upper bound of the grid (the split factor is decided on the device from the live
      // length of the batch): whole units, or up to 16 pieces of each unit of the tail
      const int U = a.NG * a.NCO * a.NCI * a.NTP;
      const int r = U % ncu;
      int pieces = a.ws_nslabs < 16 * r ? a.ws_nslabs : 16 * r;
      const int grid = U + pieces;
      if (a.dbg) {
        OS2S_LAUNCH(conv1d_wgrad_pp_kernel<true>, dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
      } else {
        OS2S_LAUNCH(conv1d_wgrad_pp_kernel<false>, dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
      }
      return OS2S_OK;
    }
  }

  if (groups) return OS2S_ERR_UNSUPPORTED;               // only the ping-pong kernel ranks units over groups

  // ---- ping-pong kernel of the K = 1 case (Dense weight gradients) ----------------------------
  const bool pp1_shape = K == 1 && stride == 1 && padL == 0 && Tin == Tout && B <= 64 &&
                         Cout >= 128 && Cin >= 128 && x_row_stride * 2 * 64 < (1ll << 30) &&
                         (long long)Cout * 2 * 64 < (1ll << 30);
  if (pp1_shape && g_wgrad_variant != 0 && g_wgrad_variant != 1 &&
      (g_wgrad_variant == 2 || wgrad1x1_pp_auto(B, Tout, Cin, Cout, workspace != nullptr))) {
    a.NCO = ceil_div(Cout, 256);
    a.NCI = ceil_div(Cin, 256);
    a.NTP = 1;
    a.NSPLIT = 1; a.steps_per_split = 0; a.use_atomic = 0;
    a.xrows = 64; a.xrows_pad = 64; a.xbuf_bytes = 0; a.steptab_bytes = 0;
    static std::once_flag once1;
    static hipError_t attr_rc1 = hipSuccess;
    static int ncu1 = 256;
    std::call_once(once1, [] {
      attr_rc1 = hipFuncSetAttribute((const void*)conv1d_wgrad1x1_pp_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        ncu1 = n;
    });
    if (attr_rc1 != hipSuccess) return OS2S_ERR_LAUNCH;
    a.ncu = ncu1;
    const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
    if (workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
      a.ws_cnt = reinterpret_cast<int*>(workspace);
      a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
      size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
      const size_t cap = (size_t)3 * ncu1;
      a.ws_nslabs = (int)(n < cap ? n : cap);
    }
    const int U = a.NCO * a.NCI;
    const int r = U % ncu1;
    const int pieces = a.ws_nslabs < 16 * r ? a.ws_nslabs : 16 * r;
    WgradGroupTable none;
    none.ngroups = 0; none.total_units = 0;
    OS2S_LAUNCH(conv1d_wgrad1x1_pp_kernel, dim3(U + pieces), dim3(512), (size_t)160 * 1024,
                (hipStream_t)stream, a, none);
    return OS2S_OK;
  }

  // ---- lockstep kernel ---------------------------------------------------------------------
  const int TAPS = K == 1 ? 1 : 2;      // 1x1 convs: one accumulator set, no idle tap slot
  // the 256-wide tile pays off once there are enough 256-channel tiles to fill the chip
  const bool wide = (Cout % 256 == 0) && (K >= 8) && (Cout >= 512);
  const int COT = wide ? 256 : 128;
  a.NCO = ceil_div(Cout, COT);
  a.NCI = ceil_div(Cin, 128);
  a.NTP = ceil_div(K, TAPS);
  const int base_blocks = a.NCO * a.NCI * a.NTP;
  const int total_steps = B * ceil_div(Tout, 64);
  int nsplit = 1;
  if (accumulate) {
    // ~1 round of workgroups: the kernel shares the GPU with the data-gradient chain (side
    // stream), where fewer, longer workgroups and fewer atomic passes over dW win slightly
    const int target = wide ? 256 : 512;
    nsplit = ceil_div(target, base_blocks);
    const int max_split = total_steps / 8 > 0 ? total_steps / 8 : 1;   // >= 8 steps per block
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if (os2s_deterministic()) nsplit = 1;     // one add per dW element
  }
  a.steps_per_split = ceil_div(total_steps, nsplit);
  a.NSPLIT = ceil_div(total_steps, a.steps_per_split);
  a.use_atomic = accumulate ? 1 : 0;
  a.xrows = 63 * stride + (TAPS - 1) * dil + 1;
  a.xrows_pad = ceil_div(a.xrows, 4) * 4;
  const size_t smem = (size_t)2 * 64 * 2 * COT + (size_t)2 * a.xrows_pad * 256;
  if (smem > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  static std::once_flag once2;
  static hipError_t attr_rc2 = hipSuccess;
  std::call_once(once2, [] {
    if (hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<2, 128>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<2, 256>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<1, 128>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<1, 256>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      attr_rc2 = hipErrorUnknown;
  });
  if (attr_rc2 != hipSuccess) return OS2S_ERR_LAUNCH;
  const int nunits = a.NCO * a.NCI * a.NSPLIT;
  const int grid = ceil_div(nunits, 8) * 8 * a.NTP;
  if (wide) {
    if (TAPS == 1)
      OS2S_LAUNCH((conv1d_wgrad_kernel<1, 256>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    else
      OS2S_LAUNCH((conv1d_wgrad_kernel<2, 256>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
  } else {
    if (TAPS == 1)
      OS2S_LAUNCH((conv1d_wgrad_kernel<1, 128>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
    else
      OS2S_LAUNCH((conv1d_wgrad_kernel<2, 128>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  }
  return OS2S_OK;
}


// dW_i[co][ci] += sum_(b,t) dY_i[b,t,co] * X_i[b,t,ci] for up to 16 (X_i, dY_i, dW_i) over one ragged
// batch, in one launch (the K = 1 weight gradients of the dense-residual branches of a block end).
// dW_i are fp32 and are ACCUMULATED into (fp32 atomics: with one reduction split — whenever the
// groups together hold >= 256 output tiles — every element receives exactly one add).
extern "C" int os2s_conv1x1_wgrad_grouped(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                                          const int32_t* in_len, int B, int T) {
  using namespace os2s;
  OS2S_REQUIRE(groups && ngroups >= 1 && ngroups <= kMaxWgradGroups && B >= 0 && T >= 1);
  if (B == 0) return OS2S_OK;
  WgradGroupTable gt;
  gt.ngroups = ngroups;
  int tiles = 0;
  for (int i = 0; i < ngroups; ++i) {
    const os2s_wgrad_group_t& s = groups[i];
    OS2S_REQUIRE(s.x && s.dy && s.dw && s.Cin >= 8 && s.Cout >= 8 && s.Cin % 8 == 0 && s.Cout % 8 == 0);
    OS2S_REQUIRE(s.x_row_stride >= s.Cin && s.x_row_stride % 8 == 0);
    tiles += ceil_div(s.Cout, 128) * ceil_div(s.Cin, 128);
  }
  const int total_steps = B * ceil_div(T, 64);
  // ~2 workgroups per CU (the kernel shares the GPU with the data-gradient chain); >= 8 steps each
  int nsplit = ceil_div(512, tiles);
  const int max_split = total_steps / 8 > 0 ? total_steps / 8 : 1;
  if (nsplit > max_split) nsplit = max_split;
  if (tiles >= 256 || nsplit < 1 || os2s_deterministic()) nsplit = 1;
  WgradArgs a;
  a.x = nullptr; a.dy = nullptr; a.dw = nullptr; a.in_len = in_len;
  a.B = B; a.Tin = T; a.Tout = T; a.Cin = 0; a.Cout = 0; a.K = 1;
  a.stride = 1; a.dil = 1; a.padL = 0; a.x_ld = 0; a.accumulate = 1;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = -1;
  a.dbg = nullptr; a.dbg_mode = 0; a.xcd_order = g_wgrad_xcd; a.NG = 1;
  a.NCO = 0; a.NCI = 0; a.NTP = 1;
  a.steps_per_split = ceil_div(total_steps, nsplit);
  a.NSPLIT = ceil_div(total_steps, a.steps_per_split);
  a.use_atomic = 1;
  a.xrows = 64; a.xrows_pad = 64; a.xbuf_bytes = 0; a.steptab_bytes = 0;
  int units = 0;
  for (int i = 0; i < kMaxWgradGroups; ++i) {
    const os2s_wgrad_group_t& s = groups[i < ngroups ? i : 0];
    WgradGroup& g = gt.g[i];
    g.x = s.x; g.dy = s.dy; g.dw = s.dw; g.x_ld = s.x_row_stride;
    g.Cin = s.Cin; g.Cout = s.Cout; g.NCI = ceil_div(s.Cin, 128);
    g.unit_begin = units;
    if (i < ngroups) units += ceil_div(s.Cout, 128) * g.NCI * a.NSPLIT;
  }
  gt.total_units = units;
  const size_t smem = (size_t)2 * 64 * 2 * 128 + (size_t)2 * a.xrows_pad * 256;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad_grouped_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH(conv1d_wgrad_grouped_kernel, dim3(units), dim3(256), smem, (hipStream_t)stream, a, gt);
  return OS2S_OK;
}


// The same grouped K = 1 weight gradients on the ping-pong TN-GEMM kernel (conv1d_wgrad1x1_pp_kernel with its
// group table; round 5): the residual branches of a Jasper block end are 3 x 1 ... 3 x 3 tiles of 256 x 256 each
// — far too few for a launch of their own, 50 - 60 together, cut along the reduction by the tail split —, the
// reduction runs over the live 64-row chunks of the ragged batch only, one owner writes each dW element (no
// atomics: deterministic). The lockstep grouped kernel above reached 0.105 of the MFMA peak on these launches.
// Falls back to it when a group is narrower than 128 channels or the batch has fewer than 2048 rows.
extern "C" int os2s_conv1x1_wgrad_grouped_ws(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                                             const int32_t* in_len, int B, int T, void* workspace,
                                             size_t workspace_bytes) {
  using namespace os2s;
  OS2S_REQUIRE(groups && ngroups >= 1 && ngroups <= kMaxWgradGroups && B >= 0 && T >= 1);
  if (B == 0) return OS2S_OK;
  bool pp = B <= 64 && (long long)B * T >= 2048 && workspace != nullptr;
  for (int i = 0; i < ngroups && pp; ++i) {
    const os2s_wgrad_group_t& s = groups[i];
    OS2S_REQUIRE(s.x && s.dy && s.dw && s.Cin >= 8 && s.Cout >= 8 && s.Cin % 8 == 0 && s.Cout % 8 == 0);
    OS2S_REQUIRE(s.x_row_stride >= s.Cin && s.x_row_stride % 8 == 0);
    pp = s.Cin >= 128 && s.Cout >= 128 && s.x_row_stride * 2 * 64 < (1ll << 30) &&
         (long long)s.Cout * 2 * 64 < (1ll << 30);
  }
  if (!pp) return os2s_conv1x1_wgrad_grouped(stream, groups, ngroups, in_len, B, T);
  WgradGroupTable gt;
  gt.ngroups = ngroups;
  int units = 0;
  for (int i = 0; i < kMaxWgradGroups; ++i) {
    const os2s_wgrad_group_t& s = groups[i < ngroups ? i : 0];
    WgradGroup& g = gt.g[i];
    g.x = s.x; g.dy = s.dy; g.dw = s.dw; g.x_ld = s.x_row_stride;
    g.Cin = s.Cin; g.Cout = s.Cout; g.NCI = ceil_div(s.Cin, 256);
    g.unit_begin = units;
    if (i < ngroups) units += ceil_div(s.Cout, 256) * g.NCI;
  }
  gt.total_units = units;
  WgradArgs a;
  a.x = gt.g[0].x; a.dy = gt.g[0].dy; a.dw = gt.g[0].dw; a.in_len = in_len;
  a.B = B; a.Tin = T; a.Tout = T; a.Cin = gt.g[0].Cin; a.Cout = gt.g[0].Cout; a.K = 1;
  a.stride = 1; a.dil = 1; a.padL = 0; a.x_ld = gt.g[0].x_ld;
  a.accumulate = 1;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = g_wgrad_split;
  a.dbg = nullptr; a.dbg_mode = 0; a.xcd_order = g_wgrad_xcd; a.NG = 1;
  a.NCO = units; a.NCI = 1; a.NTP = 1;                   // U = NCO * NCI = all units of all groups
  a.NSPLIT = 1; a.steps_per_split = 0; a.use_atomic = 0;
  a.xrows = 64; a.xrows_pad = 64; a.xbuf_bytes = 0; a.steptab_bytes = 0;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad1x1_pp_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n;
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  a.ncu = ncu;
  const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
  if (workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
    a.ws_cnt = reinterpret_cast<int*>(workspace);
    a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
    size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
    const size_t cap = (size_t)3 * ncu;
    a.ws_nslabs = (int)(n < cap ? n : cap);
  }
  const int r = units % ncu;
  const int pieces = a.ws_nslabs < 16 * r ? a.ws_nslabs : 16 * r;
  OS2S_LAUNCH(conv1d_wgrad1x1_pp_kernel, dim3(units + pieces), dim3(512), (size_t)160 * 1024,
              (hipStream_t)stream, a, gt);
  return OS2S_OK;
}

// The Dense weight gradients dw_i[Cout_i, Cin_i] (+)= dy_i^T x_i of up to 16 layers that see the
// same rows (one packed token batch [M, .]) in ONE launch of the K = 1 ping-pong kernel
// (conv1d_wgrad1x1_pp_kernel): deterministic (one owner per dW element, split reductions summed in
// piece order), no atomics. For the small outputs — the 1024 x 1024 projections of the Transformer:
// 16 tiles each — that a launch of their own cannot spread over the chip.
extern "C" int os2s_gemm_wgrad_grouped(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                                       long long M, int accumulate, void* workspace, size_t workspace_bytes) {
  using namespace os2s;
  OS2S_REQUIRE(groups && ngroups >= 1 && ngroups <= kMaxWgradGroups && M >= 1 && M < (1ll << 30));
  WgradGroupTable gt;
  gt.ngroups = ngroups;
  int units = 0;
  for (int i = 0; i < kMaxWgradGroups; ++i) {
    const os2s_wgrad_group_t& s = groups[i < ngroups ? i : 0];
    OS2S_REQUIRE(s.x && s.dy && s.dw && s.Cin >= 128 && s.Cout >= 128 && s.Cin % 8 == 0 && s.Cout % 8 == 0);
    OS2S_REQUIRE(s.x_row_stride >= s.Cin && s.x_row_stride % 8 == 0);
    OS2S_REQUIRE(s.x_row_stride * 2 * 64 < (1ll << 30) && (long long)s.Cout * 2 * 64 < (1ll << 30));
    WgradGroup& g = gt.g[i];
    g.x = s.x; g.dy = s.dy; g.dw = s.dw; g.x_ld = s.x_row_stride;
    g.Cin = s.Cin; g.Cout = s.Cout; g.NCI = ceil_div(s.Cin, 256);
    g.unit_begin = units;
    if (i < ngroups) units += ceil_div(s.Cout, 256) * g.NCI;
  }
  gt.total_units = units;
  WgradArgs a;
  a.x = gt.g[0].x; a.dy = gt.g[0].dy; a.dw = gt.g[0].dw; a.in_len = nullptr;
  a.B = 1; a.Tin = (int)M; a.Tout = (int)M; a.Cin = gt.g[0].Cin; a.Cout = gt.g[0].Cout; a.K = 1;
  a.stride = 1; a.dil = 1; a.padL = 0; a.x_ld = gt.g[0].x_ld;
  a.accumulate = accumulate ? 1 : 0;
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = g_wgrad_split;
  a.dbg = nullptr; a.dbg_mode = 0; a.xcd_order = g_wgrad_xcd; a.NG = 1;
  a.NCO = units; a.NCI = 1; a.NTP = 1;                   // U = NCO * NCI = all units of all groups
  a.NSPLIT = 1; a.steps_per_split = 0; a.use_atomic = 0;
  a.xrows = 64; a.xrows_pad = 64; a.xbuf_bytes = 0; a.steptab_bytes = 0;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1d_wgrad1x1_pp_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n;
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  a.ncu = ncu;
  const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
  if (workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
    a.ws_cnt = reinterpret_cast<int*>(workspace);
    a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
    size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
    const size_t cap = (size_t)3 * ncu;
    a.ws_nslabs = (int)(n < cap ? n : cap);
  }
  const int r = units % ncu;
  const int pieces = a.ws_nslabs < 16 * r ? a.ws_nslabs : 16 * r;
  OS2S_LAUNCH(conv1d_wgrad1x1_pp_kernel, dim3(units + pieces), dim3(512), (size_t)160 * 1024,
              (hipStream_t)stream, a, gt);
  return OS2S_OK;
}
