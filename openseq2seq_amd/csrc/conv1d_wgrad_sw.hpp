// Weight gradient, one wave per SIMD (round 6): included by conv1d_wgrad.hip inside namespace os2s, after
// WgradArgs / lds_frag / WgradGroupTable.
//
// The ping-pong weight-gradient kernels (conv1d_wgrad_pp_kernel, conv1d_wgrad1x1_pp_kernel) give every wave 8
// accumulator blocks (32 x 32 fp32): a k-slice of 16 reduction rows then needs 6 operand fragments for 8 MFMAs, 1.5
// ds_read_b64_tr_b16 per MFMA, and the LOAD slots (592 / 838 cycles, tools/pp_timeline.py) are longer than the
// COMPUTE slots they are paired with (16 MFMAs = 512) — the matrix pipe waits for the LDS. A wave with 16 blocks
// (4 x 4: 8 fragments for 16 MFMAs, 1.0 read per MFMA — the minimum for 16 blocks) owns 256 accumulator registers,
// so there is ONE wave per SIMD (4 waves = 256 threads per workgroup, one workgroup per CU, up to 512 registers a
// lane) and nothing to ping-pong with: the wave pipelines itself. Its instruction stream is written out by hand
// (every instruction of the loop is inline assembly, so the order below IS the issue order):
//
//   phase kk of step s (kk = 0..3, 16 reduction rows each):
//     s_waitcnt lgkmcnt(0)                      fragments of this phase (requested a phase ago) are in registers
//     16 MFMAs on them; behind MFMA 0..7 the two transpose reads of ONE fragment of the next phase (so the last
//     read is 8 MFMAs = 256 cycles old when the next phase starts), behind MFMAs 9 / 11 / 13 of phases 0..2 one
//     LDS-DMA instruction of step s + 2 (9 per wave and step);
//   phase 3: after MFMA 3: s_waitcnt vmcnt(9) — the tiles of step s + 1, requested a step ago — and the ONE
//     s_barrier of the step; the reads of step s + 1's first fragments follow it, two fragments per MFMA gap.
//
// Rings of three [64 rows][128 channels] dY tiles and three X windows: the DMA of step s + 2 lands in the slot read
// during step s - 1, which every wave left before the barrier of step s - 1.
//
// Two tile geometries share the loop (the fragment address tables and the staging differ):
//   CONV  (K >= 2): workgroup = 128 co x 128 ci x FOUR adjacent taps from one dY tile and one (64 + 3 dil)-row X
//          window (the geometry of conv1d_wgrad_pp_kernel: same units, step table, tail split, epilogue order);
//          wave (tg, wn) owns all 128 co x 64 ci x taps {2 tg, 2 tg + 1}: acc[i][q], i = co block, q = 2 e + j.
//          A wave whose taps lie past K (K = 4n + 1: the last quad has one live tap) skips reads and MFMAs.
#pragma once
#include <type_traits>

// ---- the hand-written stream's instruction wrappers --------------------------------------------------------------
// accumulators are pinned to the accumulation registers (256 of the 512), operands to vector registers
__device__ __forceinline__ void sw_mfma(f32x16& c, const bf16x8& a, const bf16x8& b) {
  // ("memory": the LDS-DMA builtins between the MFMAs stay where they are written)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
}
template <int N>
__device__ __forceinline__ void sw_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void sw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void sw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// 8 operand fragments of one 16-row k-slice: y[i] = dY rows x 32 co (MFMA A operand), x[q] = X rows x 32 ci (B)
struct SwFrags {
  bf16x8 y[4];
  bf16x8 x[4];
};

// fragment f (0..3 = y, 4..7 = x) of k-slice KK; NQ = live x fragments (4, 2 or 0: taps past K are not read)
template <int KK, int NQ>
__device__ __forceinline__ void sw_read(SwFrags& f, const int n, const unsigned* ya, const unsigned* xa) {
  if (NQ == 0) return;
  switch (n) {
    case 0: f.y[0] = lds_frag<KK>(ya[0]); break;
    case 1: f.y[1] = lds_frag<KK>(ya[1]); break;
    case 2: f.y[2] = lds_frag<KK>(ya[2]); break;
    case 3: f.y[3] = lds_frag<KK>(ya[3]); break;
    case 4: f.x[0] = lds_frag<KK>(xa[0]); break;
    case 5: f.x[1] = lds_frag<KK>(xa[1]); break;
    case 6: if (NQ > 2) f.x[2] = lds_frag<KK>(xa[2]); break;
    default: if (NQ > 2) f.x[3] = lds_frag<KK>(xa[3]); break;
  }
}
// The LAST write of an accumulator block (phase 3 of a unit's last step) carries its own wait states: the compiler
// cannot see that the assembly is a matrix instruction, so neither its hazard recogniser nor its scheduler keeps a
// register read — a v_accvgpr_read, or a spill store it decides to place right behind the definition — out of the
// instruction's shadow (XDL write -> VALU / VMEM read of the result: 18 wait states, not interlocked by the hardware).
// Found as a rare wrong 32 x 32 block at full size: four blocks were spilled between the MFMAs of the last step.
__device__ __forceinline__ void sw_mfma_final(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+a"(c) : "v"(a), "v"(b) : "memory");
}
// MFMA m (0..15) of a phase: q-major so that the two x fragments requested last are needed last
template <int NQ, bool FINAL = false>
__device__ __forceinline__ void sw_mma(f32x16 (&acc)[4][4], const SwFrags& f, const int m) {
  const int q = m >> 2, i = m & 3;
  if (q < NQ) {
    if (FINAL) sw_mfma_final(acc[i][q], f.y[i], f.x[q]); else sw_mfma(acc[i][q], f.y[i], f.x[q]);
  }
}

// One reduction step = 4 phases. A holds k-slice 0 on entry; on exit it holds k-slice 0 of the NEXT step (read
// from ya_n / xa_n after the barrier) when NEXT is set. dma(k) issues LDS-DMA instruction k (0..8) of the step two
// ahead (DMA_ON: 9 stay in flight across the wait in front of the barrier; else everything drains). The three
// (DMA_ON, NEXT) combinations are separate instantiations — steady state, second-to-last and last step — so the
// stream has no branch.
// ABL: ablation mask of the measurement builds (-DOS2S_SW_ABLATE, tools/sw_ablate.py; results are then wrong):
// 1 = no LDS-DMA issue in the loop, 2 = no barrier, 4 = no transpose reads, 8 = no MFMAs. 0 in the product build.
template <int NQ, bool DMA_ON, bool NEXT, int ABL, class Dma>
__device__ __forceinline__ void sw_step(f32x16 (&acc)[4][4], SwFrags& A, SwFrags& B, const unsigned* ya,
                                        const unsigned* xa, const unsigned* ya_n, const unsigned* xa_n, Dma&& dma) {
  // phase 0: compute A (kk 0), read B (kk 1)
  sw_wait_lgkm<0>();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (!(ABL & 8)) sw_mma<NQ>(acc, A, m);
    if (m < 8 && !(ABL & 4)) sw_read<1, NQ>(B, m, ya, xa);
    if (DMA_ON && !(ABL & 1) && (m == 9 || m == 11 || m == 13)) dma((m - 9) >> 1);
  }
  // phase 1: compute B (kk 1), read A (kk 2)
  sw_wait_lgkm<0>();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (!(ABL & 8)) sw_mma<NQ>(acc, B, m);
    if (m < 8 && !(ABL & 4)) sw_read<2, NQ>(A, m, ya, xa);
    if (DMA_ON && !(ABL & 1) && (m == 9 || m == 11 || m == 13)) dma(3 + ((m - 9) >> 1));
  }
  // phase 2: compute A (kk 2), read B (kk 3)
  sw_wait_lgkm<0>();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (!(ABL & 8)) sw_mma<NQ>(acc, A, m);
    if (m < 8 && !(ABL & 4)) sw_read<3, NQ>(B, m, ya, xa);
    if (DMA_ON && !(ABL & 1) && (m == 9 || m == 11 || m == 13)) dma(6 + ((m - 9) >> 1));
  }
  // phase 3: compute B (kk 3); the step's barrier; read A (kk 0 of the next step)
  sw_wait_lgkm<0>();
#pragma unroll
  for (int m = 0; m < 4; ++m) if (!(ABL & 8)) sw_mma<NQ, !NEXT>(acc, B, m);
  if (NEXT) {
    if (DMA_ON) sw_wait_vm<9>(); else sw_wait_vm<0>();
    if (!(ABL & 2)) sw_barrier();
  }
#pragma unroll
  for (int m = 4; m < 16; ++m) {
    if (!(ABL & 8)) sw_mma<NQ, !NEXT>(acc, B, m);
    if (NEXT && m < 8 && !(ABL & 4)) {
      sw_read<0, NQ>(A, 2 * (m - 4), ya_n, xa_n);
      sw_read<0, NQ>(A, 2 * (m - 4) + 1, ya_n, xa_n);
    }
  }
}

constexpr int kSwXInstr = 5;                               // X-window DMA instructions per wave and step (20 KB slot)
constexpr int kSwYBuf = 64 * 256, kSwXBuf = kSwXInstr * 4 * 1024, kSwRing = 3;

template <int ABL>
__global__ __launch_bounds__(256, 1) void conv1d_wgrad_sw_kernel(WgradArgs p) {
  constexpr int BT = 64, COT = 128, CIT = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wid >> 1, wn = wid & 1;                   // tap pair of the quad, ci 64-half

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

  // ---- block -> (unit, piece): the mapping of conv1d_wgrad_pp_kernel --------------------------
  const int U = p.NCO * p.NCI * p.NTP, G = p.ncu;
  const int qd = U / G, r = U - qd * G;
  int f = 1;
  {
    int fmax = total_live / 8;                             // >= 8 steps per piece
    fmax = fmax > 16 ? 16 : fmax;
    if (p.force_split > 0 && p.ws_slabs) {
      f = p.force_split < fmax ? p.force_split : (fmax > 1 ? fmax : 1);
      while (f > 1 && r * f > p.ws_nslabs) --f;
      if (r == 0) f = 1;
    } else if (r > 0 && p.ws_slabs) {
      f = split_factor(r, G, 0.9f * total_live, fmax, p.ws_nslabs);
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
  const int tp = rank % p.NTP;
  const int rem = rank / p.NTP;
  const int co0 = (rem / p.NCI) * COT, ci0 = (rem % p.NCI) * CIT;
  const int k0 = tp * kWppTaps;
  const int sps = (total_live + npiece - 1) / npiece;
  const int s_begin = __builtin_amdgcn_readfirstlane(min(piece * sps, total_live));
  const int s_end = __builtin_amdgcn_readfirstlane(min(total_live, s_begin + sps));
  const int nsteps = s_end - s_begin;

  // LDS: dY ring (3 x 16 KB) | X ring (3 x 20 KB) | step table
  int* const steptab = reinterpret_cast<int*>(smem + kSwRing * (kSwYBuf + kSwXBuf));
  {
    const int excl = scan - nl;
    for (int c = 0; c < tchunks; ++c) {
      const int idx = excl + c - s_begin;
      if (lane < p.B && c < nl && idx >= 0 && idx < nsteps && wid == 0)
        steptab[idx] = lane | (c << 8) | (len_l << 16);
    }
  }

  // ---- DMA: per-lane byte offsets, fixed for the kernel (4 dY + 5 X instructions per wave and step) ------------
  int yv[4], xv[kSwXInstr];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int qq = (it * 4 + wid) * 64 + lane;
    const int row = qq >> 4, ps = qq & 15;
    const int u = (ps >> 1) ^ ((row & 3) << 1);
    const int ch = co0 + ((u << 1) | (ps & 1)) * 8;
    yv[it] = ch < p.Cout ? (row * p.Cout + ch) * 2 : (int)0x80000000;
  }
#pragma unroll
  for (int n = 0; n < kSwXInstr; ++n) {
    const int qq = (n * 4 + wid) * 64 + lane;
    const int row = qq >> 4, ps = qq & 15;
    const int u = (ps >> 1) ^ ((row & 3) << 1);
    const int ch = ci0 + ((u << 1) | (ps & 1)) * 8;
    xv[n] = (ch < p.Cin && row < p.xrows) ? (row * (int)p.x_ld + ch) * 2 : (int)0x80000000;
  }
  const unsigned long long dy_base = (unsigned long long)p.dy, x_base = (unsigned long long)p.x;
  const int ycol_bytes = __builtin_amdgcn_readfirstlane(p.Cout * 2);
  const int xcol_bytes = __builtin_amdgcn_readfirstlane((int)p.x_ld * 2);
  const unsigned long long xsample_bytes = (unsigned long long)p.Tin * (unsigned long long)p.x_ld * 2ull;
  auto prep = [&](int ent) __attribute__((always_inline)) -> WppStaged {
    const int b = ent & 0xff, t0 = ((ent >> 8) & 0xff) * BT, len_b = (int)((unsigned)ent >> 16);
    WppStaged st;
    const unsigned long long yb = dy_base + (unsigned long long)(unsigned)(b * p.Tout + t0) * (unsigned)ycol_bytes;
    st.ylo = (unsigned)yb; st.yhi = (unsigned)(yb >> 32);
    st.ynr = (unsigned)((p.Tout - t0) * ycol_bytes);
    const unsigned long long xb = x_base + (unsigned long long)(unsigned)b * xsample_bytes;
    st.xlo = (unsigned)xb; st.xhi = (unsigned)(xb >> 32);
    st.xnr = (unsigned)(len_b * xcol_bytes);
    st.xro = (t0 + k0 * p.dil - p.padL) * xcol_bytes;
    return st;
  };
  // LDS-DMA instruction k (0..3: dY, 4..8: X) of the step described by `st`, into ring slot `slot`
  auto issue1 = [&](const WppStaged& st, int slot, int k) __attribute__((always_inline)) {
    if (k < 4) {
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(((unsigned long long)st.yhi << 32) | st.ylo), 0, (int)st.ynr, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          yrs, (__attribute__((address_space(3))) void*)(smem + slot * kSwYBuf + (k * 4 + wid) * 1024), 16, yv[k], 0, 0, 0);
    } else {
      const int n = k - 4;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(((unsigned long long)st.xhi << 32) | st.xlo), 0, (int)st.xnr, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          xrs, (__attribute__((address_space(3))) void*)(smem + kSwRing * kSwYBuf + slot * kSwXBuf + (n * 4 + wid) * 1024),
          16, xv[n] + st.xro, 0, 0, 0);
    }
  };

  const bool live0 = k0 + 2 * tg < p.K, live1 = k0 + 2 * tg + 1 < p.K;
  f32x16 acc[4][4];                                        // [co block i][q = 2 * (tap of the pair) + ci block j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][q][v] = 0.f;

  const int lhi = lane >> 5;
  // Everything after the loop is one lambda, expanded at the end of EACH of the three stream variants below (and
  // once for a piece without steps): were the variants to join first, every accumulator would be a phi of three
  // register tuples, and with all 256 accumulation registers occupied the allocator resolves those through scratch.
  auto finish = [&]() __attribute__((always_inline)) {
    __syncthreads();
    // ---- epilogue: the one owner of the tile writes dW (fp32), ci contiguous across lanes. The accumulators stay
    //      in the accumulation registers: batches of 8 register groups (2 blocks x 4 groups of 4 co rows) are read
    //      out, added to the old values (accumulate: all 32 loads of a batch first) and stored; a split unit's
    //      reducer feeds the same batches from the slabs instead (os2s_split_reduce.hpp) -------------------------
    const int l31 = lane & 31;
    auto emit8 = [&](const int h, const f32x4 (&val)[8]) __attribute__((always_inline)) {
      // groups g = 8 h + k: block v = 2 h + (k >> 2) = (i, q) = (v >> 2, v & 3), rows 8 (k & 3) + c + 4 lhi of it
      const int i = (2 * h) >> 2;                            // both blocks of a batch share the co block
      // addressing: a wave-uniform 64-bit base per (block, co row) in scalar registers + ONE per-lane 32-bit byte
      // offset (global_load / global_store with a scalar base): per-element 64-bit vector addresses for the 256
      // elements of a lane, computed ahead by the scheduler, do not fit next to 256 pinned accumulators
      const unsigned loff = (unsigned)((4 * lhi) * p.Cin + l31) * 4u;
      const char* ub[2];
      bool tap_ok[2];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int q = (2 * h + b2) & 3, e = q >> 1, j = q & 1;
        const int tap = k0 + 2 * tg + e;
        tap_ok[b2] = tap < p.K;
        ub[b2] = reinterpret_cast<const char*>(p.dw + (long long)tap * p.Cout * p.Cin + (long long)(co0 + i * 32) * p.Cin +
                                               ci0 + wn * 64 + j * 32);
      }
      auto el = [&](int k, int c) __attribute__((always_inline)) -> float* {
        return reinterpret_cast<float*>(const_cast<char*>(ub[k >> 2]) + (size_t)((8 * (k & 3) + c) * p.Cin) * 4u + loff);
      };
      // (tiles are whole: the launcher takes this kernel only for Cout and Cin that are multiples of 128)
      f32x4 old[8];
      if (p.accumulate) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (tap_ok[k >> 2])
#pragma unroll
            for (int c = 0; c < 4; ++c) old[k][c] = *el(k, c);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (tap_ok[k >> 2])
#pragma unroll
          for (int c = 0; c < 4; ++c) *el(k, c) = p.accumulate ? val[k][c] + old[k][c] : val[k][c];
    };
    auto at = [&](int v) -> f32x16& { return acc[v >> 2][v & 3]; };
    if (npiece > 1) {
      const int sidx = rank - nfull;
      float* const slab0 = p.ws_slabs + (size_t)sidx * f * kSplitSlabFloats;
      if (!split_publish<256, 16>(at, slab0, p.ws_cnt + sidx, piece, f, smem, tid)) return;
      split_reduce_emit<256, 16>(slab0, f, tid, emit8);
      return;
    }
  #pragma unroll
    for (int h = 0; h < 8; ++h) {
      f32x4 val[8];
  #pragma unroll
      for (int k = 0; k < 8; ++k) {
        const f32x16& a = at(2 * h + (k >> 2));
        val[k] = f32x4{a[4 * (k & 3)], a[4 * (k & 3) + 1], a[4 * (k & 3) + 2], a[4 * (k & 3) + 3]};
      }
      emit8(h, val);
    }
  };

  if (nsteps > 0) {
    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    const int rsub = i16 >> 2, csub = (i16 & 3) * 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned ya0[4], xa0[4];                               // fragment addresses relative to the ring slot
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = 2 * i + g16;
      ya0[i] = lds0 + (lhi * 8 + rsub) * 256 + ((u ^ ((rsub & 3) << 1)) << 5) + csub;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r0 = lhi * 8 + rsub + (2 * tg + e) * p.dil;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = 4 * wn + 2 * j + g16;
        xa0[2 * e + j] = lds0 + kSwRing * kSwYBuf + r0 * 256 + ((u ^ ((r0 & 3) << 1)) << 5) + csub;
      }
    }
    const unsigned tab0 = lds0 + kSwRing * (kSwYBuf + kSwXBuf);
    __syncthreads();                                       // step table complete (no DMA in flight yet)
    {
      // both table entries are read BEFORE the first DMA is issued: behind an LDS-DMA the compiler drains
      // vmcnt(0) in front of every LDS access it can see
      const int e0 = __builtin_amdgcn_readfirstlane(steptab[0]);
      const int e1 = __builtin_amdgcn_readfirstlane(steptab[nsteps > 1 ? 1 : 0]);
      const WppStaged s0 = prep(e0);
#pragma unroll
      for (int k = 0; k < 9; ++k) issue1(s0, 0, k);
      if (nsteps > 1) {
        const WppStaged s1 = prep(e1);
#pragma unroll
        for (int k = 0; k < 9; ++k) issue1(s1, 1, k);
        sw_wait_vm<9>();
      } else {
        sw_wait_vm<0>();
      }
    }
    sw_barrier();

    SwFrags A, B;
    unsigned ya[4], xa[4], ya_n[4], xa_n[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ya[i] = ya0[i]; xa[i] = xa0[i]; }
    int slot = 0;
    // table entry of the step two ahead: requested a step early, consumed (one readfirstlane) in phase 0
    int ent_v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(ent_v) : "v"(tab0 + (2 < nsteps ? 2 : 0) * 4) : "memory");
    auto first_reads = [&](auto nq) {
      constexpr int NQ = decltype(nq)::value;
#pragma unroll
      for (int n = 0; n < 8; ++n) sw_read<0, NQ>(A, n, ya, xa);
    };
    // three loops / blocks in a row, not one loop with a three-way branch: a branch around the hand-written stream
    // makes every accumulator a phi of register tuples, which the register allocator resolves with copies through
    // vector registers inside the loop
    auto run = [&](auto nq) {
      constexpr int NQ = decltype(nq)::value;
      auto advance = [&]() __attribute__((always_inline)) {
        int s1 = slot + 1;
        s1 = s1 >= kSwRing ? 0 : s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ya_n[i] = ya0[i] + s1 * kSwYBuf;
          xa_n[i] = xa0[i] + s1 * kSwXBuf;
        }
        return s1;
      };
      auto commit = [&](int s1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ya[i] = ya_n[i]; xa[i] = xa_n[i]; }
        slot = s1;
      };
      int s = 0;
      for (; s + 2 < nsteps; ++s) {
        int s2 = slot + 2;
        s2 = s2 >= kSwRing ? s2 - kSwRing : s2;
        const int s1 = advance();
        // the entry requested a step ago has landed with the lgkmcnt(0) that opens phase 0; the dependence on
        // the wait is made explicit so that the read of the register cannot be scheduled in front of it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ent_v) : : "memory");
        const WppStaged st = prep(__builtin_amdgcn_readfirstlane(ent_v));
        asm volatile("ds_read_b32 %0, %1" : "=v"(ent_v) : "v"(tab0 + (s + 3 < nsteps ? s + 3 : 0) * 4) : "memory");
        sw_step<NQ, true, true, ABL>(acc, A, B, ya, xa, ya_n, xa_n,
                                [&](int k) __attribute__((always_inline)) { issue1(st, s2, k); });
        commit(s1);
      }
      if (s + 1 < nsteps) {
        const int s1 = advance();
        sw_step<NQ, false, true, ABL>(acc, A, B, ya, xa, ya_n, xa_n, [](int) {});
        commit(s1);
      }
      sw_step<NQ, false, false, ABL>(acc, A, B, ya, xa, ya_n, xa_n, [](int) {});
    };
    // (the MFMAs are invisible to the compiler's hazard recogniser: the last ones retire behind the s_nops before
    //  the accumulation registers are read)
    if (live1) {
      first_reads(std::integral_constant<int, 4>());
      run(std::integral_constant<int, 4>());
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      finish();
    } else if (live0) {
      first_reads(std::integral_constant<int, 2>());
      run(std::integral_constant<int, 2>());
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      finish();
    } else {
      run(std::integral_constant<int, 0>());
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      finish();
    }
  } else {
    finish();
  }
}
