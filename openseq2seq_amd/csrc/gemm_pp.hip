// Plain GEMM on the gfx950 matrix cores, hand-written (no vendor library):
//
//   C[M,N] (bf16, or fp32) (+)= A[M,K] . W[N,K]^T  (+ bias, ReLU, dropout, residual)
//
// Both operands are K-contiguous ("NT"): the tf.layers.Dense calls of the Transformer
// (open_seq2seq/parts/transformer/attention_layer.py:54-62,125-127,219, ffn_layer.py:51-85,
// the tied softmax of embedding_layer.py:90-105) and — with the transposed weight copies the
// parameter store keeps next to every bf16 weight — their data gradients. The weight gradient
// (reduction over the rows of both operands) is the K = 1 case of conv1d_wgrad.
//
// Same ping-pong structure as conv1d_pp_kernel: tile 256 x 256, 8 waves, waves 0-3 own rows
// 0..127, waves 4-7 rows 128..255 (wave tile 128 x 64, 128 accumulator registers); wave w and
// wave w+4 share a SIMD and alternate between issuing 16 MFMA 32x32x16 (COMPUTE) and fetching
// fragments from LDS + issuing LDS-DMA (LOAD), one s_barrier per slot. A step is 64 deep; an
// item is one 64-row half of the wave tile:
//
//   LOAD(2s)   : W fragments of step s (kept for both items) + A fragments of rows 0..63;
//                issue the group's OWN half of the A tile of step s+2 (ring of 3: nobody else
//                reads that half, its previous content was last read one step ago)
//   LOAD(2s+1) : A fragments of rows 64..127; drain everything but the 4 DMA just issued
//                (counted vmcnt); issue the W tile of step s+2 (ring of 2: W is read in even
//                slots only)
//
// Every LOAD slot carries exactly 4 LDS-DMA instructions per wave (16 KB per CU per slot): a
// 256 x 256 tile moves 64 KB per step through the L2->LDS path (128 FLOP per byte — there is no
// tap reuse as in the convolution), and a CU keeps only ~16 KB of such requests in flight; issued
// as one burst per step the same traffic ran at 0.5x the speed.
#include <mutex>
#include <type_traits>

#include "conv1d_common.hpp"
#include "os2s_split_reduce.hpp"

namespace os2s {

__device__ __forceinline__ void gpp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One 256 x 256 tile: main loop + the convolution's fused epilogue. a_half / a_bytes describe the
// 128 rows this wave's GROUP owns (first row, bytes that may be read from there: rows past it are
// out of the descriptor's range and read as zeros); everything else comes from p: x_st = row
// stride of A, Cin = reduction length (multiple of 64, nchunks = Cin / 64), w = W [Cout, Cin].
// The tile covers the 64-deep steps c_beg .. c_beg + nsteps - 1 of the reduction; a piece of a split
// unit (sp.f > 1) publishes its fp32 partial tile and only the last arriver runs the epilogue.
struct GppSplit { float* slab0; int* ticket; int piece, f; };
__device__ __forceinline__ void gpp_tile(const ConvArgs& p, const bf16_t* a_half, long long a_bytes, int n0,
                                         const int (&wb)[2], const int (&wt0)[2], const int (&wmid)[2],
                                         char* smem, int c_beg, int nsteps, const GppSplit& sp,
                                         unsigned long long* stamps = nullptr) {
  constexpr int BM = 128, BN = 256, NWIN = 2, WM = 2, WN = 4;
  constexpr int MI = 4, NI = 2;
  constexpr int TILE = 256 * 128;                        // one operand tile: 256 rows x 64 k (bf16)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;
  char* const abuf0 = smem;                              // A tiles: ring of 3 x 32 KB
  char* const wbuf0 = smem + 3 * TILE;                   // W tiles: ring of 2 x 32 KB
  // LDS-DMA through buffer descriptors: rows past the end of A / W are out of range of the
  // descriptor and read as zeros; the k position of the tile sits in the scalar offset
  __amdgpu_buffer_rsrc_t ars;
  {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)a_half);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)a_half >> 32));
    const long long cl = a_bytes < 0 ? 0 : (a_bytes < 0x7fffffffll ? a_bytes : 0x7fffffffll);
    const int bytes = __builtin_amdgcn_readfirstlane((int)cl);
    ars = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  }
  const long long w_bytes = (long long)(p.Cout - n0) * p.Cin * 2;
  const unsigned long long wbs = (unsigned long long)p.w + (unsigned long long)n0 * (unsigned long long)p.Cin * 2ull;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wbs, 0, (int)(w_bytes < 0x7fffffffll ? w_bytes : 0x7fffffffll), 0x00020000);
  int av[4], wv[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    // A: 16 instructions per 128-row half, 4 per wave of the group; W: 32 per tile, 4 per wave
    const int arow = (pi * 4 + wn) * 8 + (lane >> 3), wrow = (pi * 8 + wid) * 8 + (lane >> 3);
    const int jj = lane & 7;
    av[pi] = (arow * (int)p.x_st + (jj ^ ((arow >> 1) & 7)) * 8) * 2;
    wv[pi] = (wrow * p.Cin + (jj ^ ((wrow >> 1) & 7)) * 8) * 2;
  }
  auto stage_a = [&](int kt, int buf) {                  // this group's half of the A tile
    const int soff = __builtin_amdgcn_readfirstlane(kt * 128);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          ars, (__attribute__((address_space(3))) void*)(abuf0 + buf * TILE + grp * (TILE / 2) + (pi * 4 + wn) * 1024),
          16, av[pi], soff, 0, 0);
  };
  auto stage_w = [&](int kt, int buf) {
    const int soff = __builtin_amdgcn_readfirstlane(kt * 128);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(wbuf0 + buf * TILE + (pi * 8 + wid) * 1024), 16,
          wv[pi], soff, 0, 0);
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int im = 0; im < MI; ++im)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[in][im][e] = 0.f;

  {
    const int l31 = lane & 31, lhi = lane >> 5;
    // fragment offsets inside a tile: 16-B slot (kk*2 + lhi) ^ swizzle(row)
    int aoff[4], woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int sw = ((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4;
      aoff[kk] = (grp * 128 + l31) * 128 + sw;
      woff[kk] = (wn * 64 + l31) * 128 + sw;
    }
    // only step 0 has to land before the loop starts (a CU keeps ~16 KB of LDS-DMA in flight: the
    // 128 KB of two steps were 8-10 k cycles of pipeline fill); the tiles of step 1 are drained by
    // the counted wait of LOAD(1) like those of every later step
    stage_a(c_beg, 0);
    stage_w(c_beg, 0);
    if (nsteps > 1) {
      stage_a(c_beg + 1, 1);
      stage_w(c_beg + 1, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    gpp_barrier();                                       // (__syncthreads() would drain vmcnt(0) again)
    if (stamps && tid == 0) stamps[2] = __builtin_readcyclecounter();
    if (grp) gpp_barrier();                              // group B runs one slot behind group A

    int ai = 0;                                          // ring slot of the current A tile
    auto step = [&](auto PAR, int s) {
      constexpr int PB = decltype(PAR)::value;
      const char* const as = abuf0 + ai * TILE;
      const char* const ws = wbuf0 + PB * TILE;
      const bool more = s + 2 < nsteps;
      bf16x8 wf[NI][4], af[2][4];
      // ---- LOAD(2s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int in = 0; in < NI; ++in)
          wf[in][kk] = *reinterpret_cast<const bf16x8*>(ws + woff[kk] + in * 4096);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          af[i2][kk] = *reinterpret_cast<const bf16x8*>(as + aoff[kk] + i2 * 4096);
      }
      {
        int a2 = ai + 2;
        a2 = a2 >= 3 ? a2 - 3 : a2;
        if (more) stage_a(c_beg + s + 2, a2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gpp_barrier();
      // ---- COMPUTE(2s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < NI; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            acc[in][i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], af[i2][kk], acc[in][i2], 0, 0, 0);
      gpp_barrier();
      // ---- LOAD(2s+1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          af[i2][kk] = *reinterpret_cast<const bf16x8*>(as + aoff[kk] + (2 + i2) * 4096);
      // everything older than the 4 A-tile instructions issued in LOAD(2s) must have landed: the
      // W tile of step s+1 (issued a step ago) is read right after the next barrier pair
      if (more) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        stage_w(c_beg + s + 2, PB);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ai = ai + 1 >= 3 ? 0 : ai + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gpp_barrier();
      // ---- COMPUTE(2s+1)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < NI; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            acc[in][2 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], af[i2][kk], acc[in][2 + i2], 0, 0, 0);
      gpp_barrier();
    };
    for (int s = 0; s < nsteps; s += 2) {
      step(std::integral_constant<int, 0>{}, s);
      if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
    }
    if (!grp) gpp_barrier();
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[3] = __builtin_readcyclecounter();
  if (sp.f > 1) {
    auto at = [&](int v) -> f32x16& { return acc[v >> 2][v & 3]; };
    if (!split_publish_and_reduce(at, sp.slab0, sp.ticket, sp.piece, sp.f, smem, tid)) return;
  }
  conv_epilogue<BM, BN, WM, WN, NWIN>(p, acc, smem, tid, lane, wid, wmid, n0, wb, wt0);
  if (stamps) {
    __syncthreads();
    if (tid == 0) stamps[4] = __builtin_readcyclecounter();
  }
}

// ConvArgs is used as the argument block so that the fused epilogue is literally the convolution's:
// B = 1, Tout = M rows, Cin = K, Cout = N, x = A (row stride x_st), w = W [N, K] contiguous.
// MT = number of 128-row windows, MT8 = 256-row blocks per XCD in the main part, NT = 256-column
// tiles. Units are ranked 0 .. U-1; the units at rank >= nfull (the last partial round of
// workgroups, chosen on the host by the cost model of os2s_split_reduce.hpp) are cut f ways along
// K: a vocabulary-sized reduction over few output tiles (the data gradient of a 32 k softmax into
// a 512-wide layer: 50 tiles x 512 steps) otherwise runs on a fifth of the chip.
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(ConvArgs p, int gm, int rem, int nfull, int f) {
  constexpr int BM = 128, BN = 256, NWIN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2;
  const int bid = blockIdx.x;
  int rank = bid, piece = 0, npiece = 1;
  if (bid >= nfull) {
    const int i = bid - nfull;
    rank = nfull + i / f;
    piece = i - (i / f) * f;
    npiece = f;
  }
  const int u_main = 8 * p.MT8 * p.NT;
  int m_blk, n_idx;
  if (rank < u_main) {
    // ---- main part, 8 * MT8 row blocks: per XCD, groups of gm row blocks sweep the n-tiles together
    // (rows of A stay in that L2, every weight panel is shared by gm workgroups) -------------------
    const int xcd = rank & 7, loc = rank >> 3;
    const int mg = loc / (p.NT * gm), rr = loc - mg * (p.NT * gm);
    const int left = p.MT8 - mg * gm, gl = left < gm ? left : gm;   // blocks in this group
    n_idx = rr / gl;
    m_blk = (mg * gm + (rr - n_idx * gl)) * 8 + xcd;
  } else {
    // ---- the last rem < 8 row blocks: their tiles go round the XCDs one by one (given to XCD 0..rem-1
    // as whole row blocks, 8300 rows = 33 blocks put 5 blocks on XCD 0 and 4 on the others: the
    // 32768-column vocabulary GEMM ran 20 rounds of tiles on XCD 0 and 16 elsewhere) -----------------
    const int t = rank - u_main;
    n_idx = t / rem;
    m_blk = p.MT8 * 8 + (t - n_idx * rem);
  }
  const int m_first = m_blk * NWIN;
  const int n0 = n_idx * BN;
  int wb[NWIN], wt0[NWIN], wmid[NWIN];
#pragma unroll
  for (int w = 0; w < NWIN; ++w) {
    wb[w] = 0;
    wt0[w] = (m_first + w) * BM;
    wmid[w] = (m_first + w < p.MT) ? m_first + w : -1;
  }
  const int m0 = m_first * BM + grp * BM;                // first row of this group's half
  const int c_beg = __builtin_amdgcn_readfirstlane(piece * p.nchunks / npiece);
  const int c_end = __builtin_amdgcn_readfirstlane((piece + 1) * p.nchunks / npiece);
  GppSplit sp;
  sp.f = npiece; sp.piece = piece;
  sp.slab0 = npiece > 1 ? p.ws_slabs + (size_t)(rank - nfull) * f * kSplitSlabFloats : nullptr;
  sp.ticket = npiece > 1 ? p.ws_cnt + (rank - nfull) : nullptr;
  gpp_tile(p, p.x + (long long)m0 * p.x_st, (long long)(p.Tout - m0) * p.x_st * 2, n0, wb, wt0, wmid, smem,
           c_beg, c_end - c_beg, sp);
}

// ---------------------------------------------------------------------------------------------
// Round 6: a SHORTER tile for the launches the 256-row tile cannot spread over the chip. The Transformer's batch is
// 8 192 +- 222 packed tokens: 33 row blocks of 256, so every N = 1024 product — 90 of the 129 GEMM launches of a
// Transformer-big step — is 132 tiles on 256 CUs. 160 rows (five MFMA row blocks) make it 52 x 4 = 208 tiles, still one
// round. Five row blocks do not split over the two wave groups of the tile above, so the 8 waves split the COLUMNS
// instead: wave w owns all RB * 32 rows x columns [32 w, 32 w + 32) (RB accumulator blocks), group A = waves 0-3,
// group B = waves 4-7; wave w and w + 4 share a SIMD and alternate, slot by slot, between the RB * 4 MFMAs of a 64-deep
// step (COMPUTE) and its 4 * (RB + 1) fragment reads + ALL of its LDS-DMA (LOAD) — two slots per step, not four:
//
//   LOAD(s)    : W fragments (4) and A fragments (RB * 4) of step s; issue this wave's share of step s + 2 (W: 4
//                instructions, A: 3 or 2) into ring slot (s + 2) % 3 — last read a step ago by the other group,
//                which finished before the previous barrier; wait until only these stay in flight (this wave's
//                share of step s + 1 has landed: the other group reads it two barriers later)
//   COMPUTE(s) : RB * 4 MFMA 32x32x16
//
// Rings of three for both operands: 3 x (RB * 4 KB) + 3 x 32 KB = 156 KB at RB = 5. The reduction runs in the same
// order as in the 256-row tile: results are bit-identical to it. One round of tiles only (no tail split): the
// launcher takes this tile when the 256-row tiling would leave more than 40 % of the CUs without a tile.
// ---------------------------------------------------------------------------------------------
template <int RB>
__global__ __launch_bounds__(512, 2) void gemm_pp_cols_kernel(ConvArgs p, int mblocks) {
  constexpr int BMT = RB * 32, BN = 256;
  constexpr int ATILE = BMT * 128, WTILE = 256 * 128;     // bytes of one 64-deep operand tile
  constexpr int AINSTR = RB * 4;                          // LDS-DMA instructions per A tile (8 rows x 128 B each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int U = mblocks * p.NT;
  if ((int)blockIdx.x >= U) return;
  // consecutive ranks = the column tiles of one row block: behind one XCD's L2 (contiguous rank range per XCD)
  int rank;
  {
    const int q = U >> 3, r = U & 7, x = blockIdx.x & 7;
    rank = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
  }
  const int m_blk = rank / p.NT, n_idx = rank - m_blk * p.NT;
  const int m0 = m_blk * BMT, n0 = n_idx * BN;
  char* const abuf0 = smem;
  char* const wbuf0 = smem + 3 * ATILE;
  const bf16_t* const a0 = p.x + (long long)m0 * p.x_st;
  __amdgpu_buffer_rsrc_t ars;
  {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)a0);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)a0 >> 32));
    const long long ab = (long long)(p.Tout - m0) * p.x_st * 2;
    const long long cl = ab < 0 ? 0 : (ab < 0x7fffffffll ? ab : 0x7fffffffll);
    ars = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                            __builtin_amdgcn_readfirstlane((int)cl), 0x00020000);
  }
  const long long w_bytes = (long long)(p.Cout - n0) * p.Cin * 2;
  const unsigned long long wbs = (unsigned long long)p.w + (unsigned long long)n0 * (unsigned long long)p.Cin * 2ull;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wbs, 0, (int)(w_bytes < 0x7fffffffll ? w_bytes : 0x7fffffffll), 0x00020000);
  // this wave's DMA instructions: W 4 (of 32), A up to 3 (of RB * 4): instruction i covers tile rows 8 i .. 8 i + 7
  int av[3], wv[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    const int wrow = (pi * 8 + wid) * 8 + (lane >> 3), jj = lane & 7;
    wv[pi] = (wrow * p.Cin + (jj ^ ((wrow >> 1) & 7)) * 8) * 2;
    if (pi < 3) {
      const int arow = (pi * 8 + wid) * 8 + (lane >> 3);
      av[pi] = (arow * (int)p.x_st + (jj ^ ((arow >> 1) & 7)) * 8) * 2;
    }
  }
  const bool a3 = 16 + wid < AINSTR;                      // waves 0-3 carry a third A instruction at RB = 5
  auto stage = [&](int kt, int buf) {
    const int soff = __builtin_amdgcn_readfirstlane(kt * 128);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(wbuf0 + buf * WTILE + (pi * 8 + wid) * 1024), 16,
          wv[pi], soff, 0, 0);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          ars, (__attribute__((address_space(3))) void*)(abuf0 + buf * ATILE + (pi * 8 + wid) * 1024), 16,
          av[pi], soff, 0, 0);
    if (a3)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          ars, (__attribute__((address_space(3))) void*)(abuf0 + buf * ATILE + (16 + wid) * 1024), 16,
          av[2], soff, 0, 0);
  };
  static_assert(AINSTR > 16 && AINSTR <= 24, "the A tile takes 2 - 3 instructions per wave");

  f32x16 acc[1][RB];
#pragma unroll
  for (int im = 0; im < RB; ++im)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][im][e] = 0.f;

  const int nsteps = p.nchunks;
  {
    const int l31 = lane & 31, lhi = lane >> 5;
    int aoff[4], woff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int sw = ((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4;
      aoff[kk] = l31 * 128 + sw;
      woff[kk] = (wid * 32 + l31) * 128 + sw;
    }
    stage(0, 0);
    if (nsteps > 1) {
      stage(1, 1);
      if (a3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    gpp_barrier();
    if (grp) gpp_barrier();                              // group B runs one slot behind group A
    int bi = 0;                                          // ring slot of the current step
    for (int s = 0; s < nsteps; ++s) {
      const char* const as = abuf0 + bi * ATILE;
      const char* const ws = wbuf0 + bi * WTILE;
      bf16x8 wf[4], af[RB][4];
      // ---- LOAD(s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        wf[kk] = *reinterpret_cast<const bf16x8*>(ws + woff[kk]);
#pragma unroll
        for (int im = 0; im < RB; ++im)
          af[im][kk] = *reinterpret_cast<const bf16x8*>(as + aoff[kk] + im * 4096);
      }
      if (s + 2 < nsteps) {
        int b2 = bi + 2;
        b2 = b2 >= 3 ? b2 - 3 : b2;
        stage(s + 2, b2);
        if (a3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bi = bi + 1 >= 3 ? 0 : bi + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gpp_barrier();
      // ---- COMPUTE(s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int im = 0; im < RB; ++im)
          acc[0][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], af[im][kk], acc[0][im], 0, 0, 0);
      gpp_barrier();
    }
    if (!grp) gpp_barrier();
  }
  __syncthreads();
  const int wb[1] = {0}, wt0[1] = {m0}, wmid[1] = {m_blk};
  conv_epilogue<BMT, BN, 1, 8, 1>(p, acc, smem, tid, lane, wid, wmid, n0, wb, wt0);
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolutions of a ragged batch on the same tile: up to kMaxConvGroups independent
// convolutions with the same batch geometry and lengths (the dense-residual branches of a Jasper
// block end and their data gradients, conv_blocks.py:78-85; a single group = a plain K = 1 layer)
// in ONE grid. A 128-row half of a tile is a LIVE window (b, t0) of the batch: the windows with
// at least one real row are enumerated on the device from in_len / out_len (one wave scan,
// B <= 64) and paired over the compacted list, exactly as in conv1d_pp_kernel; rows at or past
// in_len[b] are outside the group's descriptor and read as zeros. Dead windows of a forward call
// get their zero rows / zero BatchNorm partials from store-only workgroups at the end of the grid.
// Unit = (group, window pair, 256-column tile); units of one pair sit on one XCD (its rows are
// fetched into one L2), consecutive workgroups hold different pairs.
// ---------------------------------------------------------------------------------------------
constexpr int kGppZeroWin = 2;
#ifdef OS2S_EPI_STAMPS
constexpr int kGppStampStride = 24;
#else
constexpr int kGppStampStride = 8;
#endif

__global__ __launch_bounds__(512, 2) void conv1x1_pp_kernel(ConvArgs p, ConvGroupTable gt) {
  constexpr int BM = 128, BN = 256, NWIN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  // experiment hook: phase time stamps of the first 2048 work units (8 x uint64 each: entry,
  // unit decoded, pipeline filled, main loop done, epilogue done, steps)
  unsigned long long* const stamps = (p.dbg && blockIdx.x < 2048) ? p.dbg + (size_t)blockIdx.x * kGppStampStride : nullptr;
  if (stamps && tid == 0) stamps[0] = __builtin_readcyclecounter();
  // ---- live windows per sample, inclusive scan over the batch (one value per lane) ----------
  int nw = 0;
  if (lane < p.B) {
    int len = p.Tin;
    if (p.in_len) { const int l = p.in_len[lane]; len = l < 0 ? 0 : (l < len ? l : len); }
    if (p.out_len) { const int l = p.out_len[lane]; len = l < 0 ? 0 : (l < len ? l : len); }
    nw = (len + BM - 1) / BM;
    nw = nw < p.mtiles_per_b ? nw : p.mtiles_per_b;
  }
  int scan = nw;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(scan, o, 64);
    if (lane >= o) scan += t;
  }
  const int L = __builtin_amdgcn_readlane(scan, 63);
  const int P = (L + 1) >> 1;
  const int nwork = P * gt.total_tiles;                  // total_tiles = sum of the groups' n-tiles
  const int bid = blockIdx.x;

  if (bid >= nwork) {
    // ---- store-only role (forward calls): zero rows + zero BN partials of the dead windows ----
    const int z = (int)gridDim.x - 1 - bid;
    const int zper = (p.MT + kGppZeroWin - 1) / kGppZeroWin;
    if (p.out_len || z >= zper * gt.ngroups) return;
    const int gi = z / zper, zb = z - gi * zper;
    ConvGroup g = gt.g[0];
#pragma unroll
    for (int i = 1; i < kMaxConvGroups; ++i)
      if (i == gi) g = gt.g[i];
    for (int m = zb * kGppZeroWin; m < (zb + 1) * kGppZeroWin && m < p.MT; ++m) {
      const int b = m / p.mtiles_per_b, j = m - b * p.mtiles_per_b;
      if (j < __shfl(nw, b, 64)) continue;
      const int t0 = j * BM, rows = min(BM, p.Tout - t0);
      if (!g.accumulate) {
        const int yst = g.y_st ? g.y_st : g.Cout, c8n = g.Cout >> 3;
        bf16_t* const yb = reinterpret_cast<bf16_t*>(g.y) + ((long long)b * p.Tout + t0) * yst;
        const u32x4 zv = {0u, 0u, 0u, 0u};
        if (yst == g.Cout) {
          for (int e = tid; e < rows * c8n; e += 512) *reinterpret_cast<u32x4*>(yb + (long long)e * 8) = zv;
        } else {
          for (int e = tid; e < rows * c8n; e += 512)
            *reinterpret_cast<u32x4*>(yb + (long long)(e / c8n) * yst + (e % c8n) * 8) = zv;
        }
      }
      if (g.stats)
        for (int e = tid; e < 2 * g.Cout; e += 512) g.stats[(long long)m * 2 * g.Cout + e] = 0.f;
    }
    return;
  }

  // ---- work role: bid -> (group, unit rank inside the group) -> (window pair, n-tile) ---------
  int gi = 0;
#pragma unroll
  for (int i = 1; i < kMaxConvGroups; ++i)
    if (i < gt.ngroups && bid >= gt.g[i].tile_begin * P) gi = i;
  ConvGroup g = gt.g[0];
#pragma unroll
  for (int i = 1; i < kMaxConvGroups; ++i)
    if (i == gi) g = gt.g[i];
  p.x = g.x; p.w = g.w; p.y = g.y; p.stats = g.stats;
  p.Cin = g.Cin; p.Cout = g.Cout; p.accumulate = g.accumulate;
  p.x_st = g.x_st ? g.x_st : g.Cin; p.x_sb = (long long)p.Tin * p.x_st;
  p.y_st = g.y_st ? g.y_st : g.Cout; p.y_sb = (long long)p.Tout * p.y_st;
  p.NT = (g.Cout + BN - 1) / BN;
  p.nchunks = g.Cin / 64;
  const int rank = bid - g.tile_begin * P;
  int xcd, loc;
  {
    const int P8 = (P + 7) >> 3, rem = P - 8 * (P8 - 1), base = (P8 - 1) * p.NT * 8;
    if (rank < base) { xcd = rank & 7; loc = rank >> 3; }
    else { const int i = rank - base; loc = (P8 - 1) * p.NT + i / rem; xcd = i - (i / rem) * rem; }
  }
  const int pair = (loc / p.NT) * 8 + xcd;
  const int n_idx = loc - (loc / p.NT) * p.NT;
  int wb[NWIN], wt0[NWIN], wlen[NWIN], wmid[NWIN];
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
    wlen[w] = live ? len_b : 0;                          // dead slot: every row reads as zero
    wmid[w] = live ? b * p.mtiles_per_b + (i - before) : -1;
  }
  const int b_own = grp ? wb[1] : wb[0], t_own = grp ? wt0[1] : wt0[0], len_own = grp ? wlen[1] : wlen[0];
  if (stamps && tid == 0) { stamps[1] = __builtin_readcyclecounter(); stamps[5] = (unsigned long long)p.nchunks; }
  GppSplit sp;
  sp.f = 1; sp.piece = 0; sp.slab0 = nullptr; sp.ticket = nullptr;
  gpp_tile(p, p.x + (long long)b_own * p.x_sb + (long long)t_own * p.x_st,
           (long long)(len_own - t_own) * p.x_st * 2, n_idx * BN, wb, wt0, wmid, smem, 0, p.nchunks, sp, stamps);
}

// Host side of conv1x1_pp_kernel. `a` carries the batch geometry (B, Tin = Tout = T, in_len, out_len)
// and the epilogue switches shared by all groups; groups[i].tile_begin is filled here.
int launch_conv1x1_pp(hipStream_t stream, ConvArgs a, ConvGroupTable gt) {
  if (a.B > 64 || a.K != 1 || a.stride != 1 || a.padL != 0 || a.Tin != a.Tout || a.out_f32)
    return OS2S_ERR_UNSUPPORTED;
  int nt = 0;
  for (int i = 0; i < gt.ngroups; ++i) {
    const ConvGroup& g = gt.g[i];
    if (g.Cin % 64 != 0 || g.Cin < 64 || g.Cout % 8 != 0) return OS2S_ERR_UNSUPPORTED;
    if ((long long)(g.x_st ? g.x_st : g.Cin) * 2 * 256 >= (1ll << 31)) return OS2S_ERR_UNSUPPORTED;
    if (g.x_st % 8 != 0 || g.y_st % 8 != 0 || (g.x_st && g.x_st < g.Cin) || (g.y_st && g.y_st < g.Cout))
      return OS2S_ERR_UNSUPPORTED;
    gt.g[i].tile_begin = nt;
    nt += ceil_div(g.Cout, 256);
  }
  for (int i = gt.ngroups; i < kMaxConvGroups; ++i) gt.g[i] = gt.g[0];
  gt.total_tiles = nt;
  a.mtiles_per_b = ceil_div(a.Tout, 128);
  a.MT = a.B * a.mtiles_per_b;
  a.R = 128; a.Rpad = 128;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)conv1x1_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  const int pmax = ceil_div(a.MT, 2);
  const int nzero = a.out_len ? 0 : ceil_div(a.MT, kGppZeroWin) * gt.ngroups;
  const size_t smem = (size_t)5 * 256 * 128;            // A ring of 3 + W ring of 2 (> epilogue staging)
  OS2S_LAUNCH(conv1x1_pp_kernel, dim3(pmax * nt + nzero), dim3(512), smem, stream, a, gt);
  return OS2S_OK;
}

}  // namespace os2s

static int g_gemm_split = -1;
static int g_gemm_tile = 0;        // gemm_nt.tile: 0 = by shape, 256 = the 256-row tile always, 160 = the 160-row tile whenever legal

// C[M,N] = A[M,K] . W[N,K]^T with the fused epilogue C = residual + dropout(act(. + bias)),
// optional accumulation into C (bf16) and fp32 output. lda / ldc / residual row stride in
// elements; W is contiguous [N, K]. With a workspace (os2s_conv1d_workspace_bytes(), zero tickets,
// one per stream) the last partial round of tiles is cut along K when the cost model says so.
static int gemm_nt_impl(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W,
                        void* C, long long ldc, int M, int N, int K, const float* bias, int act,
                        float keep_prob, unsigned long long seed, const uint16_t* residual,
                        int accumulate, int out_f32, void* workspace, size_t workspace_bytes,
                        const uint16_t* mask_ref, float mask_scale, float* stats) {
  using namespace os2s;
  OS2S_REQUIRE(A && W && C && M >= 1 && N >= 1 && K >= 64 && K % 64 == 0);
  OS2S_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= N);
  OS2S_REQUIRE(act == 0 || act == 1 || act == 3);
  OS2S_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f);
  if (out_f32) OS2S_REQUIRE(act == 0 && keep_prob == 1.f && residual == nullptr);
  if (!out_f32) OS2S_REQUIRE(N % 8 == 0 && ldc % 8 == 0);
  OS2S_REQUIRE((long long)K * 2 * 256 < (1ll << 31) && lda * 2 * 256 < (1ll << 31));
  ConvArgs a;
  a.x = A; a.w = W; a.y = C; a.in_len = nullptr; a.out_len = nullptr; a.bias = bias; a.stats = nullptr;
  a.B = 1; a.Tin = M; a.Tout = M; a.Cin = K; a.Cout = N; a.K = 1; a.stride = 1; a.dil = 1; a.padL = 0;
  a.x_sb = 0; a.x_st = lda; a.y_sb = 0; a.y_st = ldc;
  a.out_f32 = out_f32; a.accumulate = accumulate; a.act = act; a.keep_prob = keep_prob; a.seed = seed;
  a.residual = residual;
  a.mask_ref = mask_ref; a.mask_scale = mask_scale; a.stats = stats; a.stat_ref = nullptr;
  if (mask_ref) OS2S_REQUIRE(!residual && !accumulate && !out_f32 && act == 0 && keep_prob == 1.f && !bias);
  if (stats) OS2S_REQUIRE(!out_f32);
  a.ws_slabs = nullptr; a.ws_cnt = nullptr; a.ws_nslabs = 0; a.ncu = 256; a.force_split = -1;
  a.dbg = nullptr; a.dbg_fixed_w = 0;
  a.mtiles_per_b = ceil_div(M, 128);
  a.MT = a.mtiles_per_b;
  a.NT = ceil_div(N, 256);
  a.nchunks = K / 64;
  a.R = 128; a.Rpad = 128;
  const int mblocks = ceil_div(a.MT, 2);
  const int full8 = mblocks / 8, rem = mblocks % 8;
  const int gm = full8 < 4 ? (full8 > 0 ? full8 : 1) : 4;
  a.MT8 = full8;
  const size_t main_bytes = (size_t)5 * 256 * 128;    // A ring of 3 + W ring of 2 = 160 KB
  const size_t epi_bytes = conv_epilogue_lds_bytes<128, 256, 2, 512>();
  const size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute((const void*)gemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n;
  });
  if (attr_rc != hipSuccess) return OS2S_ERR_LAUNCH;
  a.ncu = ncu;
  // ---- the 160-row tile (gemm_pp_cols_kernel<5>): when the 256-row tiling is ONE round that leaves more than 40 % of
  // the CUs without a tile and 160-row tiles still fit one round; per-window statistics keep the 128-row windows
  {
    const int mb160 = ceil_div(M, 160);
    const int u160 = mb160 * a.NT, u256 = mblocks * a.NT;
    // (not for launches of a few tiles — nothing to gain — and not while a test forces the tail split of the 256-row tile)
    const bool fits = !stats && u160 <= ncu && u256 * 10 <= ncu * 6 && u160 > u256 && u256 >= 32 && g_gemm_split <= 0;
    if (g_gemm_tile == 160 ? (!stats && u160 <= 4 * ncu) : (g_gemm_tile == 0 && fits)) {
      static std::once_flag once5;
      static hipError_t rc5 = hipSuccess;
      std::call_once(once5, [] {
        rc5 = hipFuncSetAttribute((const void*)gemm_pp_cols_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
      });
      if (rc5 != hipSuccess) return OS2S_ERR_LAUNCH;
      const size_t main5 = (size_t)3 * 160 * 128 + (size_t)3 * 256 * 128;
      const size_t epi5 = conv_epilogue_lds_bytes<160, 256, 1, 512>();
      OS2S_LAUNCH(gemm_pp_cols_kernel<5>, dim3(ceil_div(u160, 8) * 8), dim3(512), main5 > epi5 ? main5 : epi5,
                  (hipStream_t)stream, a, mb160);
      return OS2S_OK;
    }
  }
  // ---- tail split (decided here: nothing about the launch is only known on the device) ---------
  const int U = mblocks * a.NT;
  const int r = U % ncu;
  int f = 1;
  const size_t slab_bytes = (size_t)kSplitSlabFloats * 4;
  if (r > 0 && workspace && workspace_bytes >= kSplitTicketBytes + 2 * slab_bytes) {
    a.ws_cnt = reinterpret_cast<int*>(workspace);
    a.ws_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kSplitTicketBytes);
    size_t n = (workspace_bytes - kSplitTicketBytes) / slab_bytes;
    const size_t cap = (size_t)3 * ncu;
    a.ws_nslabs = (int)(n < cap ? n : cap);
    int fmax = a.nchunks / 8;                           // >= 8 steps per piece
    fmax = fmax > 16 ? 16 : fmax;
    // a round of whole units: ~1.05 us per 64-deep step + ~10 us of fill and epilogue
    if (g_gemm_split > 0) {
      f = g_gemm_split < fmax ? g_gemm_split : (fmax > 1 ? fmax : 1);
      while (f > 1 && r * f > a.ws_nslabs) --f;
    } else if (g_gemm_split < 0 && fmax > 1) {
      f = split_factor(r, ncu, 1.05f * a.nchunks + 10.f, fmax, a.ws_nslabs);
    }
  }
  const int nfull = f > 1 ? U - r : U;
  const int grid = nfull + (f > 1 ? r * f : 0);
  OS2S_LAUNCH(gemm_pp_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream, a, gm, rem > 0 ? rem : 1, nfull, f);
  return OS2S_OK;
}

extern "C" int os2s_gemm_nt_ws(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W,
                               void* C, long long ldc, int M, int N, int K, const float* bias, int act,
                               float keep_prob, unsigned long long seed, const uint16_t* residual,
                               int accumulate, int out_f32, void* workspace, size_t workspace_bytes) {
  return gemm_nt_impl(stream, A, lda, W, C, ldc, M, N, K, bias, act, keep_prob, seed, residual, accumulate,
                      out_f32, workspace, workspace_bytes, nullptr, 1.f, nullptr);
}

// C[M,N] = (A[M,K] . W[N,K]^T) * (mask_ref > 0 ? mask_scale : 0), bf16: the data gradient of a Dense
// layer whose INPUT is the output of a ReLU + dropout layer (ffn_layer.py:51-85), with that layer's
// activation / dropout backward applied in the epilogue — mask_ref is its saved forward output
// (zero where the ReLU or the dropout mask was off), mask_scale = 1 / keep_prob. stats (or null):
// [ceil(M/128), 2, N] per-window column sums (and sums of squares) of C = the partials of the bias
// gradient of that layer.
extern "C" int os2s_gemm_nt_mask_ws(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W,
                                    void* C, long long ldc, int M, int N, int K, const uint16_t* mask_ref,
                                    float mask_scale, float* stats, void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(mask_ref != nullptr);
  return gemm_nt_impl(stream, A, lda, W, C, ldc, M, N, K, nullptr, 0, 1.f, 0ull, nullptr, 0, 0, workspace,
                      workspace_bytes, mask_ref, mask_scale, stats);
}

extern "C" int os2s_gemm_nt(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W,
                            void* C, long long ldc, int M, int N, int K, const float* bias, int act,
                            float keep_prob, unsigned long long seed, const uint16_t* residual,
                            int accumulate, int out_f32) {
  return os2s_gemm_nt_ws(stream, A, lda, W, C, ldc, M, N, K, bias, act, keep_prob, seed, residual, accumulate,
                         out_f32, nullptr, 0);
}

// os2s_set_option("gemm_nt.split", f): > 0 forces the tail split factor, 0 disables the split, < 0 = cost model
static os2s::OptionReg r_gemm_split("gemm_nt.split", [](double v) { g_gemm_split = (int)v; });
// os2s_set_option("gemm_nt.tile", t): 0 = by shape, 256 = always the 256 x 256 tile, 160 = the 160 x 256 tile whenever legal
static os2s::OptionReg r_gemm_tile("gemm_nt.tile", [](double v) { g_gemm_tile = (int)v; });
