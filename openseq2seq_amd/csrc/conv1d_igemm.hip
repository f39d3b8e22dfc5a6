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
#include <array>
#include <map>
#include <mutex>

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
  int act;                      // 0 none, 1 relu
  float keep_prob;              // 1 = no dropout
  unsigned long long seed;
  const bf16_t* residual;       // same layout/strides as y (bf16 output only) or null
};

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// BM = rows of ONE time window (one batch item); a workgroup processes NWIN consecutive
// windows (possibly of different batch items) against the same weight stream, so the
// M extent of a block is NWIN*BM while the tile granularity in time stays BM.
template <int BM, int BN, int WM, int WN, int NWIN, bool XSINGLE = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN <= 4) ? (XSINGLE ? 3 : 2) : 1) void conv1d_igemm_kernel(
    ConvArgs p) {
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
  const int bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int n_idx = loc / p.MT8;
  const int m_first = ((loc - n_idx * p.MT8) * 8 + xcd) * NWIN;
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

  // ---- epilogue ------------------------------------------------------------
  if (p.out_f32) {
    // small/rare path (FC logits): scattered fp32 stores straight from registers
    const int b = my_b, t0 = my_t0;
    const int valid_rows = (m_first + my_win < p.MT) ? min(BM, p.Tout - t0) : 0;
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

  constexpr int OP = BN * 2 + 16;  // out-tile pitch in bytes
  // The out tile is staged through LDS (coalesced 16-B row stores + the BN partial sums). Wide
  // tiles do not fit all windows at once: stage EW windows per pass.
  constexpr bool EPI_SPLIT = (size_t)NWIN * BM * OP > 112 * 1024;
  constexpr int EW = EPI_SPLIT ? 1 : NWIN;
  char* const ot = smem;
#pragma unroll
  for (int w0 = 0; w0 < NWIN; w0 += EW) {
  __syncthreads();                 // staging buffers / previous pass are no longer read
  if (my_win >= w0 && my_win < w0 + EW) {
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
        }
        if (p.keep_prob < 1.f) {
          // same (seed, element index / 8) convention as the elementwise kernels
          const long long e0 = ((long long)b * p.Tout + t0 + tt) * p.Cout + n0 + cc;
          const uint32_t bits = dropout_bits8(p.seed, (unsigned long long)(e0 >> 3), p.keep_prob) >>
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
  }
  __syncthreads();

#pragma unroll
  for (int w = w0; w < w0 + EW; ++w) {
  const int b = wb[w], t0 = wt0[w];
  const int valid_rows = (m_first + w < p.MT) ? min(BM, p.Tout - t0) : 0;
  const char* const otw = ot + (w - w0) * BM * OP;
  bf16_t* const yb = reinterpret_cast<bf16_t*>(p.y) + (long long)b * p.y_sb;
  for (int q = tid; q < BM * (BN / 8); q += NTHR) {
    const int row = q / (BN / 8), c8 = q - row * (BN / 8);
    const int gc = n0 + c8 * 8;
    if (row < valid_rows && gc < p.Cout) {
      u32x4 v = *reinterpret_cast<const u32x4*>(otw + row * OP + c8 * 16);
      bf16_t* dst = yb + (long long)(t0 + row) * p.y_st + gc;
      if (p.residual) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(
            p.residual + (long long)b * p.y_sb + (long long)(t0 + row) * p.y_st + gc);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = pack2bf(bflo(v[e]) + bflo(o[e]), bfhi(v[e]) + bfhi(o[e]));
      }
      if (p.accumulate) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = pack2bf(bflo(v[e]) + bflo(o[e]), bfhi(v[e]) + bfhi(o[e]));
      }
      *reinterpret_cast<u32x4*>(dst) = v;
    }
  }
  }

  if (p.stats) {
    constexpr int CP = BN / 2;       // column pairs
    constexpr int RG = (NTHR / CP) > 0 ? (NTHR / CP) : 1;    // row groups
    const int cp = tid % CP, rg = tid / CP;
#pragma unroll
    for (int w = w0; w < w0 + EW; ++w) {
    if (m_first + w >= p.MT) break;
    const int m_idx = m_first + w;
    const int valid_rows = min(BM, p.Tout - wt0[w]);
    const char* const otw = ot + (w - w0) * BM * OP;
    if (w > w0) __syncthreads();      // scratch re-use between windows
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    if (rg < RG)
    for (int row = rg; row < valid_rows; row += RG) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(otw + row * OP + cp * 4);
      const float a = bflo(v), bb = bfhi(v);
      s0 += a; q0 += a * a;
      s1 += bb; q1 += bb * bb;
    }
    float* sc = reinterpret_cast<float*>(smem + EW * BM * OP);  // [RG][BN][2]
    if (rg < RG) {
    sc[(rg * BN + cp * 2 + 0) * 2 + 0] = s0;
    sc[(rg * BN + cp * 2 + 0) * 2 + 1] = q0;
    sc[(rg * BN + cp * 2 + 1) * 2 + 0] = s1;
    sc[(rg * BN + cp * 2 + 1) * 2 + 1] = q1;
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, qq = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        s += sc[(g * BN + tid) * 2 + 0];
        qq += sc[(g * BN + tid) * 2 + 1];
      }
      const int gc = n0 + tid;
      if (gc < p.Cout) {
        p.stats[((long long)m_idx * 2 + 0) * p.Cout + gc] = s;
        p.stats[((long long)m_idx * 2 + 1) * p.Cout + gc] = qq;
      }
    }
    }
  }
  }
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
  constexpr size_t kOP = BN * 2 + 16;
  constexpr int kEW = ((size_t)NWIN * BM * kOP > 112 * 1024) ? 1 : NWIN;
  constexpr int kRG = (NTHR / (BN / 2)) > 0 ? (NTHR / (BN / 2)) : 1;
  size_t epi_bytes = (size_t)kEW * BM * kOP + (size_t)kRG * BN * 2 * 4;
  size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  if (smem > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  static size_t attr_set = 0;
  if (smem > attr_set) {
    if (hipFuncSetAttribute((const void*)conv1d_igemm_kernel<BM, BN, WM, WN, NWIN, XSINGLE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return OS2S_ERR_LAUNCH;
    attr_set = 160 * 1024;
  }
  const int grid = a.MT8 * 8 * a.NT;
  OS2S_LAUNCH((conv1d_igemm_kernel<BM, BN, WM, WN, NWIN, XSINGLE>), dim3(grid), dim3(NTHR), smem,
              stream, a);
  return OS2S_OK;
}

}  // namespace os2s

// -1 (default) = pick per problem shape between the 128x128 tile (variant 3 / 0), the
// 256x256 tile (variant 5) and the 256x320 / 256x384 tiles (variants 8 / 9: Cout = 640 / 768
// in ONE round of workgroups) by timing them once — the lazily built kernel cache of the ABI
// contract; which one wins is decided by tile quantisation against the 256 CUs (e.g. B*T' =
// 28k rows: Cout 512 -> 224 256^2 tiles = one round at 975 TF/s vs 824; Cout 640 -> 336 tiles
// = 1.3 rounds at 712 vs 913).
static int g_conv_variant = -1;
// tuning hook (not part of the stable ABI surface used by the host layer)
extern "C" void os2s_conv1d_set_variant(int v) { g_conv_variant = v; }

extern "C" int os2s_conv1d_num_mtiles(int B, int Tout) {
  return B * os2s::ceil_div(Tout, os2s::kConvBM);
}

static int conv1d_fwd_impl(os2s_stream_t stream, const uint16_t* x, const uint16_t* w, void* y,
                           const int32_t* in_len, const float* bias, float* stats, int B,
                           int Tin, int Cin, int Cout, int K, int stride, int dil, int padL,
                           int Tout, long long y_stride_b, long long y_stride_t, int out_f32,
                           int accumulate, int act, float keep_prob, unsigned long long seed,
                           const uint16_t* residual, const int32_t* out_len);

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
                         seed, residual, out_len);
}

extern "C" int os2s_conv1d_fwd(os2s_stream_t stream, const uint16_t* x, const uint16_t* w,
                               void* y, const int32_t* in_len, const float* bias, float* stats,
                               int B, int Tin, int Cin, int Cout, int K, int stride, int dil,
                               int padL, int Tout, long long y_stride_b, long long y_stride_t,
                               int out_f32, int accumulate) {
  return conv1d_fwd_impl(stream, x, w, y, in_len, bias, stats, B, Tin, Cin, Cout, K, stride, dil,
                         padL, Tout, y_stride_b, y_stride_t, out_f32, accumulate, 0, 1.f, 0,
                         nullptr, nullptr);
}

static int conv1d_fwd_impl(os2s_stream_t stream, const uint16_t* x,
                           const uint16_t* w, void* y, const int32_t* in_len, const float* bias,
                           float* stats, int B, int Tin, int Cin, int Cout, int K, int stride,
                           int dil, int padL, int Tout, long long y_stride_b,
                           long long y_stride_t, int out_f32, int accumulate, int act,
                           float keep_prob, unsigned long long seed, const uint16_t* residual, const int32_t* out_len) {
  using namespace os2s;
  OS2S_REQUIRE(act == 0 || act == 1);
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
  if (g_conv_variant == -1) {
    const int base = K >= 8 ? 3 : 0;
    auto run = [&](int v) -> int {
      if (v == 5) return launch_conv<kConvBM, 256, 2, 4, 2, false>((hipStream_t)stream, a);
      if (v == 8) return launch_conv<kConvBM, 320, 4, 2, 2, true>((hipStream_t)stream, a);
      if (v == 9) return launch_conv<kConvBM, 384, 4, 2, 2, true>((hipStream_t)stream, a);
      if (v == 3) return launch_conv<kConvBM, kConvBN, 2, 2, 1, true>((hipStream_t)stream, a);
      return launch_conv<kConvBM, kConvBN, 2, 2, 1>((hipStream_t)stream, a);
    };
    const bool tunable = Cout >= 256 && !accumulate && !residual && !out_f32;
    static std::mutex mu;
    static std::map<std::array<int, 8>, int> cache;
    const std::array<int, 8> key = {B, Tin, Cin, Cout, K, stride, dil, Tout};
    int choice = -1;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = cache.find(key);
      if (it != cache.end()) choice = it->second;
    }
    if (choice < 0 && Cout < 256) choice = base;
    if (choice < 0 && !tunable) return run(base);     // decided by a later tunable call of this shape
    if (choice < 0) {
      float best = 1e30f;
      choice = base;
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return run(base);
      // other streams (the weight-gradient side stream of the host layer) must be idle while the
      // candidates are timed; tuning happens once per shape, outside any graph capture
      hipDeviceSynchronize();
      for (int v : {base, 5, 8, 9}) {
        if (v == 8 && Cout < 512) continue;           // 256x320 / 256x384 tiles: wide layers only
        if (v == 9 && Cout < 640) continue;
        if (run(v) != OS2S_OK) continue;              // warm-up (also: unsupported LDS size)
        for (int rep = 0; rep < 3; ++rep) {           // best of three timings: the clock ramps
          hipEventRecord(e0, (hipStream_t)stream);
          for (int r = 0; r < 3; ++r) run(v);
          hipEventRecord(e1, (hipStream_t)stream);
          hipEventSynchronize(e1);
          float ms = 0.f;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms > 0.f && ms < best) { best = ms; choice = v; }
        }
      }
      hipEventDestroy(e0);
      hipEventDestroy(e1);
      std::lock_guard<std::mutex> lk(mu);
      cache[key] = choice;
    }
    return run(choice);
  }
  if (g_conv_variant == 3 && K >= 8) {
    // single-buffered X window: 3 workgroups per CU
    return launch_conv<kConvBM, kConvBN, 2, 2, 1, true>((hipStream_t)stream, a);
  }
  if (g_conv_variant >= 4 && g_conv_variant <= 6) {
    // experiments with 256-wide output-channel tiles
    int rc = OS2S_ERR_UNSUPPORTED;
    if (g_conv_variant == 4) rc = launch_conv<kConvBM, 256, 2, 4, 2, true>((hipStream_t)stream, a);
    if (g_conv_variant == 5) rc = launch_conv<kConvBM, 256, 2, 4, 2, false>((hipStream_t)stream, a);
    if (g_conv_variant == 6) rc = launch_conv<kConvBM, 256, 2, 2, 1, true>((hipStream_t)stream, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  if (g_conv_variant == 9)
    return launch_conv<kConvBM, 384, 4, 2, 2, true>((hipStream_t)stream, a);
  if (g_conv_variant == 7 || g_conv_variant == 8) {
    // 256 x 384 / 256 x 320 tiles: one round of workgroups for Cout = 768 / 640
    int rc = OS2S_ERR_UNSUPPORTED;
    if (g_conv_variant == 7) rc = launch_conv<kConvBM, 384, 2, 4, 2, true>((hipStream_t)stream, a);
    if (g_conv_variant == 8) rc = launch_conv<kConvBM, 320, 4, 2, 2, true>((hipStream_t)stream, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  if (g_conv_variant == 2) {
    // two windows, 4 waves, 128x64 wave tiles (0.75 LDS fragment reads per MFMA)
    const int rc = launch_conv<kConvBM, kConvBN, 2, 2, 2>((hipStream_t)stream, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  if (g_conv_variant == 1) {
    // two 128-row windows per 8-wave workgroup; falls back when the double-buffered
    // windows do not fit in LDS (large stride / very long kernels)
    const int rc = launch_conv<kConvBM, kConvBN, 4, 2, 2>((hipStream_t)stream, a);
    if (rc != OS2S_ERR_UNSUPPORTED) return rc;
  }
  return launch_conv<kConvBM, kConvBN, 2, 2, 1>((hipStream_t)stream, a);
}
