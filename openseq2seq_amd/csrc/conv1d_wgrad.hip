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

#include "os2s_common.hpp"

namespace os2s {

struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  const int32_t* in_len;
  int B, Tin, Tout, Cin, Cout, K, stride, dil, padL;
  int NCO, NCI, NTP, NSPLIT, steps_per_split, use_atomic;
  int xrows, xrows_pad;  // X window rows per 64-step (and padded to x4)
  long long x_ld;        // x row stride in elements (>= Cin; channel slice of a wider tensor)
};

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
template <int TAPS, int COT>
__global__ __launch_bounds__(COT * 2, (COT == 128) ? 2 : 1) void conv1d_wgrad_kernel(WgradArgs p) {
  constexpr int BT = 64;  // reduction rows per step
  constexpr int NW = COT / 32, NTHR = NW * 64, NSUB = COT / 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;  // wave tile 64 (co) x 64 (ci)
  const int ysub = wm >> 1, wmi = wm & 1;  // 128-channel dY sub-image, 64-row half inside it

  const int bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int unit = (loc / p.NTP) * 8 + xcd;
  const int tp = loc - (loc / p.NTP) * p.NTP;
  const int nunits = p.NCO * p.NCI * p.NSPLIT;
  if (unit >= nunits) return;
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

}  // namespace os2s

// accumulate: 0 = dW is overwritten (one batch split) / must be pre-zeroed by
// the caller when the kernel decides to split; to keep the contract simple the
// caller always passes a buffer that is either zero or holds the running sum
// it wants to add to, and sets accumulate accordingly:
//   accumulate = 0 : dW = grad       (kernel never splits the batch)
//   accumulate = 1 : dW += grad      (atomics; batch may be split for occupancy)
extern "C" int os2s_conv1d_wgrad_ex(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                                    const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                                    int Tin, int Cin, int Cout, int K, int stride, int dil,
                                    int padL, int Tout, int accumulate);

extern "C" int os2s_conv1d_wgrad(os2s_stream_t stream, const uint16_t* x,
                                 const uint16_t* dy, float* dw,
                                 const int32_t* in_len, int B, int Tin, int Cin,
                                 int Cout, int K, int stride, int dil, int padL,
                                 int Tout, int accumulate) {
  return os2s_conv1d_wgrad_ex(stream, x, Cin, dy, dw, in_len, B, Tin, Cin, Cout, K, stride, dil,
                              padL, Tout, accumulate);
}

extern "C" int os2s_conv1d_wgrad_ex(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                                    const uint16_t* dy, float* dw, const int32_t* in_len, int B,
                                    int Tin, int Cin, int Cout, int K, int stride, int dil,
                                    int padL, int Tout, int accumulate) {
  using namespace os2s;
  OS2S_REQUIRE(x_row_stride >= Cin && x_row_stride % 8 == 0);
  OS2S_REQUIRE(x && dy && dw);
  OS2S_REQUIRE(B >= 0 && Tin >= 1 && Tout >= 1 && Cin >= 8 && Cout >= 8 && K >= 1);
  OS2S_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && stride >= 1 && dil >= 1);
  if (B == 0) return OS2S_OK;
  // taps per workgroup: the dY tile of a step is reused by every tap, so L2->LDS bytes per
  // FLOP fall as 1/TAPS (2 taps: 173 FLOP/B, 3 taps: 257 FLOP/B — the kernels run against
  // the ~7.3 TB/s L2->CU delivery limit, not against MFMA); 3 x 64 accumulator registers fit
  // the 256-register budget of 2 waves/SIMD.
  // (measured: the 3-tap instantiation needs 256 VGPRs + 84 B/lane of scratch and is 20-25 %
  // SLOWER than 2 taps on every Jasper layer — kept selectable for future register work.)
  int TAPS = K == 1 ? 1 : 2;      // 1x1 convs: one accumulator set, no idle tap slot
  if (const char* f = getenv("OS2S_WGRAD_TAPS")) { const int v = atoi(f); if ((v == 2 || v == 3) && K > 1) TAPS = v; }
  // the 256-wide tile pays off once there are enough 256-channel tiles to fill the chip
  const bool wide = (Cout % 256 == 0) && (K >= 8) && (Cout >= 512);
  const int COT = wide ? 256 : 128;
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.in_len = in_len;
  a.B = B; a.Tin = Tin; a.Tout = Tout; a.Cin = Cin; a.Cout = Cout; a.K = K;
  a.stride = stride; a.dil = dil; a.padL = padL; a.x_ld = x_row_stride;
  a.NCO = ceil_div(Cout, COT);
  a.NCI = ceil_div(Cin, 128);
  a.NTP = ceil_div(K, TAPS);
  const int base_blocks = a.NCO * a.NCI * a.NTP;
  const int total_steps = B * ceil_div(Tout, 64);
  int nsplit = 1;
  if (accumulate) {
    // ~1 round of workgroups: the kernel shares the GPU with the data-gradient chain (side
    // stream), where fewer, longer workgroups and fewer atomic passes over dW win slightly
    // (Jasper step 56.1 -> 55.7 ms; 2 rounds were better when the kernel ran alone)
    int target = wide ? 256 : 512;
    if (const char* f = getenv("OS2S_WGRAD_TARGET")) { const int v = atoi(f); if (v > 0) target = v; }
    nsplit = ceil_div(target, base_blocks);
    const int max_split = total_steps / 8 > 0 ? total_steps / 8 : 1;   // >= 8 steps per block
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
  }
  if (const char* f = getenv("OS2S_WGRAD_NSPLIT")) {      // tuning hook (tools/bench_wgrad_shapes.py)
    const int v = atoi(f);
    if (v >= 1 && accumulate) nsplit = v > total_steps ? total_steps : v;
  }
  a.steps_per_split = ceil_div(total_steps, nsplit);
  a.NSPLIT = ceil_div(total_steps, a.steps_per_split);
  a.use_atomic = accumulate ? 1 : 0;
  a.xrows = 63 * stride + (TAPS - 1) * dil + 1;
  a.xrows_pad = ceil_div(a.xrows, 4) * 4;
  const size_t smem = (size_t)2 * 64 * 2 * COT + (size_t)2 * a.xrows_pad * 256;
  if (smem > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<2, 128>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<2, 256>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<1, 128>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<1, 256>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<3, 128>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv1d_wgrad_kernel<3, 256>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OS2S_ERR_LAUNCH;
    attr_set = true;
  }
  const int nunits = a.NCO * a.NCI * a.NSPLIT;
  const int grid = ceil_div(nunits, 8) * 8 * a.NTP;
  if (wide) {
    if (TAPS == 3)
      OS2S_LAUNCH((conv1d_wgrad_kernel<3, 256>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    else if (TAPS == 1)
      OS2S_LAUNCH((conv1d_wgrad_kernel<1, 256>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
    else
      OS2S_LAUNCH((conv1d_wgrad_kernel<2, 256>), dim3(grid), dim3(512), smem, (hipStream_t)stream, a);
  } else {
    if (TAPS == 3)
      OS2S_LAUNCH((conv1d_wgrad_kernel<3, 128>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
    else if (TAPS == 1)
      OS2S_LAUNCH((conv1d_wgrad_kernel<1, 128>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
    else
      OS2S_LAUNCH((conv1d_wgrad_kernel<2, 128>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  }
  return OS2S_OK;
}
