// Depthwise (channel-wise) 1-D convolution: the first half of tf.layers.separable_conv1d
// (layer type "sep_conv1d" of conv_bn_actv / conv_bn_res_bn_actv,
// open_seq2seq/parts/cnns/conv_blocks.py:11-16, the QuartzNet configs); the pointwise half is
// the K = 1 case of the implicit-GEMM kernel.
//   y[b,t,c] = sum_k x[b, t*stride + k*dil - padL, c] * w[k,c]        (x rows >= in_len[b] are zero)
// HBM-bound (one read + one write of the activation, 2*K FLOP per element): a workgroup
// stages the (BT-1)*stride + (K-1)*dil + 1 input rows of a 128-step x 64-channel tile in LDS
// once and every tap re-reads them from there; a thread owns 8 channels (16-byte vectors).
// The data gradient is the same kernel with the taps flipped; the weight gradient
// dw[k,c] = sum_{b,t} dy[b,t,c] x[b, t*stride + k*dil - padL, c] gives each thread one
// (tap, 8-channel) cell over the staged tile and ends in fp32 atomics (K x C is tiny).
#include "os2s_common.hpp"

namespace os2s {

constexpr int kDwBT = 128;   // output time steps per workgroup
constexpr int kDwBC = 64;    // channels per workgroup
constexpr int kDwP = 72;     // LDS row pitch in floats (8 consecutive rows hit distinct banks)

struct DwArgs {
  const bf16_t* x; const float* w; bf16_t* y; const bf16_t* dy; float* dw;
  const int32_t* in_len; const int32_t* out_len;
  int B, Tin, Tout, C, K, stride, dil, padL, flip, R;
  int tile0;      // generic weight gradient: first (sample, time tile) of this launch (deterministic mode: one per launch)
  // matrix-core forward kernel as the LAST data gradient of a conv + BatchNorm + ReLU (+ dropout) layer's output
  // (os2s_depthwise_dgrad_bnact): y = mask(conv + addend), mask = (mask_ref > 0) * mask_scale, and
  // stats[blockIdx.x][0 | 1][c] = sum over the tile's rows of y, y * stat_ref (all in y's layout; addend may be y)
  const bf16_t* mask_ref; const bf16_t* stat_ref; const bf16_t* addend; float mask_scale; float* stats;
};

__device__ __forceinline__ void dw_stage_x(const DwArgs& p, int b, int t0, int c0, float* xs, int len_b) {
  // xs[r][kDwBC] fp32 <- x[b, t0*stride - padL + r, c0 .. c0+63]
  const bf16_t* xb = p.x + (long long)b * p.Tin * p.C;
  const int tin0 = t0 * p.stride - p.padL;
  for (int q = threadIdx.x; q < p.R * (kDwBC / 8); q += blockDim.x) {
    const int r = q / (kDwBC / 8), cg = q - r * (kDwBC / 8);
    const int tin = tin0 + r, ch = c0 + cg * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (tin >= 0 && tin < len_b && ch < p.C) v = *reinterpret_cast<const u32x4*>(xb + (long long)tin * p.C + ch);
    float* d = xs + r * kDwP + cg * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[2 * e] = bflo(v[e]); d[2 * e + 1] = bfhi(v[e]); }
  }
}

__global__ __launch_bounds__(256) void depthwise_fwd_kernel(DwArgs p) {
  extern __shared__ float sm[];
  float* xs = sm;                      // [R][64]
  float* ws = sm + p.R * kDwP;         // [K][64]
  const int ntt = (p.Tout + kDwBT - 1) / kDwBT;
  const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * kDwBT, c0 = blockIdx.y * kDwBC;
  int len_b = p.Tin;
  if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
  if (p.out_len && t0 >= p.out_len[b]) return;        // never-read output tile
  dw_stage_x(p, b, t0, c0, xs, len_b);
  for (int q = threadIdx.x; q < p.K * kDwBC; q += 256) {
    const int k = q / kDwBC, c = q - k * kDwBC;
    ws[q] = (c0 + c < p.C) ? p.w[(long long)(p.flip ? p.K - 1 - k : k) * p.C + c0 + c] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 7, tl = threadIdx.x >> 3;   // 8 channel groups x 32 time lanes
  if (c0 + cg * 8 >= p.C) return;
  if (p.stride == 1 && p.dil == 1) {
    // register-blocked: a thread produces 4 consecutive outputs of its 8 channels; per tap it
    // loads ONE new input row and ONE weight row for 32 FMAs (a sliding 4-row window), i.e.
    // 4x fewer LDS bytes per FLOP than the generic loop below
    const int tt0 = tl * 4;
    if (t0 + tt0 >= p.Tout) return;
    float a[4][8], xw[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[j][e] = 0.f;
    const float* xr = xs + tt0 * kDwP + cg * 8;
    auto ldrow = [&](int r, float (&dst)[8]) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(xr + r * kDwP);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(xr + r * kDwP + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[e] = v0[e]; dst[4 + e] = v1[e]; }
    };
    ldrow(0, xw[0]); ldrow(1, xw[1]); ldrow(2, xw[2]);
    // window slot of output j at step kk (k = kb + kk) is (j + kk) & 3 — static after unrolling
    for (int kb = 0; kb < p.K; kb += 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = kb + kk;
        if (k < p.K) {
          ldrow(k + 3, xw[(3 + kk) & 3]);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + k * kDwBC + cg * 8);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws + k * kDwBC + cg * 8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a[j][e] += xw[(j + kk) & 3][e] * w0[e];
              a[j][4 + e] += xw[(j + kk) & 3][4 + e] * w1[e];
            }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + tt0 + j;
      if (t >= p.Tout) break;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(a[j][2 * e], a[j][2 * e + 1]);
      *reinterpret_cast<u32x4*>(p.y + ((long long)b * p.Tout + t) * p.C + c0 + cg * 8) = o;
    }
    return;
  }
  for (int i = 0; i < kDwBT / 32; ++i) {
    const int tt = tl + 32 * i, t = t0 + tt;
    if (t >= p.Tout) break;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* xr = xs + (tt * p.stride) * kDwP + cg * 8;
    for (int k = 0; k < p.K; ++k) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr + k * p.dil * kDwP);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(xr + k * p.dil * kDwP + 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + k * kDwBC + cg * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws + k * kDwBC + cg * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] += x0[e] * w0[e]; a[4 + e] += x1[e] * w1[e]; }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(a[2 * e], a[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p.y + ((long long)b * p.Tout + t) * p.C + c0 + cg * 8) = o;
  }
}

__global__ __launch_bounds__(256) void depthwise_wgrad_kernel(DwArgs p) {
  extern __shared__ float sm[];
  float* xs = sm;                       // [R][64]
  float* ds = sm + p.R * kDwP;          // [128][pitch] dy tile
  const int ntt = (p.Tout + kDwBT - 1) / kDwBT;
  const int tile = blockIdx.x + p.tile0;
  const int b = tile / ntt, t0 = (tile - b * ntt) * kDwBT, c0 = blockIdx.y * kDwBC;
  int len_b = p.Tin;
  if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
  if (t0 * p.stride - p.padL >= len_b) return;          // the whole X window is padding: zero
  dw_stage_x(p, b, t0, c0, xs, len_b);
  const bf16_t* dyb = p.dy + (long long)b * p.Tout * p.C;
  for (int q = threadIdx.x; q < kDwBT * (kDwBC / 8); q += 256) {
    const int r = q / (kDwBC / 8), cg = q - r * (kDwBC / 8);
    const int t = t0 + r, ch = c0 + cg * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t < p.Tout && ch < p.C) v = *reinterpret_cast<const u32x4*>(dyb + (long long)t * p.C + ch);
    float* d = ds + r * kDwP + cg * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[2 * e] = bflo(v[e]); d[2 * e + 1] = bfhi(v[e]); }
  }
  __syncthreads();
  if (p.stride == 1 && p.dil == 1) {
    // register-blocked: a thread owns 8 consecutive taps x 8 channels over one third of the
    // tile's time steps; per step it loads one dy row and ONE new x row (sliding 8-row
    // window) for 64 FMAs
    const int ngrp = (p.K + 7) / 8;               // tap groups
    const int ncell = ngrp * (kDwBC / 8);
    const int nseg = max(1, min(256 / max(ncell, 1), 4));
    const int cell = threadIdx.x % max(ncell, 1), seg = threadIdx.x / max(ncell, 1);
    if (ncell <= 256) {
      const int kg = cell / (kDwBC / 8), cg = cell - kg * (kDwBC / 8);
      const int k0 = kg * 8;
      const bool active = seg < nseg && c0 + cg * 8 < p.C;
      float a[8][8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[j][e] = 0.f;
      if (active) {
        const int ta = seg * kDwBT / nseg, tb = (seg + 1) * kDwBT / nseg;
        float xw[8][8];
        auto xrow = [&](int r, float (&dst)[8]) {
          if (r < p.R) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(xs + r * kDwP + cg * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(xs + r * kDwP + cg * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dst[e] = v0[e]; dst[4 + e] = v1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = 0.f;
          }
        };
#pragma unroll
        for (int j = 0; j < 7; ++j) xrow(ta + k0 + j, xw[j]);
        // window slot of tap j at step ti (tt = tb0 + ti) is (j + ti) & 7 — static after unrolling
        for (int tb0 = ta; tb0 < tb; tb0 += 8) {
#pragma unroll
          for (int ti = 0; ti < 8; ++ti) {
            const int tt = tb0 + ti;
            if (tt < tb) {
              xrow(tt + k0 + 7, xw[(7 + ti) & 7]);
              const f32x4 d0 = *reinterpret_cast<const f32x4*>(ds + tt * kDwP + cg * 8);
              const f32x4 d1 = *reinterpret_cast<const f32x4*>(ds + tt * kDwP + cg * 8 + 4);
#pragma unroll
              for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  a[j][e] += xw[(j + ti) & 7][e] * d0[e];
                  a[j][4 + e] += xw[(j + ti) & 7][4 + e] * d1[e];
                }
            }
          }
        }
      }
      // combine the time segments through LDS: ONE atomic per (tap, channel) and workgroup
      __syncthreads();                          // everyone is done reading xs / ds
      float* red = sm;                          // [nseg][ncell][64]
      if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) red[((seg * ncell + cell) * 8 + j) * 8 + e] = a[j][e];
      }
      __syncthreads();
      for (int q = threadIdx.x; q < ncell * 64; q += 256) {
        const int cl = q >> 6, je = q & 63, j = je >> 3, e = je & 7;
        const int kg2 = cl / (kDwBC / 8), cg2 = cl - kg2 * (kDwBC / 8);
        const int k = kg2 * 8 + j, ch = c0 + cg2 * 8 + e;
        if (k >= p.K || ch >= p.C) continue;
        float v = 0.f;
        for (int sg = 0; sg < nseg; ++sg) v += red[(sg * ncell + cl) * 64 + je];
        if (v != 0.f) __hip_atomic_fetch_add(p.dw + (long long)k * p.C + ch, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }
  for (int cell = threadIdx.x; cell < p.K * (kDwBC / 8); cell += 256) {
    const int k = cell / (kDwBC / 8), cg = cell - k * (kDwBC / 8);
    if (c0 + cg * 8 >= p.C) continue;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tt = 0; tt < kDwBT; ++tt) {
      const float* xr = xs + (tt * p.stride + k * p.dil) * kDwP + cg * 8;
      const float* dr = ds + tt * kDwP + cg * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += xr[e] * dr[e];
    }
    float* o = p.dw + (long long)k * p.C + c0 + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (a[e] != 0.f) __hip_atomic_fetch_add(o + e, a[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// Stride-1, dilation-1 kernels (every QuartzNet block: K = 33 .. 75). With 2*K FLOP per element
// the op is VALU-bound, not HBM-bound (K = 75, C = 512: 2 GFLOP per launch against 55 MB), and the
// first kernels above were bound by LDS bytes per FMA and by one wave per SIMD (fp32 tiles of
// 58-95 KB). Here:
//   * tiles live in LDS as bf16 (136-byte rows: the 16-row offset between the time groups of a
//     wave lands on the other half of the banks) — 64-78 KB per workgroup, two workgroups per CU;
//   * a thread owns 4 channels x 16 outputs (forward) or 4 channels x 16 taps (weight gradient):
//     64 accumulators and a sliding 16-row register window of the input, so one new 8-byte row
//     (+ one weight row / one dy row) feeds 64 FMAs — 0.25-0.4 LDS bytes per FMA instead of 1-2;
//   * the FMAs are v_pk_fma_f32 over channel pairs (the fp32 vector peak is the packed rate).
// ---------------------------------------------------------------------------------------------
constexpr int kD16BT = 256;      // time steps per tile
constexpr int kD16Pitch = 136;   // LDS bytes per 64-channel bf16 row of the x tile

// rows [0, nrows) of the x tile: LDS row r <-> input time t0 - padL + r, channels c0 .. c0+63
__device__ __forceinline__ void d16_stage_x(const DwArgs& p, int b, int t0, int c0, char* xs, int nrows, int len_b) {
  const bf16_t* xb = p.x + (long long)b * p.Tin * p.C;
  const int tin0 = t0 - p.padL;
  for (int q = threadIdx.x; q < nrows * 8; q += 256) {
    const int r = q >> 3, cg = q & 7;
    const int tin = tin0 + r, ch = c0 + cg * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (tin >= 0 && tin < len_b && ch < p.C) v = *reinterpret_cast<const u32x4*>(xb + (long long)tin * p.C + ch);
    u32x2* d = reinterpret_cast<u32x2*>(xs + r * kD16Pitch + cg * 16);
    d[0] = u32x2{v[0], v[1]};
    d[1] = u32x2{v[2], v[3]};
  }
}

__device__ __forceinline__ void d16_row(const char* ptr, f32x2 (&dst)[2]) {
  const u32x2 v = *reinterpret_cast<const u32x2*>(ptr);
  dst[0] = f32x2{bflo(v[0]), bfhi(v[0])};
  dst[1] = f32x2{bflo(v[1]), bfhi(v[1])};
}

// Dilation D (stride 1): outputs whose (t - t0) has the same residue mod D only touch LDS rows of
// that residue — y[t0 + q + D u] = sum_k x_row[q + D (u + k)] w[k] is a dilation-1 convolution on
// the sub-sequence of residue q. A thread therefore owns 16 outputs of ONE residue class (time
// group tg -> class tg % D, 16-output block tg / D) and walks the rows with a step of D.
template <int D>
__global__ __launch_bounds__(256, 2) void depthwise_fwd16_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int R = kD16BT + (p.K - 1) * D;
  char* const xs = smc;                                              // [R][136 B] bf16
  float* const ws = reinterpret_cast<float*>(smc + ((R * kD16Pitch + 15) & ~15));   // [K][64] fp32
  const int ntt = (p.Tout + kD16BT - 1) / kD16BT;
  const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * kD16BT, c0 = blockIdx.y * kDwBC;
  int len_b = p.Tin;
  if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
  if (p.out_len && t0 >= p.out_len[b]) return;        // never-read output tile
  if (t0 - p.padL >= len_b) {
    // the whole input window lies past the sequence end: the outputs are exact zeros
    const int rows = min(kD16BT, p.Tout - t0), c8n = min(kDwBC, p.C - c0) >> 3;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int q = threadIdx.x; q < rows * c8n; q += 256) {
      const int r = q / c8n, cg8 = q - r * c8n;
      *reinterpret_cast<u32x4*>(p.y + ((long long)b * p.Tout + t0 + r) * p.C + c0 + cg8 * 8) = z;
    }
    return;
  }
  d16_stage_x(p, b, t0, c0, xs, R, len_b);
  for (int q = threadIdx.x; q < p.K * kDwBC; q += 256) {
    const int k = q / kDwBC, c = q - k * kDwBC;
    ws[q] = (c0 + c < p.C) ? p.w[(long long)(p.flip ? p.K - 1 - k : k) * p.C + c0 + c] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 15, tg = threadIdx.x >> 4;            // 16 channel quads x 16 time groups
  const int tt0 = (tg % D) + D * 16 * (tg / D);                      // first output of the thread (tile-relative)
  if (c0 + cg * 4 >= p.C || t0 + tt0 >= p.Tout) return;
  f32x2 a[16][2], xw[16][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) { a[j][0] = f32x2{0.f, 0.f}; a[j][1] = f32x2{0.f, 0.f}; }
  constexpr int rstep = D * kD16Pitch;                               // LDS bytes between rows of the class
  const char* const xr = xs + tt0 * kD16Pitch + cg * 8;
#pragma unroll
  for (int sl = 0; sl < 15; ++sl) d16_row(xr + sl * rstep, xw[sl]);
  // window slot of output j at tap k = kb + kk is (j + kk) & 15 — static after unrolling
  for (int kb = 0; kb < p.K; kb += 16) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kb + kk;
      if (k < p.K) {
        d16_row(xr + (k + 15) * rstep, xw[(15 + kk) & 15]);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + k * kDwBC + cg * 4);
        const f32x2 w0 = {wv[0], wv[1]}, w1 = {wv[2], wv[3]};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          a[j][0] = __builtin_elementwise_fma(xw[(j + kk) & 15][0], w0, a[j][0]);
          a[j][1] = __builtin_elementwise_fma(xw[(j + kk) & 15][1], w1, a[j][1]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int t = t0 + tt0 + D * j;
    if (t >= p.Tout) break;
    u32x2 o;
    o[0] = pack2bf(a[j][0][0], a[j][0][1]);
    o[1] = pack2bf(a[j][1][0], a[j][1][1]);
    *reinterpret_cast<u32x2*>(p.y + ((long long)b * p.Tout + t) * p.C + c0 + cg * 4) = o;
  }
}

// dw[k,c] += sum_{b,t} dy[b,t,c] x[b, t + k - padL, c]. A workgroup walks (sample, 256-step) tiles of
// its 64 channels with a grid stride and keeps the sums in registers: thread = (16-tap group, channel
// quad, time segment); one LDS reduction over the segments and ONE atomic per (tap, channel) and
// workgroup at the end (the first kernel issued them per 128-step tile).
// Dilation D: a time segment is (residue class of t mod D, time range); within it consecutive
// steps are D apart and the x rows of tap j are D (k0 + j) further on — the dilation-1 walk with a
// row step of D. nseg is a multiple of D.
template <int D>
__global__ __launch_bounds__(256, 2) void depthwise_wgrad16_kernel(DwArgs p, int ngrp, int nseg) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int XR = kD16BT + 16 * ngrp * D;                             // x rows staged per tile
  char* const xs = smc;                                              // [XR][136 B] bf16
  char* const ds = smc + XR * kD16Pitch;                             // [256][128 B] bf16 dy tile
  const int ntt = (p.Tout + kD16BT - 1) / kD16BT;
  const int c0 = blockIdx.y * kDwBC;
  const int ncell = ngrp * 16;
  const int cell = threadIdx.x % ncell, seg = threadIdx.x / ncell;
  const int kg = cell >> 4, cg = cell & 15, k0 = kg * 16;
  const bool active = seg < nseg && c0 + cg * 4 < p.C;
  const int nts = nseg / D, par = seg % D, tsg = seg / D;            // time ranges, residue class, range
  const int ta = tsg * kD16BT / nts + par, tb = (tsg + 1) * kD16BT / nts;
  const int nst = tb > ta ? (tb - ta + D - 1) / D : 0;               // steps of this thread per tile
  f32x2 a[16][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) { a[j][0] = f32x2{0.f, 0.f}; a[j][1] = f32x2{0.f, 0.f}; }
  for (int tile = blockIdx.x; tile < p.B * ntt; tile += gridDim.x) {
    const int b = tile / ntt, t0 = (tile - b * ntt) * kD16BT;
    int len_b = p.Tin;
    if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
    if (t0 - p.padL >= len_b) continue;                              // the whole X window is padding
    d16_stage_x(p, b, t0, c0, xs, XR, len_b);
    const bf16_t* dyb = p.dy + (long long)b * p.Tout * p.C;
    for (int q = threadIdx.x; q < kD16BT * 8; q += 256) {
      const int r = q >> 3, cg8 = q & 7;
      const int t = t0 + r, ch = c0 + cg8 * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (t < p.Tout && ch < p.C) v = *reinterpret_cast<const u32x4*>(dyb + (long long)t * p.C + ch);
      *reinterpret_cast<u32x4*>(ds + r * 128 + cg8 * 16) = v;
    }
    __syncthreads();
    if (active) {
      f32x2 xw[16][2];
      constexpr int rstep = D * kD16Pitch;
      const char* const xr = xs + (ta + k0 * D) * kD16Pitch + cg * 8;
      const char* const dr = ds + ta * 128 + cg * 8;
#pragma unroll
      for (int sl = 0; sl < 15; ++sl) d16_row(xr + sl * rstep, xw[sl]);
      // window slot of tap j at step ti is (j + ti) & 15 — static after unrolling
      for (int i0 = 0; i0 < nst; i0 += 16) {
#pragma unroll
        for (int ti = 0; ti < 16; ++ti) {
          const int i = i0 + ti;
          if (i < nst) {
            d16_row(xr + (i + 15) * rstep, xw[(15 + ti) & 15]);
            f32x2 dv[2];
            d16_row(dr + i * D * 128, dv);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              a[j][0] = __builtin_elementwise_fma(xw[(j + ti) & 15][0], dv[0], a[j][0]);
              a[j][1] = __builtin_elementwise_fma(xw[(j + ti) & 15][1], dv[1], a[j][1]);
            }
          }
        }
      }
    }
    __syncthreads();                                                 // the tiles are overwritten next
  }
  // combine the time segments through LDS: ONE atomic per (tap, channel) and workgroup
  float* const red = reinterpret_cast<float*>(smc);                  // [nseg][ncell][16 taps][4 ch]
  if (active) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<f32x4*>(red + ((seg * ncell + cell) * 16 + j) * 4) = f32x4{a[j][0][0], a[j][0][1], a[j][1][0], a[j][1][1]};
  }
  __syncthreads();
  for (int q = threadIdx.x; q < ncell * 64; q += 256) {
    const int cl = q >> 6, je = q & 63, j = je >> 2, e = je & 3;
    const int k = (cl >> 4) * 16 + j, ch = c0 + (cl & 15) * 4 + e;
    if (k >= p.K || ch >= p.C) continue;
    float v = 0.f;
    for (int sg = 0; sg < nseg; ++sg) v += red[(sg * ncell + cl) * 64 + je];
    if (v != 0.f) __hip_atomic_fetch_add(p.dw + (long long)k * p.C + ch, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


// ---------------------------------------------------------------------------------------------
// Stride-1, dilation-1 forward / data gradient on the matrix cores (round 6).
//
// The register-window kernel above runs at ~37 TFLOP/s of packed fp32 FMAs — 0.13 of what HBM would allow
// (2 K FLOP per 4 bytes). A depthwise convolution is time-invariant: with the time axis of ONE (sample,
// channel) cut into 32 segments of 32 steps,
//     y[32 s + i] = sum_d w[d] x[32 s + i + d - padL]   =   sum_p T[i][p] X[p][s],
//     T[i][p] = w[p - i]  (a 32 x (32 + K - 1) Toeplitz band),   X[p][s] = x[32 s + p - padL],
// i.e. a [32 x 16 J] . [16 J x 32] product per channel whose B operand is the channel's time series read at 32
// overlapping offsets — v_mfma_f32_32x32x16_bf16 with M = position in the segment, N = segment, K = window
// position. A workgroup owns 32 channels (64 B of every row) x 1024 time steps of one sample: the rows are
// transposed on the way into LDS (a lane loads 4 rows x 8 channels and writes one 8-byte run of 4 time steps
// per channel), every channel is a PLANE of consecutive time steps, a B fragment is one aligned ds_read_b128
// at chunk 4 s + 2 j + lhi, an A fragment one UNALIGNED ds_read_b128 of the zero-padded tap table at
// 32 + 16 j + 8 lhi - i (gfx950 reads LDS at any 2-byte address). The taps enter as a bf16 hi + lo pair
// (two MFMAs per step: the result is the fp32-weight convolution the VALU kernel computes, to fp32 rounding).
// Outputs go back into the wave's own plane (time-contiguous) and leave through the inverse transposition.
// 8 waves x 4 channels; <= 80 KB of LDS: two workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int kDmCh = 32;        // channels per workgroup

struct DmGeom {
  int Jt;            // 16-wide window steps per segment: ceil((32 + K - 1) / 16)
  int nseg;          // 32-step segments per workgroup tile (<= 32): as many as fit 128 logical chunks
  int nchunk;        // logical 16-byte chunks (8 time steps) of a plane
  int plane_bytes;   // physical: one pad chunk after every 16; plane_bytes % 256 == 128
  int wt;            // elements of one tap table (index = d + 32)
};
__host__ __device__ inline DmGeom dm_geom(int K) {
  DmGeom g;
  g.Jt = (32 + K - 1 + 15) / 16;
  g.nseg = (1056 - 16 * g.Jt) / 32;                  // window of the last segment ends at 32 (nseg - 1) + 16 Jt <= 1024
  if (g.nseg > 32) g.nseg = 32;
  g.nchunk = (32 * (g.nseg - 1) + 16 * g.Jt) / 8;    // <= 128
  int phys = (g.nchunk - 1) + ((g.nchunk - 1) >> 4) + 1;
  while ((phys * 16) % 256 != 128) ++phys;           // 136 for every K <= 96: 69.6 KB of planes, two workgroups per CU
  g.plane_bytes = phys * 16;
  g.wt = 16 * g.Jt + 40;
  return g;
}
__device__ __forceinline__ int dm_chunk_off(int q) { return (q + (q >> 4)) * 16; }   // byte offset of logical chunk q
__device__ __forceinline__ int dm_plane_of(int ch) { return (ch & 7) * 4 + (ch >> 3); }

__global__ __launch_bounds__(512, 2) void depthwise_mfma_fwd_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const DmGeom g = dm_geom(p.K);
  char* const planes = smc;                                            // [32 planes][plane_bytes]
  bf16_t* const wtab_all = reinterpret_cast<bf16_t*>(smc + kDmCh * g.plane_bytes);   // [8 waves][2][wt]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Tt = 32 * g.nseg;                                // time steps per workgroup
  const int ntt = (p.Tout + Tt - 1) / Tt;
  const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * Tt, c0 = blockIdx.y * kDmCh;
  int len_b = p.Tin;
  if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
  const int rows_out = min(Tt, p.Tout - t0);
  if (p.out_len && t0 >= p.out_len[b]) {                   // never-read output tile
    if (p.mask_ref) {
      // fused data gradient: the producer's BatchNorm backward may walk every row (separable producers): zeros
      const int ncg0 = min(4, (p.C - c0) >> 3);
      const u32x4 z = {0u, 0u, 0u, 0u};
      for (int q = tid; q < rows_out * ncg0; q += 512) {
        const int r = q / ncg0, cg = q - r * ncg0;
        *reinterpret_cast<u32x4*>(p.y + ((long long)b * p.Tout + t0 + r) * p.C + c0 + cg * 8) = z;
      }
    }
    return;
  }
  const int ncg = min(4, (p.C - c0) >> 3);                  // 8-channel groups of this block
  if (t0 - p.padL >= len_b && !p.mask_ref) {
    // the whole input window lies past the sequence end: exact zeros
    // (fused data gradient: the tile still carries the addend and the statistics — the general path, whose row
    // loads are all masked)
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int q = tid; q < rows_out * ncg; q += 512) {
      const int r = q / ncg, cg = q - r * ncg;
      *reinterpret_cast<u32x4*>(p.y + ((long long)b * p.Tout + t0 + r) * p.C + c0 + cg * 8) = z;
    }
    return;
  }
  // the taps of the wave's four channels: issued in front of the row loads (one memory round trip for both)
  float wreg[4][2];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = lane + 64 * r, ch = j4 * 8 + wave;
      wreg[j4][r] = (k < p.K && c0 + ch < p.C) ? p.w[(long long)(p.flip ? p.K - 1 - k : k) * p.C + c0 + ch] : 0.f;
    }
  // ---- stage: rows [t0 - padL, t0 - padL + 8 nchunk) -> planes (time-contiguous per channel) -------------
  const bf16_t* const xb = p.x + (long long)b * p.Tin * p.C;
  const int n4 = g.nchunk * 2;                                // groups of 4 positions
  // positions past the last live row of the tile hold zeros. All row loads of a thread are issued before the
  // first is consumed (up to 3 items x 4 rows: one memory round trip, not one per loop trip)
  constexpr int kIt = 2;                                       // 8 nchunk <= 1024 items / 512 threads
  u32x4 r[kIt][4];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int q = tid + 512 * it;
    const int cg = q & 3, p4 = q >> 2;
    const int tin0 = t0 - p.padL + p4 * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tin = tin0 + u;
      r[it][u] = u32x4{0u, 0u, 0u, 0u};
      if (q < n4 * 4 && cg < ncg && tin >= 0 && tin < len_b && !(p.tile0 & 4))
        r[it][u] = *reinterpret_cast<const u32x4*>(xb + (long long)tin * p.C + c0 + cg * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int q = tid + 512 * it;
    if (q >= n4 * 4) break;
    const int cg = q & 3, p4 = q >> 2;
    char* const dst = planes + dm_chunk_off(p4 >> 1) + (p4 & 1) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // channel 2k (low halves) and 2k + 1 (high halves) of the 8-channel group
      u32x2 lo, hi;
      lo[0] = (r[it][0][k] & 0xffffu) | (r[it][1][k] << 16);
      lo[1] = (r[it][2][k] & 0xffffu) | (r[it][3][k] << 16);
      hi[0] = (r[it][0][k] >> 16) | (r[it][1][k] & 0xffff0000u);
      hi[1] = (r[it][2][k] >> 16) | (r[it][3][k] & 0xffff0000u);
      *reinterpret_cast<u32x2*>(dst + dm_plane_of(cg * 8 + 2 * k) * g.plane_bytes) = lo;
      *reinterpret_cast<u32x2*>(dst + dm_plane_of(cg * 8 + 2 * k + 1) * g.plane_bytes) = hi;
    }
  }
  // the wave's tap table: zeros outside [32, 32 + K)
  bf16_t* const wt_hi = wtab_all + wave * 2 * g.wt;
  bf16_t* const wt_lo = wt_hi + g.wt;
  for (int i = lane; i < 2 * g.wt; i += 64) wt_hi[i] = 0;
  __syncthreads();
  // ---- per wave: channels wave, wave + 8, wave + 16, wave + 24 (planes 4 wave .. 4 wave + 3) -------------
  // (no workgroup barrier inside the loop: table and plane of a channel belong to ONE wave, whose LDS
  // instructions execute in issue order)
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) {
    const int ch = j4 * 8 + wave;
    const bool live = c0 + ch < p.C && !(p.tile0 & 1);
    if (live) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        if (k < p.K) {
          const bf16_t h = f2bf(wreg[j4][r]);
          wt_hi[32 + k] = h;
          wt_lo[32 + k] = f2bf(wreg[j4][r] - bf2f(h));
        }
      }
    }
    if (live) {
      char* const pl = planes + dm_plane_of(ch) * g.plane_bytes;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      // A fragment = 8 taps at element 32 + 16 jj + 8 lhi - l31 of the table: a 2-byte-aligned 16-byte window.
      // (One unaligned ds_read_b128 does it, but a lane-private misalignment costs ~100 cycles per instruction:
      // the first version of this kernel was bound by those 32 reads per channel.) Five dword-aligned reads
      // and a funnel shift by 0 or 16 bits instead.
      const int seg = l31 < g.nseg ? l31 : 0;       // columns past the last segment recompute segment 0 (never stored)
      const int e0 = 32 + 8 * lhi - l31;
      const uint32_t* const ahw = reinterpret_cast<const uint32_t*>(wt_hi) + (e0 >> 1);
      const uint32_t* const alw = reinterpret_cast<const uint32_t*>(wt_lo) + (e0 >> 1);
      const uint32_t sh = (e0 & 1) ? 16u : 0u;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj < g.Jt) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(pl + dm_chunk_off(4 * seg + 2 * jj + lhi));
          uint32_t dh[5], dl[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) { dh[k] = ahw[8 * jj + k]; dl[k] = alw[8 * jj + k]; }
          u32x4 fh, fl;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            fh[k] = __builtin_amdgcn_alignbit(dh[k + 1], dh[k], sh);
            fl[k] = __builtin_amdgcn_alignbit(dl[k + 1], dl[k], sh);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh), bfr, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fl), bfr, acc, 0, 0, 0);
        }
      }
      // outputs back into the plane: acc[4 q + e] = step 32 s + 8 q + 4 lhi + e of segment s = l31
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x2 o;
        o[0] = pack2bf(acc[4 * q], acc[4 * q + 1]);
        o[1] = pack2bf(acc[4 * q + 2], acc[4 * q + 3]);
        if (l31 < g.nseg) *reinterpret_cast<u32x2*>(pl + dm_chunk_off(4 * l31 + q) + lhi * 8) = o;
      }
    }
  }
  __syncthreads();
  // ---- planes -> rows: 4 time steps x 8 channels per item ------------------------------------------------
  bf16_t* const yb = p.y + ((long long)b * p.Tout + t0) * p.C + c0;
  const int n4o = (p.tile0 & 2) ? 0 : (rows_out + 3) >> 2;
  const bool fused = p.mask_ref != nullptr;
  const long long tile_off = ((long long)b * p.Tout + t0) * p.C + c0;
  float sz[8], sy[8];                       // fused: sums of the stored values / of stored value x stat_ref
#pragma unroll
  for (int e = 0; e < 8; ++e) { sz[e] = 0.f; sy[e] = 0.f; }
  for (int q = tid; q < n4o * 4; q += 512) {
    const int cg = q & 3, p4 = q >> 2;
    if (cg >= ncg) continue;
    const char* const src = planes + dm_chunk_off(p4 >> 1) + (p4 & 1) * 8;
    u32x2 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const u32x2*>(src + dm_plane_of(cg * 8 + e) * g.plane_bytes);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p4 * 4 + u >= rows_out) break;
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t a = v[2 * k][u >> 1], c = v[2 * k + 1][u >> 1];
        o[k] = (u & 1) ? ((a >> 16) | (c & 0xffff0000u)) : ((a & 0xffffu) | (c << 16));
      }
      const long long off = (long long)(p4 * 4 + u) * p.C + cg * 8;
      if (fused) {
        const u32x4 mk = *reinterpret_cast<const u32x4*>(p.mask_ref + tile_off + off);
        const u32x4 st = *reinterpret_cast<const u32x4*>(p.stat_ref + tile_off + off);
        u32x4 ad = {0u, 0u, 0u, 0u};
        if (p.addend) ad = *reinterpret_cast<const u32x4*>(p.addend + tile_off + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float lo = bflo(o[k]) + bflo(ad[k]), hi = bfhi(o[k]) + bfhi(ad[k]);
          lo = bflo(mk[k]) > 0.f ? lo * p.mask_scale : 0.f;
          hi = bfhi(mk[k]) > 0.f ? hi * p.mask_scale : 0.f;
          o[k] = pack2bf(lo, hi);
          lo = bflo(o[k]); hi = bfhi(o[k]);            // the statistics see the stored (rounded) values
          sz[2 * k] += lo; sz[2 * k + 1] += hi;
          sy[2 * k] += lo * bflo(st[k]); sy[2 * k + 1] += hi * bfhi(st[k]);
        }
      }
      *reinterpret_cast<u32x4*>(yb + off) = o;
    }
  }
  if (!fused) return;
  // per-channel sums of the tile in a fixed order: thread partials through LDS (the planes are dead), then one
  // thread per (statistic, channel) adds the 128 partials of its 8-channel group
  __syncthreads();
  float* const red = reinterpret_cast<float*>(smc);            // [512][16]
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = sz[e]; red[tid * 16 + 8 + e] = sy[e]; }
  __syncthreads();
  if (tid < 64) {
    const int ch = tid & 31, which = tid >> 5;
    if (c0 + ch < p.C) {
      const int cg = ch >> 3, e = ch & 7;
      float a = 0.f;
      for (int t = cg; t < 512; t += 4) a += red[t * 16 + which * 8 + e];
      p.stats[((long long)blockIdx.x * 2 + which) * p.C + c0 + ch] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stride-1, dilation-1 weight gradient on the matrix cores (round 6).
//   dw[k] = sum_t dy[t] x[t + k - padL]   per (sample, channel), summed over the batch.
// With the time axis cut into 16-step segments s, H[i][p] = sum_s dy[16 s + i] x[16 s + p - padL] is a product
// that contracts over the SEGMENT index — both operands are the time-contiguous channel planes of the forward
// kernel read with the transposing LDS read (ds_read_b64_tr_b16: memory rows = segments, a lane receives 4
// consecutive segments of one position) — and dw[k] = sum_i H[i][i + k]: v_mfma_f32_16x16x32_bf16, M = position
// in the segment, N = window position (ceil((K + 15) / 16) tiles of 16), K-dim = 32 segments. H is translation
// invariant, so a wave keeps the accumulators of its two channels (<= 2 x 7 x 4 registers) over every tile it
// walks and sums the diagonals once at the end (through LDS, in a fixed order; one atomic per (tap, channel) and
// workgroup — ONE workgroup per channel block in deterministic mode). A workgroup owns 16 channels (32 B of a row)
// and tiles of 928 time steps of one sample: x plane 1120 positions, dy plane 1024 (zeros past the tile), 73.7 KB,
// two workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int kDwgCh = 16;                 // channels per workgroup
constexpr int kDwgTt = 928;                // time steps per tile (58 segments of 16; the two K-steps read 64)
constexpr int kDwgXPlane = 2432;           // 152 chunks: 1120 positions + a pad chunk per 16; % 256 == 128
constexpr int kDwgYPlane = 2176;           // 136 chunks: 1024 positions
__device__ __forceinline__ int dwg_plane_of(int ch) { return (ch & 7) * 2 + (ch >> 3); }
__device__ __forceinline__ int dwg_pos_off(int pos) {        // byte offset of time position pos inside a plane
  const int q = pos >> 3;
  return (q + (q >> 4)) * 16 + (pos & 7) * 2;
}

// rows -> planes for NCG 8-channel groups: position p <-> row first_row + p, rows outside [lo, hi) are zeros
template <int NIT, int NCG, typename PlaneOf>
__device__ __forceinline__ void dm_stage_planes(const bf16_t* __restrict__ rows, long long row_stride, int first_row,
                                                int lo, int hi, int ngroups4, int ncg_live, char* planes,
                                                int plane_bytes, PlaneOf plane_of) {
  const int tid = threadIdx.x;
  u32x4 r[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = tid + 512 * it;
    const int cg = q % NCG, p4 = q / NCG;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = first_row + p4 * 4 + u;
      r[it][u] = u32x4{0u, 0u, 0u, 0u};
      if (p4 < ngroups4 && cg < ncg_live && row >= lo && row < hi)
        r[it][u] = *reinterpret_cast<const u32x4*>(rows + (long long)row * row_stride + cg * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = tid + 512 * it;
    const int cg = q % NCG, p4 = q / NCG;
    if (p4 >= ngroups4) break;
    char* const dst = planes + dm_chunk_off(p4 >> 1) + (p4 & 1) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x2 lo2, hi2;
      lo2[0] = (r[it][0][k] & 0xffffu) | (r[it][1][k] << 16);
      lo2[1] = (r[it][2][k] & 0xffffu) | (r[it][3][k] << 16);
      hi2[0] = (r[it][0][k] >> 16) | (r[it][1][k] & 0xffff0000u);
      hi2[1] = (r[it][2][k] >> 16) | (r[it][3][k] & 0xffff0000u);
      *reinterpret_cast<u32x2*>(dst + plane_of(cg * 8 + 2 * k) * plane_bytes) = lo2;
      *reinterpret_cast<u32x2*>(dst + plane_of(cg * 8 + 2 * k + 1) * plane_bytes) = hi2;
    }
  }
}

__device__ __forceinline__ bf16x4 dm_lds_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

__global__ __launch_bounds__(512, 2) void depthwise_mfma_wgrad_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  char* const xpl = smc;                                   // [16][kDwgXPlane]
  char* const ypl = smc + kDwgCh * kDwgXPlane;             // [16][kDwgYPlane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t16 = lane & 15, g4 = lane >> 4;
  const int c0 = blockIdx.y * kDwgCh;
  const int ncg = min(2, (p.C - c0) >> 3);
  const int ntt = (p.Tout + kDwgTt - 1) / kDwgTt;
  const int ntiles = p.B * ntt;
  const int NT = (p.K + 15 + 15) / 16;                     // window tiles of 16 positions (<= 7)
  f32x4 acc[2][7];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int n = 0; n < 7; ++n) acc[c][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / ntt, t0 = (tile - b * ntt) * kDwgTt;
    int len_b = p.Tin;
    if (p.in_len) len_b = min(max(p.in_len[b], 0), p.Tin);
    if (t0 - p.padL >= len_b) continue;                    // every x row of the window is masked: no contribution
    __syncthreads();                                       // the previous tile's planes are no longer read
    dm_stage_planes<2, 2>(p.x + (long long)b * p.Tin * p.C + c0, p.C, t0 - p.padL, 0, (p.tile0 & 4) ? 0 : len_b, 280, ncg, xpl,
                          kDwgXPlane, dwg_plane_of);
    dm_stage_planes<1, 2>(p.dy + (long long)b * p.Tout * p.C + c0, p.C, t0, t0, (p.tile0 & 4) ? 0 : min(p.Tout, t0 + kDwgTt), 256, ncg,
                          ypl, kDwgYPlane, dwg_plane_of);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = c * 8 + wave;                         // planes 2 wave, 2 wave + 1
      if (c0 + ch >= p.C || (p.tile0 & 1)) continue;
      const char* const xp = xpl + dwg_plane_of(ch) * kDwgXPlane;
      const char* const yp = ypl + dwg_plane_of(ch) * kDwgYPlane;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // the lane supplies segment row 32 ks + 8 g4 + 4 h + (t16 >> 2), positions 4 (t16 & 3) .. + 3 of the tile
        const int row0 = 32 * ks + 8 * g4 + (t16 >> 2), col0 = 4 * (t16 & 3);
        const bf16x4 a0 = dm_lds_tr(yp + dwg_pos_off(16 * row0 + col0));
        const bf16x4 a1 = dm_lds_tr(yp + dwg_pos_off(16 * (row0 + 4) + col0));
        const bf16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int n = 0; n < 7; ++n) {
          if (n < NT) {
            const bf16x4 b0 = dm_lds_tr(xp + dwg_pos_off(16 * row0 + 16 * n + col0));
            const bf16x4 b1 = dm_lds_tr(xp + dwg_pos_off(16 * (row0 + 4) + 16 * n + col0));
            const bf16x8 bb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[c][n], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- diagonal sums: dw[k] = sum_i H[i][i + k], H[i = 4 g4 + e][p = 16 n + t16] = acc[.][n][e] ------------
  __syncthreads();
  float* const sc = reinterpret_cast<float*>(smc) + wave * (16 * 112);    // [16 i][112 k] fp32 per wave
  float* const dwl = reinterpret_cast<float*>(smc) + 8 * (16 * 112);       // [K][16 channels] sums of the workgroup
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ch = c * 8 + wave;
    if (c0 + ch < p.C && !(p.tile0 & 2)) {
#pragma unroll
      for (int n = 0; n < 7; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e, k = 16 * n + t16 - i;
          if (n < NT && k >= 0 && k < p.K) sc[i * 112 + k] = acc[c][n][e];
        }
      // (same wave wrote and reads: LDS instructions of a wave execute in order)
      for (int k = lane; k < p.K; k += 64) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += sc[i * 112 + k];
        dwl[k * kDwgCh + ch] = sum;
      }
    }
  }
  __syncthreads();
  // one atomic per (tap, channel) and workgroup, lanes along the CHANNELS: 16 consecutive floats per tap (a lane per
  // tap put every atomic of an instruction into a cache line of its own: 35 - 65 us of a 45 - 90 us launch)
  for (int idx = tid; idx < p.K * kDwgCh; idx += 512) {
    const int k = idx / kDwgCh, ch = idx - k * kDwgCh;
    if (c0 + ch < p.C) atomicAdd(p.dw + (long long)k * p.C + c0 + ch, dwl[idx]);
  }
}

// ---- a K = 1 separable layer folded into its pointwise convolution ------------------------------------------
// The residual branches of a separable block are separable layers with ONE tap (conv_blocks.py:66,79-85): a
// per-channel scale d in front of a 1x1 convolution W, y = (x . d) W^T = x (W diag d)^T. The scaled copy of x is
// never made: the forward runs the 1x1 kernel on x with W diag(d) (and its transpose for the data gradient), the
// backward takes G = dy^T x from the 1x1 weight-gradient kernel and splits it: dW += G diag(d),
// dd[ci] += sum_co W[co, ci] G[co, ci].
__global__ __launch_bounds__(256) void pointwise_fold_kernel(const float* __restrict__ w, const float* __restrict__ d,
                                                             bf16_t* __restrict__ w_eff, bf16_t* __restrict__ wt_eff,
                                                             int cout, int cin) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int ci = blockIdx.x * 32 + tx;
  const float dv = ci < cin ? d[ci] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = ty + j * 8, co = blockIdx.y * 32 + r;
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = w[(long long)co * cin + ci] * dv;
      w_eff[(long long)co * cin + ci] = f2bf(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  const int co = blockIdx.y * 32 + tx;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = ty + j * 8, ci2 = blockIdx.x * 32 + r;
    if (co < cout && ci2 < cin) wt_eff[(long long)ci2 * cout + co] = f2bf(tile[tx][r]);
  }
}

// one workgroup per 16 input channels: thread (g, c) = (tid / 16, tid % 16) walks the output channels g, g + 16, ...
// of column c (64-byte row segments), the 16 partial sums of dd meet in LDS in a fixed order (one writer per
// element of dw and dd: deterministic). 16 - 64 workgroups of 16 - 64 steps: ~10 us where one workgroup per 64
// columns took 65
__global__ __launch_bounds__(256) void pointwise_fold_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                 const float* __restrict__ d, float* __restrict__ dw,
                                                                 float* __restrict__ dd, int cout, int cin) {
  __shared__ float part[16][17];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int ci = blockIdx.x * 16 + c;
  float acc = 0.f;
  if (ci < cin) {
    const float dv = d[ci];
#pragma unroll 4
    for (int co = grp; co < cout; co += 16) {
      const long long o = (long long)co * cin + ci;
      const float gv = g[o];
      acc = fmaf(w[o], gv, acc);
      dw[o] += gv * dv;
    }
  }
  part[grp][c] = acc;
  __syncthreads();
  if (grp == 0 && ci < cin) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][c];
    dd[ci] += t;
  }
}

}  // namespace os2s

using namespace os2s;

static int g_dw_ablate = 0;     // measurement hook (tools/bench_depthwise.py): parts of the matrix-core kernel switched off
static os2s::OptionReg r_dw_ablate("depthwise.ablate", [](double v) { g_dw_ablate = (int)v; });
static int g_dw_variant = -1;   // test / experiment hook: 0 = the generic kernels only
static os2s::OptionReg r_dw_variant("depthwise.variant", [](double v) { g_dw_variant = (int)v; });

static int dw_fill(DwArgs& a, int B, int Tin, int Tout, int C, int K, int stride, int dil, int padL) {
  OS2S_REQUIRE(B >= 1 && Tin >= 1 && Tout >= 1 && C >= 8 && C % 8 == 0 && K >= 1 && stride >= 1 && dil >= 1);
  a.B = B; a.Tin = Tin; a.Tout = Tout; a.C = C; a.K = K; a.stride = stride; a.dil = dil; a.padL = padL;
  a.R = (kDwBT - 1) * stride + (K - 1) * dil + 1;
  return OS2S_OK;
}

extern "C" int os2s_depthwise_conv1d_fwd(os2s_stream_t stream, const uint16_t* x, const float* w,
                                         uint16_t* y, const int32_t* in_len, const int32_t* out_len,
                                         int B, int Tin, int Tout, int C, int K, int stride, int dil,
                                         int padL, int flip_taps) {
  OS2S_REQUIRE(x && w && y);
  DwArgs a{};
  const int rc = dw_fill(a, B, Tin, Tout, C, K, stride, dil, padL);
  if (rc != OS2S_OK) return rc;
  a.x = (const bf16_t*)x; a.w = w; a.y = (bf16_t*)y; a.in_len = in_len; a.out_len = out_len;
  a.flip = flip_taps;
  // matrix-core kernel: stride 1, dilation 1, K <= 96 (depthwise.variant 1 = the register-window kernels only)
  if (stride == 1 && dil == 1 && K >= 2 && K <= 96 && g_dw_variant != 0 && g_dw_variant != 1) {
    const DmGeom g = dm_geom(K);
    const size_t ldsm = (size_t)kDmCh * g.plane_bytes + (size_t)8 * 2 * g.wt * sizeof(bf16_t);
    if (ldsm <= 160 * 1024) {
      static bool attrm = false;
      if (!attrm) {
        if (hipFuncSetAttribute((const void*)depthwise_mfma_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return OS2S_ERR_LAUNCH;
        attrm = true;
      }
      a.tile0 = g_dw_ablate;
      dim3 gridm(B * ceil_div(Tout, 32 * g.nseg), ceil_div(C, kDmCh));
      OS2S_LAUNCH(depthwise_mfma_fwd_kernel, gridm, dim3(512), ldsm, (hipStream_t)stream, a);
      return OS2S_OK;
    }
  }
  if (stride == 1 && (dil == 1 || dil == 2 || dil == 4) && g_dw_variant != 0) {
    const size_t lds16 = (((size_t)(kD16BT + (K - 1) * dil) * kD16Pitch + 15) & ~(size_t)15) + (size_t)K * kDwBC * sizeof(float);
    if (lds16 <= 160 * 1024) {             // <= 80 KB: two workgroups per CU (every dilation-1 QuartzNet layer)
      static bool attr16 = false;
      if (!attr16) {
        if (hipFuncSetAttribute((const void*)depthwise_fwd16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)depthwise_fwd16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)depthwise_fwd16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return OS2S_ERR_LAUNCH;
        attr16 = true;
      }
      dim3 grid16(B * ceil_div(Tout, kD16BT), ceil_div(C, kDwBC));
      if (dil == 1) { OS2S_LAUNCH(depthwise_fwd16_kernel<1>, grid16, dim3(256), lds16, (hipStream_t)stream, a); }
      else if (dil == 2) { OS2S_LAUNCH(depthwise_fwd16_kernel<2>, grid16, dim3(256), lds16, (hipStream_t)stream, a); }
      else { OS2S_LAUNCH(depthwise_fwd16_kernel<4>, grid16, dim3(256), lds16, (hipStream_t)stream, a); }
      return OS2S_OK;
    }
  }
  const size_t lds = ((size_t)a.R * kDwP + (size_t)K * kDwBC) * sizeof(float);
  if (lds > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)depthwise_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  dim3 grid(B * ceil_div(Tout, kDwBT), ceil_div(C, kDwBC));
  OS2S_LAUNCH(depthwise_fwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return OS2S_OK;
}

// The data gradient of a stride-1 / dilation-1 depthwise convolution as the LAST contribution to the gradient of a
// conv + BatchNorm + ReLU (+ dropout) layer's output (QuartzNet: the next separable layer's depthwise half), with
// that layer's activation backward and BatchNorm-backward partial sums in the store phase of the matrix-core
// kernel — the depthwise twin of os2s_conv1d_dgrad_bnact_ws.
extern "C" int os2s_depthwise_dgrad_bnact_num_parts(int B, int Tout, int K) {
  if (K < 2 || K > 96) return 0;
  return B * os2s::ceil_div(Tout, 32 * os2s::dm_geom(K).nseg);
}

extern "C" int os2s_depthwise_dgrad_bnact(os2s_stream_t stream, const uint16_t* dz, const float* w, uint16_t* dx,
                                          const uint16_t* addend, float* stats, const int32_t* out_len, int B,
                                          int Tin, int Tout, int C, int K, int padL, const uint16_t* mask_ref,
                                          float mask_scale, const uint16_t* stat_ref) {
  OS2S_REQUIRE(dz && w && dx && stats && mask_ref && stat_ref && K >= 2 && K <= 96);
  DwArgs a{};
  const int rc = dw_fill(a, B, Tin, Tout, C, K, 1, 1, padL);
  if (rc != OS2S_OK) return rc;
  a.x = (const bf16_t*)dz; a.w = w; a.y = (bf16_t*)dx; a.in_len = nullptr; a.out_len = out_len;
  a.flip = 1;
  a.mask_ref = (const bf16_t*)mask_ref; a.stat_ref = (const bf16_t*)stat_ref; a.addend = (const bf16_t*)addend;
  a.mask_scale = mask_scale; a.stats = stats;
  const DmGeom g = dm_geom(K);
  const size_t ldsm = (size_t)kDmCh * g.plane_bytes + (size_t)8 * 2 * g.wt * sizeof(bf16_t);
  OS2S_REQUIRE(ldsm <= 160 * 1024 && ldsm >= 512 * 16 * sizeof(float));
  static bool attrm = false;
  if (!attrm) {
    if (hipFuncSetAttribute((const void*)depthwise_mfma_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OS2S_ERR_LAUNCH;
    attrm = true;
  }
  dim3 gridm(B * ceil_div(Tout, 32 * g.nseg), ceil_div(C, kDmCh));
  OS2S_LAUNCH(depthwise_mfma_fwd_kernel, gridm, dim3(512), ldsm, (hipStream_t)stream, a);
  return OS2S_OK;
}

extern "C" int os2s_depthwise_conv1d_wgrad(os2s_stream_t stream, const uint16_t* x, const uint16_t* dy,
                                           float* dw, const int32_t* in_len, int B, int Tin, int Tout,
                                           int C, int K, int stride, int dil, int padL) {
  OS2S_REQUIRE(x && dy && dw);
  DwArgs a{};
  const int rc = dw_fill(a, B, Tin, Tout, C, K, stride, dil, padL);
  if (rc != OS2S_OK) return rc;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw; a.in_len = in_len;
  // matrix-core kernel: stride 1, dilation 1, 2 <= K <= 96 (depthwise.variant 1 = the register-window kernels only)
  if (stride == 1 && dil == 1 && K >= 2 && K <= 96 && g_dw_variant != 0 && g_dw_variant != 1) {
    const size_t ldsw = (size_t)kDwgCh * (kDwgXPlane + kDwgYPlane);
    static bool attrw = false;
    if (!attrw) {
      if (hipFuncSetAttribute((const void*)depthwise_mfma_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return OS2S_ERR_LAUNCH;
      attrw = true;
    }
    const int ntiles = B * ceil_div(Tout, kDwgTt), ngroups = ceil_div(C, kDwgCh);
    // enough workgroups for two per CU; deterministic mode: ONE per channel block (one add per dw element)
    int gx = os2s_deterministic() ? 1 : ceil_div(512, ngroups);
    gx = gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
    a.tile0 = g_dw_ablate;
    OS2S_LAUNCH(depthwise_mfma_wgrad_kernel, dim3(gx, ngroups), dim3(512), ldsw, (hipStream_t)stream, a);
    return OS2S_OK;
  }
  if (stride == 1 && (dil == 1 || dil == 2 || dil == 4) && g_dw_variant != 0) {
    const int ngrp = ceil_div(K, 16), ncell = ngrp * 16;
    int nseg = 256 / ncell;
    nseg = nseg > 8 ? 8 : nseg;
    nseg = (nseg / dil) * dil;             // whole residue classes
    const size_t tiles = (size_t)(kD16BT + 16 * ngrp * dil) * kD16Pitch + (size_t)kD16BT * 128;
    const size_t redb = (size_t)(nseg > 0 ? nseg : 1) * ncell * 64 * sizeof(float);
    const size_t lds16 = tiles > redb ? tiles : redb;
    if (nseg >= 1 && lds16 <= 160 * 1024) {
      static bool attr16 = false;
      if (!attr16) {
        if (hipFuncSetAttribute((const void*)depthwise_wgrad16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)depthwise_wgrad16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)depthwise_wgrad16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return OS2S_ERR_LAUNCH;
        attr16 = true;
      }
      const int ntiles = B * ceil_div(Tout, kD16BT);
      // deterministic mode: ONE workgroup per channel block walks every tile (one add per dw element)
      dim3 grid16(os2s_deterministic() ? 1 : (ntiles < 64 ? ntiles : 64), ceil_div(C, kDwBC));
      if (dil == 1) { OS2S_LAUNCH(depthwise_wgrad16_kernel<1>, grid16, dim3(256), lds16, (hipStream_t)stream, a, ngrp, nseg); }
      else if (dil == 2) { OS2S_LAUNCH(depthwise_wgrad16_kernel<2>, grid16, dim3(256), lds16, (hipStream_t)stream, a, ngrp, nseg); }
      else { OS2S_LAUNCH(depthwise_wgrad16_kernel<4>, grid16, dim3(256), lds16, (hipStream_t)stream, a, ngrp, nseg); }
      return OS2S_OK;
    }
  }
  const size_t lds = ((size_t)a.R + kDwBT) * kDwP * sizeof(float);
  if (lds > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)depthwise_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  dim3 grid(B * ceil_div(Tout, kDwBT), ceil_div(C, kDwBC));
  if (os2s_deterministic()) {     // one (sample, time tile) per launch: the adds of a dw element arrive in tile order
    const int ntiles = (int)grid.x;
    grid.x = 1;
    for (a.tile0 = 0; a.tile0 < ntiles; ++a.tile0)
      OS2S_LAUNCH(depthwise_wgrad_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    return OS2S_OK;
  }
  OS2S_LAUNCH(depthwise_wgrad_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return OS2S_OK;
}

extern "C" int os2s_pointwise_fold(os2s_stream_t stream, const float* w, const float* d, uint16_t* w_eff,
                                   uint16_t* wt_eff, int cout, int cin) {
  OS2S_REQUIRE(w && d && w_eff && wt_eff && cout >= 1 && cin >= 1);
  dim3 grid(ceil_div(cin, 32), ceil_div(cout, 32));
  OS2S_LAUNCH(pointwise_fold_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, d, (bf16_t*)w_eff, (bf16_t*)wt_eff,
              cout, cin);
  return OS2S_OK;
}

extern "C" int os2s_pointwise_fold_bwd(os2s_stream_t stream, const float* g, const float* w, const float* d,
                                       float* dw, float* dd, int cout, int cin) {
  OS2S_REQUIRE(g && w && d && dw && dd && cout >= 1 && cin >= 1);
  OS2S_LAUNCH(pointwise_fold_bwd_kernel, dim3(ceil_div(cin, 16)), dim3(256), 0, (hipStream_t)stream, g, w, d, dw, dd,
              cout, cin);
  return OS2S_OK;
}
