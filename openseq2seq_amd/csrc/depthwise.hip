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
  const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * kDwBT, c0 = blockIdx.y * kDwBC;
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

}  // namespace os2s

using namespace os2s;

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
  const size_t lds = ((size_t)a.R * kDwP + (size_t)K * kDwBC) * sizeof(float);
  if (lds > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)depthwise_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  dim3 grid(B * ceil_div(Tout, kDwBT), ceil_div(C, kDwBC));
  OS2S_LAUNCH(depthwise_fwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
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
  const size_t lds = ((size_t)a.R + kDwBT) * kDwP * sizeof(float);
  if (lds > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)depthwise_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  dim3 grid(B * ceil_div(Tout, kDwBT), ceil_div(C, kDwBC));
  OS2S_LAUNCH(depthwise_wgrad_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return OS2S_OK;
}
