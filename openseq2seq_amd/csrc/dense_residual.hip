// Dense-residual block ends without branch tensors (conv_bn_res_bn_actv, parts/cnns/conv_blocks.py:61-168, with
// the dense residual list of encoders/tdnn_encoder.py:188-192).
//
// A block end k of the reference sums, before its activation,
//     BN_main(conv(x)) + sum_{i <= k} BN_ik(conv1x1_ik(r_i))          r_i = input of block i  ("source" i)
// — up to 10 branches per block end, 55 per pass of Jasper 10x5, each a [rows, Cout] tensor that is written,
// re-read by its BatchNorm, re-read twice in backward and paired with an equally large dy_ik. Every one of those
// tensors is a LINEAR image of a source, so the BatchNorm statistics, the sum and all gradients follow from
//     s_i = sum_rows r_i            [c_i]          (column sums)
//     G_i = r_i^T r_i               [c_i, c_i]     (Gram matrix, one TN GEMM per source and step)
//     P_k = [r_0 .. r_k]^T dz_k     [K_k, Cout_k]  (one TN GEMM per block end: dz_k = gradient at the sum)
// with N = rows (padded frames included, as tf.layers.batch_normalization counts them), m = s / N,
// C = G / N - m m^T, W = W_ik [Cout, c]:
//     mean = W m,   var[co] = sum_a W[co,a] (W C)[co,a],   scale = gamma rstd,   shift = beta - mean scale
//     sum of the branches = [r_0 .. r_k] . [W_0k scale_0k | .. | W_kk scale_kk]^T + sum_i shift_ik      (ONE GEMM)
//     q[co] = sum_a W[co,a] P[co,a] (= sum_rows dz y),  dbeta = sum_rows dz,  dgamma = rstd (q - mean dbeta)
//     d1 = gamma rstd,  d2 = d1 rstd dgamma / N
//     dW = d1 P - (d1 dbeta / N) s^T - N d2 (W C)
//     dr_i = sum_k dz_k (W_ik d1)  -  r_i . sum_k W_ik^T d2 W_ik  +  1 . sum_k (mean d2 - d1 dbeta / N)^T W_ik
// (checked in fp64 against autograd: tests/test_dense_residual_algebra.py; on the device against the branch-tensor path:
// tests/test_dense_residual_gpu.py). This file holds the small kernels between the GEMMs: the masked copy of a
// source into the concatenated buffer with its column sums, the covariance split into a bf16 hi / lo pair (the
// product W C runs on the bf16 matrix cores with both halves: 16 mantissa bits), the per-block-end statistics +
// scaled weight stack, and the backward coefficients + gradient / transposed-stack writer.
#include <mutex>

#include "os2s_common.hpp"

namespace os2s {

typedef os2s_dres_seg_t DresSeg;

constexpr int kDresCopyRows = 128;     // rows of one sample per workgroup of the copy kernel
constexpr int kDresCopyWaves = 4;      // ... and one partial row of column sums per wave

// dst[b, t, 0:C] = t < lens[b] ? src[b, t, 0:C] : 0 (row strides src_ld / dst_ld elements), and
// partial[(b * nblk + blk) * 4 + wave][c] = sum over the wave's live rows (fp32; nullptr = no sums).
// No LDS and no barrier: these launches run on a side stream NEXT TO the ping-pong convolutions, whose workgroups
// hold a CU's whole 160 KB — a kernel that asks for even 8 KB waits for a CU without one (measured: 114 us per
// launch instead of 16). A lane owns one 8-channel group (blockIdx.y walks chunks of 64 groups), a wave every
// fourth row of the block.
__global__ __launch_bounds__(256) void dres_copy_cols_kernel(const bf16_t* __restrict__ src, long long src_ld,
                                                             bf16_t* __restrict__ dst, long long dst_ld,
                                                             const int32_t* __restrict__ lens, int T, int C,
                                                             float* __restrict__ partial) {
  const int nblk = (T + kDresCopyRows - 1) / kDresCopyRows;
  const int b = blockIdx.x / nblk, blk = blockIdx.x - b * nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = blockIdx.y * 64 + lane;
  if (cg * 8 >= C) return;
  int len = T;
  if (lens) { const int l = lens[b]; len = l < 0 ? 0 : (l < T ? l : T); }
  const int t0 = blk * kDresCopyRows;
  const int t1 = min(T, t0 + kDresCopyRows);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  int t = t0 + wave;
  for (; t + 3 * kDresCopyWaves < t1; t += 4 * kDresCopyWaves) {        // four rows in flight per lane
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = t + u * kDresCopyWaves;
      v[u] = tt < len ? *reinterpret_cast<const u32x4*>(src + ((long long)b * T + tt) * src_ld + cg * 8) : zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[2 * e] += bflo(v[u][e]); s[2 * e + 1] += bfhi(v[u][e]); }
      *reinterpret_cast<u32x4*>(dst + ((long long)b * T + t + u * kDresCopyWaves) * dst_ld + cg * 8) = v[u];
    }
  }
  for (; t < t1; t += kDresCopyWaves) {
    const long long row = (long long)b * T + t;
    const u32x4 v = t < len ? *reinterpret_cast<const u32x4*>(src + row * src_ld + cg * 8) : zero;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += bflo(v[e]); s[2 * e + 1] += bfhi(v[e]); }
    *reinterpret_cast<u32x4*>(dst + row * dst_ld + cg * 8) = v;
  }
  if (partial) {
    float* const pr = partial + ((long long)blockIdx.x * kDresCopyWaves + wave) * C + cg * 8;
    *reinterpret_cast<f32x4*>(pr) = f32x4{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f32x4*>(pr + 4) = f32x4{s[4], s[5], s[6], s[7]};
  }
}

// s[c] = sum of the partials (fp64), m[c] = s / count. One wave per 16 channels: 4 lanes walk the partial rows of
// a channel, two shuffles add them (no LDS: see dres_copy_cols_kernel).
__global__ __launch_bounds__(64) void dres_colsum_finalize_kernel(const float* __restrict__ partial, int nparts, int C,
                                                                 double count, float* __restrict__ s,
                                                                 float* __restrict__ m) {
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0;
  if (c < C) {
    int i = pl;
    for (; i + 12 < nparts; i += 16) {
      const float v0 = partial[(long long)i * C + c], v1 = partial[(long long)(i + 4) * C + c];
      const float v2 = partial[(long long)(i + 8) * C + c], v3 = partial[(long long)(i + 12) * C + c];
      a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; i < nparts; i += 4) a += (double)partial[(long long)i * C + c];
  }
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  if (pl != 0 || c >= C) return;
  s[c] = (float)a;
  m[c] = (float)(a / count);
}

// chl[a][b] = bf16(cov), chl[C + a][b] = bf16(cov - hi),  cov = G[a][b] / count - m[a] m[b]
__global__ __launch_bounds__(256) void dres_cov_split_kernel(const float* __restrict__ G, const float* __restrict__ m,
                                                             int C, float inv_count, bf16_t* __restrict__ chl) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= (long long)C * C) return;
  const int a = (int)(i4 / C), b0 = (int)(i4 - (long long)a * C);       // C % 4 == 0: one row per thread
  const f32x4 g = *reinterpret_cast<const f32x4*>(G + i4);
  const f32x4 mb = *reinterpret_cast<const f32x4*>(m + b0);
  const float ma = m[a];
  float cov[4], hi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cov[e] = g[e] * inv_count - ma * mb[e];
    hi[e] = bf2f(f2bf(cov[e]));
  }
  u32x2 h, l;
  h[0] = pack2bf(cov[0], cov[1]); h[1] = pack2bf(cov[2], cov[3]);
  l[0] = pack2bf(cov[0] - hi[0], cov[1] - hi[1]); l[1] = pack2bf(cov[2] - hi[2], cov[3] - hi[3]);
  *reinterpret_cast<u32x2*>(chl + i4) = h;
  *reinterpret_cast<u32x2*>(chl + (long long)C * C + i4) = l;
}

// Block end, forward: one wave per output channel co walks the branches (segments). Training: batch statistics
// from m and tt = W [C_hi | C_lo] (fp32 [Cout, 2c]); else the moving statistics. Writes the BN-scaled weight row
// into the stacked matrix wp [Cout, Kk] and the sum of the branches' shifts.
__global__ __launch_bounds__(256) void dres_bn_fwd_kernel(const DresSeg* __restrict__ segs, int nseg, int Cout, int Kk,
                                                          bf16_t* __restrict__ wp, float* __restrict__ shift,
                                                          double count, float eps, float momentum, int training) {
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (co >= Cout) return;
  float shift_acc = 0.f;
  for (int si = 0; si < nseg; ++si) {
    const DresSeg& S = segs[si];
    const int c = S.c;
    const bf16_t* const wrow = S.w + (long long)co * c;
    float mean, var;
    if (training) {
      const float* const th = S.tt + (long long)co * 2 * c;
      const float* const tl = th + c;
      float am = 0.f, av = 0.f;
      for (int a = lane * 8; a < c; a += 512) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wrow + a);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(th + a), h1 = *reinterpret_cast<const f32x4*>(th + a + 4);
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(tl + a), l1 = *reinterpret_cast<const f32x4*>(tl + a + 4);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(S.m + a), m1 = *reinterpret_cast<const f32x4*>(S.m + a + 4);
        float w[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { w[2 * e] = bflo(wv[e]); w[2 * e + 1] = bfhi(wv[e]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          am += w[e] * m0[e] + w[4 + e] * m1[e];
          av += w[e] * (h0[e] + l0[e]) + w[4 + e] * (h1[e] + l1[e]);
        }
      }
      mean = wave_sum(am);
      var = fmaxf(wave_sum(av), 0.f);
      if (lane == 0) {
        S.mean[co] = mean;
        if (S.moving_mean) {
          const double unbiased = count > 1.0 ? (double)var * count / (count - 1.0) : (double)var;
          S.moving_mean[co] = S.moving_mean[co] * momentum + mean * (1.f - momentum);
          S.moving_var[co] = S.moving_var[co] * momentum + (float)unbiased * (1.f - momentum);
        }
      }
    } else {
      mean = S.moving_mean[co];
      var = S.moving_var[co];
    }
    const float rstd = rsqrtf(var + eps);
    const float scale = S.gamma[co] * rstd;
    shift_acc += S.beta[co] - mean * scale;
    if (training && lane == 0) S.rstd[co] = rstd;
    bf16_t* const out = wp + (long long)co * Kk + S.koff;
    for (int a = lane * 8; a < c; a += 512) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(wrow + a);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(bflo(wv[e]) * scale, bfhi(wv[e]) * scale);
      *reinterpret_cast<u32x4*>(out + a) = o;
    }
  }
  if (lane == 0) shift[co] = shift_acc;
}

// Block end, backward, pass 1: one wave per co. q = sum_a W P -> dgamma, dbeta (accumulated into the gradient
// vectors), the four coefficients of pass 2 per (branch, co), and the row `c` of the source's stacked -W^T d2
// matrix: e = mean d2 - d1 dbeta / N (the gradient's constant row comes out of the same GEMM as the matrix).
__global__ __launch_bounds__(256) void dres_bn_bwd_coef_kernel(const DresSeg* __restrict__ segs, int nseg, int Cout,
                                                               int Kk, const float* __restrict__ P,
                                                               const float* __restrict__ mean_dz, float count,
                                                               float* __restrict__ coef) {
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (co >= Cout) return;
  const float dbeta = mean_dz[co] * count;
  for (int si = 0; si < nseg; ++si) {
    const DresSeg& S = segs[si];
    const int c = S.c;
    const bf16_t* const wrow = S.w + (long long)co * c;
    const float* const prow = P + (long long)co * Kk + S.koff;
    float q = 0.f;
    for (int a = lane * 8; a < c; a += 512) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(wrow + a);
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(prow + a), p1 = *reinterpret_cast<const f32x4*>(prow + a + 4);
      q += bflo(wv[0]) * p0[0] + bfhi(wv[0]) * p0[1] + bflo(wv[1]) * p0[2] + bfhi(wv[1]) * p0[3] +
           bflo(wv[2]) * p1[0] + bfhi(wv[2]) * p1[1] + bflo(wv[3]) * p1[2] + bfhi(wv[3]) * p1[3];
    }
    q = wave_sum(q);
    if (lane == 0) {
      const float mean = S.mean[co], rstd = S.rstd[co];
      const float dgamma = rstd * (q - mean * dbeta);
      const float d1 = S.gamma[co] * rstd;
      const float d2 = d1 * rstd * dgamma / count;
      S.dgamma[co] += dgamma;
      S.dbeta[co] += dbeta;
      float* const cf = coef + (long long)si * 4 * Cout;
      cf[co] = d1;
      cf[Cout + co] = d2;
      cf[2 * Cout + co] = d1 * dbeta / count;
      cf[3 * Cout + co] = count * d2;
      S.wd2[(long long)c * S.ld + co] = f2bf(mean * d2 - d1 * dbeta / count);
    }
  }
}

// Block end, backward, pass 2: 64 (co) x 64 (a) tiles. dW += d1 P - cb s^T - cn (W C); the three transposed,
// per-source stacks wd1[a][co] = W d1 (data-gradient GEMM), wd2[a][co] = -W d2, wt[a][co] = W (their product over
// all block ends is the [c, c] matrix the source itself is multiplied by).
__global__ __launch_bounds__(256) void dres_bn_bwd_apply_kernel(const DresSeg* __restrict__ segs, int nseg, int Cout,
                                                                int Kk, const float* __restrict__ P,
                                                                const float* __restrict__ coef) {
  constexpr int LP = 72;                       // LDS pitch (bf16 elements): 144 B rows, 16-B aligned
  __shared__ __attribute__((aligned(16))) bf16_t t1[64 * LP], t2[64 * LP], t3[64 * LP];
  const int ag = blockIdx.x * 64;              // first concatenated channel of the tile
  const int co0 = blockIdx.y * 64;
  int si = 0;
  for (int i = 1; i < nseg; ++i)
    if (ag >= segs[i].koff) si = i;            // segments are listed in ascending koff
  const DresSeg& S = segs[si];
  const int c = S.c, a0 = ag - S.koff;         // c % 64 == 0: a tile never straddles two sources
  const float* const cf = coef + (long long)si * 4 * Cout;
  const int r0 = threadIdx.x >> 3, c8 = threadIdx.x & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = r0 + 32 * h, co = co0 + r;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = 0.f;
    float d1 = 0.f, d2 = 0.f;
    if (co < Cout) {
      const int a = a0 + c8 * 8;
      const u32x4 wv = *reinterpret_cast<const u32x4*>(S.w + (long long)co * c + a);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[2 * e] = bflo(wv[e]); w[2 * e + 1] = bfhi(wv[e]); }
      d1 = cf[co]; d2 = cf[Cout + co];
      const float cb = cf[2 * Cout + co], cn = cf[3 * Cout + co];
      const float* const pp = P + (long long)co * Kk + ag + c8 * 8;
      const float* const th = S.tt + (long long)co * 2 * c + a;
      float* const dw = S.dw + (long long)co * c + a;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(pp + 4 * v);
        const f32x4 hh = *reinterpret_cast<const f32x4*>(th + 4 * v);
        const f32x4 ll = *reinterpret_cast<const f32x4*>(th + c + 4 * v);
        const f32x4 ss = *reinterpret_cast<const f32x4*>(S.s + a + 4 * v);
        f32x4 g = *reinterpret_cast<const f32x4*>(dw + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += d1 * p[e] - cb * ss[e] - cn * (hh[e] + ll[e]);
        *reinterpret_cast<f32x4*>(dw + 4 * v) = g;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int al = c8 * 8 + e;
      t1[al * LP + r] = f2bf(w[e] * d1);
      t2[al * LP + r] = f2bf(-w[e] * d2);
      t3[al * LP + r] = f2bf(w[e]);
    }
  }
  __syncthreads();
  const bool full = co0 + 64 <= Cout;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int al = r0 + 32 * h;
    const long long off = (long long)(a0 + al) * S.ld + co0 + c8 * 8;
    if (full || co0 + c8 * 8 + 8 <= Cout) {       // Cout % 8 == 0
      *reinterpret_cast<u32x4*>(S.wd1 + off) = *reinterpret_cast<const u32x4*>(t1 + al * LP + c8 * 8);
      *reinterpret_cast<u32x4*>(S.wd2 + off) = *reinterpret_cast<const u32x4*>(t2 + al * LP + c8 * 8);
      *reinterpret_cast<u32x4*>(S.wt + off) = *reinterpret_cast<const u32x4*>(t3 + al * LP + c8 * 8);
    }
  }
}

}  // namespace os2s

extern "C" int os2s_dres_copy_num_parts(int B, int T) {
  return B * os2s::ceil_div(T, os2s::kDresCopyRows) * os2s::kDresCopyWaves;
}

extern "C" int os2s_dres_copy_cols(os2s_stream_t stream, const uint16_t* src, long long src_row_stride, uint16_t* dst,
                                   long long dst_row_stride, const int32_t* lens, int B, int T, int C,
                                   float* colsum_partial) {
  using namespace os2s;
  OS2S_REQUIRE(src && dst && B >= 0 && T >= 1 && C >= 8 && C % 8 == 0);
  OS2S_REQUIRE(src_row_stride >= C && dst_row_stride >= C && src_row_stride % 8 == 0 && dst_row_stride % 8 == 0);
  if (B == 0) return OS2S_OK;
  OS2S_LAUNCH(dres_copy_cols_kernel, dim3(B * ceil_div(T, kDresCopyRows), ceil_div(C / 8, 64)), dim3(256), 0,
              (hipStream_t)stream, src, src_row_stride, dst, dst_row_stride, lens, T, C, colsum_partial);
  return OS2S_OK;
}

extern "C" int os2s_dres_cov(os2s_stream_t stream, const float* colsum_partial, int nparts, const float* gram, int C,
                             long long count, float* s, float* m, uint16_t* chl) {
  using namespace os2s;
  OS2S_REQUIRE(colsum_partial && gram && s && m && chl && nparts >= 1 && C >= 8 && C % 8 == 0 && count >= 1);
  OS2S_LAUNCH(dres_colsum_finalize_kernel, dim3(ceil_div(C, 16)), dim3(64), 0, (hipStream_t)stream, colsum_partial,
              nparts, C, (double)count, s, m);
  OS2S_LAUNCH(dres_cov_split_kernel, dim3(ceil_div((long long)C * C / 4, 256)), dim3(256), 0, (hipStream_t)stream,
              gram, (const float*)m, C, (float)(1.0 / (double)count), chl);
  return OS2S_OK;
}

extern "C" int os2s_dres_bn_fwd(os2s_stream_t stream, const os2s_dres_seg_t* segs_dev, int nseg, int Cout, int Kk,
                                uint16_t* wp, float* shift, long long count, float eps, float momentum,
                                int training) {
  using namespace os2s;
  OS2S_REQUIRE(segs_dev && nseg >= 1 && nseg <= 16 && Cout >= 8 && Kk >= 8 && Kk % 8 == 0 && wp && shift && count >= 1);
  OS2S_LAUNCH(dres_bn_fwd_kernel, dim3(ceil_div(Cout, 4)), dim3(256), 0, (hipStream_t)stream, segs_dev, nseg, Cout, Kk,
              wp, shift, (double)count, eps, momentum, training);
  return OS2S_OK;
}

extern "C" int os2s_dres_bn_bwd(os2s_stream_t stream, const os2s_dres_seg_t* segs_dev, int nseg, int Cout, int Kk,
                                const float* P, const float* mean_dz, long long count, float* coef) {
  using namespace os2s;
  OS2S_REQUIRE(segs_dev && nseg >= 1 && nseg <= 16 && Cout >= 8 && Cout % 8 == 0 && Kk >= 64 && Kk % 64 == 0);
  OS2S_REQUIRE(P && mean_dz && coef && count >= 1);
  OS2S_LAUNCH(dres_bn_bwd_coef_kernel, dim3(ceil_div(Cout, 4)), dim3(256), 0, (hipStream_t)stream, segs_dev, nseg, Cout,
              Kk, P, mean_dz, (float)count, coef);
  OS2S_LAUNCH(dres_bn_bwd_apply_kernel, dim3(Kk / 64, ceil_div(Cout, 64)), dim3(256), 0, (hipStream_t)stream, segs_dev,
              nseg, Cout, Kk, P, (const float*)coef);
  return OS2S_OK;
}
