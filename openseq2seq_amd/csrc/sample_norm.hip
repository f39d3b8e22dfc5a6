// Per-sample normalisations of the TDNN encoder: tf.contrib.layers.layer_norm and
// tf.contrib.layers.instance_norm as conv_ln_actv / conv_in_actv call them
// (open_seq2seq/parts/cnns/conv_blocks.py:234-309) on a channels-last [B, T, C] convolution output:
//   mode 0, instance norm: statistics per (sample, channel) over the T frames of the PADDED tensor,
//           epsilon 1e-6, gamma / beta per channel;
//   mode 1, layer norm (begin_norm_axis = 1, begin_params_axis = -1: the defaults the reference
//           leaves in place): statistics per sample over all T x C values, epsilon 1e-12, gamma / beta
//           per channel.
// Both are "normalise a sample by its own statistics": one set of kernels. The activation, dropout and
// the sequence mask that follow run on the BatchNorm apply kernels with the identity transform (as for
// normalization = None, parts/cnns/conv_blocks.py:conv_actv), so these kernels are the pure
// normalisation z = gamma * xhat + beta and its gradient.
//
// HBM-bound and not on any BASELINE configuration's path: three plain passes (partial sums over
// 8 time chunks per sample -> finalize -> apply), a thread owns one channel pair and walks rows (4-byte
// loads, coalesced over the channel axis), reductions in a fixed order (deterministic), fp64 in the
// finalize steps (variance as E[x^2] - mean^2).
#include "os2s_common.hpp"

namespace os2s {

constexpr int kSnChunks = 8;       // time chunks per sample (partial sums)

// BWD = false: (sum x, sum x^2); BWD = true: (sum dz, sum dz * xhat) — per (sample, chunk, channel)
template <bool BWD>
__global__ __launch_bounds__(256) void sn_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, int T, int C,
                                                         int rows_per_chunk, float* __restrict__ part) {
  const int c2 = blockIdx.x * 256 + threadIdx.x;
  const int ts = blockIdx.y, b = blockIdx.z;
  if (c2 * 2 >= C) return;
  const int t0 = ts * rows_per_chunk;
  const int t1 = min(T, t0 + rows_per_chunk);
  const int ld = C >> 1;
  const uint32_t* xr = reinterpret_cast<const uint32_t*>(x) + ((long long)b * T + t0) * ld + c2;
  const uint32_t* dr = BWD ? reinterpret_cast<const uint32_t*>(dz) + ((long long)b * T + t0) * ld + c2 : nullptr;
  float m0 = 0.f, m1 = 0.f, r0 = 1.f, r1 = 1.f;
  if (BWD) {
    m0 = mean[(long long)b * C + 2 * c2]; m1 = mean[(long long)b * C + 2 * c2 + 1];
    r0 = rstd[(long long)b * C + 2 * c2]; r1 = rstd[(long long)b * C + 2 * c2 + 1];
  }
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  int t = t0;
  for (; t + 4 <= t1; t += 4) {
    uint32_t xv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = xr[(long long)u * ld];
      if (BWD) dv[u] = dr[(long long)u * ld];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a0 = bflo(xv[u]), a1 = bfhi(xv[u]);
      if (BWD) {
        const float g0 = bflo(dv[u]), g1 = bfhi(dv[u]);
        s0 += g0; s1 += g1;
        q0 += g0 * ((a0 - m0) * r0); q1 += g1 * ((a1 - m1) * r1);
      } else {
        s0 += a0; s1 += a1;
        q0 += a0 * a0; q1 += a1 * a1;
      }
    }
    xr += 4LL * ld;
    if (BWD) dr += 4LL * ld;
  }
  for (; t < t1; ++t) {
    const uint32_t xv = *xr;
    const float a0 = bflo(xv), a1 = bfhi(xv);
    if (BWD) {
      const uint32_t dv = *dr;
      const float g0 = bflo(dv), g1 = bfhi(dv);
      s0 += g0; s1 += g1;
      q0 += g0 * ((a0 - m0) * r0); q1 += g1 * ((a1 - m1) * r1);
      dr += ld;
    } else {
      s0 += a0; s1 += a1;
      q0 += a0 * a0; q1 += a1 * a1;
    }
    xr += ld;
  }
  float* const p = part + (((long long)b * kSnChunks + ts) * 2) * C + 2 * c2;     // [b][chunk][2][C]
  p[0] = s0; p[1] = s1;
  p[C] = q0; p[C + 1] = q1;
}

__device__ __forceinline__ double sn_block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// one workgroup per sample: mean / rstd [B, C] (layer mode: the sample's value in every channel)
__global__ __launch_bounds__(256) void sn_fwd_finalize_kernel(const float* __restrict__ part, int T, int C, int mode,
                                                              float eps, float* __restrict__ mean,
                                                              float* __restrict__ rstd) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* const pb = part + (long long)b * kSnChunks * 2 * C;
  double ls = 0.0, lq = 0.0;
  for (int c = tid; c < C; c += 256) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int ts = 0; ts < kSnChunks; ++ts) {
      s += (double)pb[(long long)ts * 2 * C + c];
      q += (double)pb[(long long)ts * 2 * C + C + c];
    }
    if (mode == 0) {
      const double m = s / T;
      double var = q / T - m * m;
      var = var > 0.0 ? var : 0.0;
      mean[(long long)b * C + c] = (float)m;
      rstd[(long long)b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      ls += s; lq += q;
    }
  }
  if (mode != 0) {
    const double n = (double)T * (double)C;
    const double S = sn_block_sum(ls, red), Q = sn_block_sum(lq, red);
    const double m = S / n;
    double var = Q / n - m * m;
    var = var > 0.0 ? var : 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid; c < C; c += 256) {
      mean[(long long)b * C + c] = mf;
      rstd[(long long)b * C + c] = rf;
    }
  }
}

// forward: z = gamma * (x - mean) * rstd + beta; backward: dx = rstd * (gamma * dz - m1 - xhat * m2)
template <bool BWD>
__global__ __launch_bounds__(256) void sn_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ m1, const float* __restrict__ m2,
                                                       int T, int C, int rows_per_chunk, bf16_t* __restrict__ out) {
  const int c2 = blockIdx.x * 256 + threadIdx.x;
  const int ts = blockIdx.y, b = blockIdx.z;
  if (c2 * 2 >= C) return;
  const int t0 = ts * rows_per_chunk;
  const int t1 = min(T, t0 + rows_per_chunk);
  const int ld = C >> 1;
  const long long bc = (long long)b * C + 2 * c2;
  const float g0 = gamma[2 * c2], g1 = gamma[2 * c2 + 1];
  const float mu0 = mean[bc], mu1 = mean[bc + 1], r0 = rstd[bc], r1 = rstd[bc + 1];
  float a0, a1, b0, b1;
  if (BWD) { a0 = m1[bc]; a1 = m1[bc + 1]; b0 = m2[bc]; b1 = m2[bc + 1]; }
  else { a0 = beta[2 * c2]; a1 = beta[2 * c2 + 1]; b0 = b1 = 0.f; }
  const long long base = ((long long)b * T + t0) * ld + c2;
  const uint32_t* xr = reinterpret_cast<const uint32_t*>(x) + base;
  const uint32_t* dr = BWD ? reinterpret_cast<const uint32_t*>(dz) + base : nullptr;
  uint32_t* o = reinterpret_cast<uint32_t*>(out) + base;
  for (int t = t0; t < t1; ++t) {
    const uint32_t xv = *xr;
    const float h0 = (bflo(xv) - mu0) * r0, h1 = (bfhi(xv) - mu1) * r1;
    if (BWD) {
      const uint32_t dv = *dr;
      *o = pack2bf(r0 * (g0 * bflo(dv) - a0 - h0 * b0), r1 * (g1 * bfhi(dv) - a1 - h1 * b1));
      dr += ld;
    } else {
      *o = pack2bf(g0 * h0 + a0, g1 * h1 + a1);
    }
    xr += ld; o += ld;
  }
}

// one workgroup per sample: sums [b][2][C] = (sum_t dz, sum_t dz * xhat); m1 / m2 [B, C] = the two group means
// of gamma * dz and gamma * dz * xhat (over t for instance norm, over (t, c) for layer norm)
__global__ __launch_bounds__(256) void sn_bwd_finalize_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ gamma, int T, int C, int mode,
                                                              float* __restrict__ sums, float* __restrict__ m1,
                                                              float* __restrict__ m2) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* const pb = part + (long long)b * kSnChunks * 2 * C;
  double la = 0.0, lb = 0.0;
  for (int c = tid; c < C; c += 256) {
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int ts = 0; ts < kSnChunks; ++ts) {
      a += (double)pb[(long long)ts * 2 * C + c];
      q += (double)pb[(long long)ts * 2 * C + C + c];
    }
    sums[((long long)b * 2) * C + c] = (float)a;
    sums[((long long)b * 2 + 1) * C + c] = (float)q;
    const double g = (double)gamma[c];
    if (mode == 0) {
      m1[(long long)b * C + c] = (float)(g * a / T);
      m2[(long long)b * C + c] = (float)(g * q / T);
    } else {
      la += g * a; lb += g * q;
    }
  }
  if (mode != 0) {
    const double n = (double)T * (double)C;
    const float A = (float)(sn_block_sum(la, red) / n), Q = (float)(sn_block_sum(lb, red) / n);
    for (int c = tid; c < C; c += 256) {
      m1[(long long)b * C + c] = A;
      m2[(long long)b * C + c] = Q;
    }
  }
}

// dgamma[c] += sum_b sum_t dz * xhat, dbeta[c] += sum_b sum_t dz (samples in order: deterministic)
__global__ __launch_bounds__(256) void sn_param_grad_kernel(const float* __restrict__ sums, int B, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, q = 0.f;
  for (int b = 0; b < B; ++b) {
    a += sums[((long long)b * 2) * C + c];
    q += sums[((long long)b * 2 + 1) * C + c];
  }
  dbeta[c] += a;
  dgamma[c] += q;
}

}  // namespace os2s

using namespace os2s;

extern "C" size_t os2s_sample_norm_partial_floats(int B, int C) {
  return (size_t)B * kSnChunks * 2 * (size_t)C;
}

extern "C" int os2s_sample_norm_fwd(os2s_stream_t stream, const uint16_t* x, const float* gamma, const float* beta,
                                    int B, int T, int C, int mode, float eps, uint16_t* z, float* mean, float* rstd,
                                    float* partial) {
  OS2S_REQUIRE(x && gamma && beta && z && mean && rstd && partial);
  OS2S_REQUIRE(B >= 1 && B <= 65535 && T >= 1 && C >= 2 && C % 2 == 0 && (mode == 0 || mode == 1) && eps > 0.f);
  const int rpc = (T + kSnChunks - 1) / kSnChunks;
  const dim3 grid((unsigned)((C / 2 + 255) / 256), kSnChunks, (unsigned)B);
  OS2S_LAUNCH(sn_partial_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)nullptr,
              (const float*)nullptr, (const float*)nullptr, T, C, rpc, partial);
  OS2S_LAUNCH(sn_fwd_finalize_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const float*)partial,
              T, C, mode, eps, mean, rstd);
  OS2S_LAUNCH(sn_apply_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)nullptr, gamma, beta,
              (const float*)mean, (const float*)rstd, (const float*)nullptr, (const float*)nullptr, T, C, rpc, z);
  return OS2S_OK;
}

extern "C" int os2s_sample_norm_bwd(os2s_stream_t stream, const uint16_t* dz, const uint16_t* x, const float* gamma,
                                    const float* mean, const float* rstd, int B, int T, int C, int mode, uint16_t* dx,
                                    float* dgamma, float* dbeta, float* partial, float* scratch) {
  OS2S_REQUIRE(dz && x && gamma && mean && rstd && dx && dgamma && dbeta && partial && scratch);
  OS2S_REQUIRE(B >= 1 && B <= 65535 && T >= 1 && C >= 2 && C % 2 == 0 && (mode == 0 || mode == 1));
  const int rpc = (T + kSnChunks - 1) / kSnChunks;
  const dim3 grid((unsigned)((C / 2 + 255) / 256), kSnChunks, (unsigned)B);
  float* const sums = scratch;                               // [B][2][C]
  float* const m1 = scratch + (size_t)B * 2 * C;             // [B][C]
  float* const m2 = m1 + (size_t)B * C;                      // [B][C]
  OS2S_LAUNCH(sn_partial_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, dz, mean, rstd, T, C, rpc, partial);
  OS2S_LAUNCH(sn_bwd_finalize_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const float*)partial,
              gamma, T, C, mode, sums, m1, m2);
  OS2S_LAUNCH(sn_param_grad_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
              (const float*)sums, B, C, dgamma, dbeta);
  OS2S_LAUNCH(sn_apply_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, dz, gamma, (const float*)nullptr,
              mean, rstd, (const float*)m1, (const float*)m2, T, C, rpc, dx);
  return OS2S_OK;
}
