// Probe: issue rate of v_mfma_f32_32x32x16_bf16 in the access patterns of the ping-pong kernels.
//   mode 0: 16 MFMAs per slot over 4 accumulators, (kk, in, i2) order of conv1d_pp_kernel
//   mode 1: the same over 8 accumulators (two items) — every accumulator used every 8th MFMA
//   mode 2: 4 accumulators, accumulator-major order (4 dependent MFMAs in a row)
// 1 or 2 waves per SIMD (blockDim 256 / 512), no barriers, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = (float)(threadIdx.x + a);
  bf16x8 wf[2][4], xf[4][4];
  for (int i = 0; i < 2; ++i) for (int kk = 0; kk < 4; ++kk) for (int e = 0; e < 8; ++e) wf[i][kk][e] = (__bf16)(float)(threadIdx.x % 7 + i + kk);
  for (int i = 0; i < 4; ++i) for (int kk = 0; kk < 4; ++kk) for (int e = 0; e < 8; ++e) xf[i][kk][e] = (__bf16)(float)(threadIdx.x % 5 + i - kk);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            acc[in * 2 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk], acc[in * 2 + i2], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2)
            acc[in * 4 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk], acc[in * 4 + i2], 0, 0, 0);
    } else {
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[in * 2 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk], acc[in * 2 + i2], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(int threads, int blocks, const char* what) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = 16.0 * iters;                       // MFMAs per wave
  const double waves_per_simd = threads / 256.0;
  printf("%-44s %d thr x %d blk: %.1f counter ticks / MFMA / wave (x%.0f waves per SIMD = %.1f per SIMD-MFMA), %.3f ms, %.0f TF/s\n",
         what, threads, blocks, c / mf, waves_per_simd, c / mf / waves_per_simd, ms,
         2.0 * 32 * 32 * 16 * mf * (threads / 64) * blocks / ms / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256}) {
    run<0>(256, blocks, "mode 0 (4 acc, kernel order), 1 wave/SIMD");
    run<0>(512, blocks, "mode 0 (4 acc, kernel order), 2 waves/SIMD");
    run<1>(256, blocks, "mode 1 (8 acc), 1 wave/SIMD");
    run<1>(512, blocks, "mode 1 (8 acc), 2 waves/SIMD");
    run<2>(256, blocks, "mode 2 (4 acc, dependent runs), 1 wave/SIMD");
  }
  return 0;
}
