// Probe: issue rate of v_mfma_f32_32x32x16_bf16 written as inline assembly with the accumulators pinned to the
// accumulation registers (16 blocks = 256 registers, one wave per SIMD) — the stream of conv1d_wgrad_sw.hpp — against
// the compiler builtin on 4 blocks. Cycles per MFMA from the shader clock counter of one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[16];
  for (int a = 0; a < 16; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 y[4], x[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) {
    y[i][e] = (__bf16)(float)((threadIdx.x * 7 + i * 3 + e) % 5 - 2);
    x[i][e] = (__bf16)(float)((threadIdx.x * 3 + i * 5 + e) % 7 - 3);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int q = m >> 2, i = m & 3;
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[i], x[q], acc[i], 0, 0, 0);
      if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + q]) : "v"(y[i]), "v"(x[q]));
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(y[i]), "v"(x[q]));
      if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(y[i]), "v"(x[q]));
      if (MODE == 4) acc[i * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[i], x[q], acc[i * 4 + q], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 16; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(int blocks, const char* what) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * blocks); hipMalloc(&cyc, 8);
  const int iters = 4000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-64s %3d blk: %.2f cycles / MFMA\n", what, blocks, c / (16.0 * iters));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256}) {
    run<0>(blocks, "builtin, 4 accumulator blocks");
    run<4>(blocks, "builtin, 16 accumulator blocks");
    run<2>(blocks, "asm, 4 blocks in vector registers");
    run<3>(blocks, "asm, 4 blocks in accumulation registers");
    run<1>(blocks, "asm, 16 blocks in accumulation registers (the sw stream)");
  }
  return 0;
}
