// Probe: what "peak" is on this box. The guide's 2 495 TFLOP/s loop (back-to-back independent
// v_mfma_f32_32x32x16_bf16, one wave per SIMD on every CU) with three operand fills — zeros, small integers,
// full-range random — run long enough (tens of ms) for the power manager to settle, reporting TFLOP/s AND the
// shader clock (cycle counter over the 100 MHz reference counter). MI355X_MICROARCH.md "DVFS give-back": the same
// instruction stream clocks 2.3 GHz on zero operands and 1.9 - 1.95 GHz on random ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline float fill_value(int fill, unsigned idx) {
  if (fill == 0) return 0.f;
  if (fill == 1) return (float)(int)(idx % 7) - 3.f;
  unsigned h = idx * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return ((float)(h & 0xffffff) / 8388608.f) - 1.f;             // uniform [-1, 1)
}

__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, int fill) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 wf[2][4], xf[2][4];
  for (int i = 0; i < 2; ++i) for (int kk = 0; kk < 4; ++kk) for (int e = 0; e < 8; ++e) {
    wf[i][kk][e] = (__bf16)fill_value(fill, threadIdx.x * 131u + i * 37u + kk * 11u + e);
    xf[i][kk][e] = (__bf16)fill_value(fill, threadIdx.x * 71u + i * 53u + kk * 17u + e + 99991u);
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          acc[in * 2 + i2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[in][kk], xf[i2][kk], acc[in * 2 + i2], 0, 0, 0);
    asm volatile("" ::: "memory");
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 17) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;          // x 16 MFMAs x 32 cycles = 102 M cycles = ~45 ms
  float* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * 256 * 256); hipMalloc(&clk, 16);
  const char* names[3] = {"zeros", "small integers", "uniform random [-1, 1)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int fill = 0; fill < 3; ++fill) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, clk, iters, fill);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double mf = 16.0 * iters;
      printf("fill %-24s 256 thr x 256 blk: %.2f cycles / MFMA / SIMD, clock %.0f MHz, %.2f ms, %.0f TFLOP/s"
             " (= %.3f of 2500; at 2400 MHz it would be %.0f)\n", names[fill], c[0] / mf, 100.0 * c[0] / c[1], ms,
             2.0 * 32 * 32 * 16 * mf * 4 * 256 / ms / 1e9, 2.0 * 32 * 32 * 16 * mf * 4 * 256 / ms / 1e9 / 2500.0,
             2.0 * 32 * 32 * 16 * mf * 4 * 256 / ms / 1e9 * 2400.0 / (100.0 * c[0] / c[1]));
    }
  return 0;
}
