// Instruction-fetch cost of straight-line code at dispatch (round 4, DESIGN section 3d: "a step kernel's latency is its
// code size"). Each kernel executes N VALU instructions (v_add_f32, 4 bytes each) per wave, either as ONE
// straight-line block of N instructions or as a 64-instruction body looped N / 64 times (same work, 256 bytes of
// code). One 64-thread workgroup per CU (256 workgroups), launched back to back on one stream; time per launch =
// total / launches. The difference between the two forms is what the code bytes cost.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_icache.hip -o tools/probe_icache && ./tools/probe_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REPT_ADD(n) asm volatile(".rept " #n "\n v_add_f32 %0, %0, %0\n .endr" : "+v"(x))

template <int KB>
__global__ void straight(float* out) {
  float x = (float)threadIdx.x;
  if constexpr (KB == 1) REPT_ADD(256);
  if constexpr (KB == 4) REPT_ADD(1024);
  if constexpr (KB == 16) REPT_ADD(4096);
  if constexpr (KB == 32) REPT_ADD(8192);
  if constexpr (KB == 64) REPT_ADD(16384);
  if (x == 123.456f) out[0] = x;
}

__global__ void looped(float* out, int iters) {
  float x = (float)threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) REPT_ADD(64);
  if (x == 123.456f) out[0] = x;
}

__global__ void empty(float* out) {
  if (threadIdx.x == 9999) out[0] = 1.f;
}

template <typename F>
static double time_us(F launch, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return 1000.0 * ms / n;
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  const int n = 400;
  for (int wg : {256, 1024}) {
    for (int threads : {64, 512}) {
      const dim3 g(wg), b(threads);
      printf("grid %d x %d threads\n", wg, threads);
      printf("  empty kernel                      %7.2f us / launch\n", time_us([&] { hipLaunchKernelGGL(empty, g, b, 0, 0, out); }, n));
#define ROW(KB) { const double s = time_us([&] { hipLaunchKernelGGL(straight<KB>, g, b, 0, 0, out); }, n); \
                  const double l = time_us([&] { hipLaunchKernelGGL(looped, g, b, 0, 0, out, KB * 4); }, n); \
                  printf("  %2d KB straight-line %7.2f us   same work looped over 256 B %7.2f us   fetch cost %6.2f us = %5.1f ns per 64 B\n", \
                         KB, s, l, s - l, 1000.0 * (s - l) / (KB * 16.0)); }
      ROW(1) ROW(4) ROW(16) ROW(32) ROW(64)
      // the same kernels alternating with a 64 KB "evictor" kernel (different code between two launches of a kernel, as in
      // a step loop of several kernels): pair time minus the evictor's own time
      const double ev = time_us([&] { hipLaunchKernelGGL(straight<64>, g, b, 0, 0, out); }, n);
#define PAIR(KB) { const double s = time_us([&] { hipLaunchKernelGGL(straight<KB>, g, b, 0, 0, out); hipLaunchKernelGGL(straight<64>, g, b, 0, 0, out); }, n) - ev; \
                   const double l = time_us([&] { hipLaunchKernelGGL(looped, g, b, 0, 0, out, KB * 4); hipLaunchKernelGGL(straight<64>, g, b, 0, 0, out); }, n) - ev; \
                   printf("  alternating: %2d KB straight-line %7.2f us   looped %7.2f us   fetch cost %6.2f us = %5.1f ns per 64 B\n", \
                          KB, s, l, s - l, 1000.0 * (s - l) / (KB * 16.0)); }
      PAIR(1) PAIR(4) PAIR(16) PAIR(32)
    }
  }
  return 0;
}
