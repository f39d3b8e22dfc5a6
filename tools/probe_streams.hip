// How fast can a multi-stream read-modify-write pass go? (round 5: the optimizer's apply kernel moves 22 B per
// parameter — grad, master, moment read; master, moment, bf16 copy written — at 4.0-4.4 TB/s, the guide's float4 copy
// reaches 6.3.) Variants over N floats per array, one 4096-element chunk per 256-thread workgroup unless noted:
//   copy       1 read + 1 write stream                              (8 B / element)
//   opt        3 read + 3 write streams, the optimizer's pattern    (22 B / element)
//   opt_nt     the same with nontemporal loads of the gradient and nontemporal stores of the bf16 copy
//   opt_grid   the same pattern, 2048 workgroups walking the chunks with a grid stride
//   opt_2x     two chunks per workgroup (24 float4 loads in flight per thread)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_streams.hip -o tools/probe_streams && ./tools/probe_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}

constexpr int kChunk = 4096;

__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ a, float* __restrict__ b) {
  const long long base = (long long)blockIdx.x * kChunk + threadIdx.x * 4;
  f32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(a + base + i * 1024);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(b + base + i * 1024) = v[i];
}

template <int MODE, int NCH>      // MODE 0 plain, 1 nontemporal g load / w16 store, 2 grid stride
__global__ __launch_bounds__(256) void opt_kernel(const float* __restrict__ g, float* __restrict__ w,
                                                  float* __restrict__ m, uint16_t* __restrict__ w16, long long nchunks) {
  for (long long c = (long long)blockIdx.x * NCH; c < nchunks; c += (MODE == 2 ? (long long)gridDim.x * NCH : nchunks)) {
    const long long base = c * kChunk + threadIdx.x * 4;
    f32x4 gv[4 * NCH], wv[4 * NCH], mv[4 * NCH];
#pragma unroll
    for (int i = 0; i < 4 * NCH; ++i) {
      const long long off = base + (long long)i * 1024;
      if (MODE == 1) gv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + off));
      else gv[i] = *reinterpret_cast<const f32x4*>(g + off);
      wv[i] = *reinterpret_cast<const f32x4*>(w + off);
      mv[i] = *reinterpret_cast<const f32x4*>(m + off);
    }
#pragma unroll
    for (int i = 0; i < 4 * NCH; ++i) {
      const long long off = base + (long long)i * 1024;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mv[i][e] = 0.95f * mv[i][e] + gv[i][e] * 0.001f;
        wv[i][e] -= 0.01f * mv[i][e];
      }
      *reinterpret_cast<f32x4*>(w + off) = wv[i];
      *reinterpret_cast<f32x4*>(m + off) = mv[i];
      u32x2 o;
      o[0] = pack2bf(wv[i][0], wv[i][1]);
      o[1] = pack2bf(wv[i][2], wv[i][3]);
      if (MODE == 1) __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(w16 + off));
      else *reinterpret_cast<u32x2*>(w16 + off) = o;
    }
  }
}

template <typename F>
static double time_ms(F launch, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / n;
}

int main() {
  const long long nchunks = 81200;                     // 332.6 M parameters, the Jasper 10x5 flat buffers
  const long long N = nchunks * kChunk;
  float *g, *w, *m;
  uint16_t* w16;
  if (hipMalloc(&g, N * 4) != hipSuccess || hipMalloc(&w, N * 4) != hipSuccess || hipMalloc(&m, N * 4) != hipSuccess ||
      hipMalloc(&w16, N * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(g, 0, N * 4); hipMemset(w, 0, N * 4); hipMemset(m, 0, N * 4); hipMemset(w16, 0, N * 2);
  const double gb_copy = 8.0 * N / 1e9, gb_opt = 22.0 * N / 1e9;
  double t;
  t = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(nchunks), dim3(256), 0, 0, g, w); }, 10);
  printf("copy      %.3f ms  %.2f TB/s\n", t, gb_copy / t);
  t = time_ms([&] { hipLaunchKernelGGL((opt_kernel<0, 1>), dim3(nchunks), dim3(256), 0, 0, g, w, m, w16, nchunks); }, 10);
  printf("opt       %.3f ms  %.2f TB/s\n", t, gb_opt / t);
  t = time_ms([&] { hipLaunchKernelGGL((opt_kernel<1, 1>), dim3(nchunks), dim3(256), 0, 0, g, w, m, w16, nchunks); }, 10);
  printf("opt_nt    %.3f ms  %.2f TB/s\n", t, gb_opt / t);
  for (int grid : {1024, 2048, 4096, 8192}) {
    t = time_ms([&] { hipLaunchKernelGGL((opt_kernel<2, 1>), dim3(grid), dim3(256), 0, 0, g, w, m, w16, nchunks); }, 10);
    printf("opt_grid %5d  %.3f ms  %.2f TB/s\n", grid, t, gb_opt / t);
  }
  t = time_ms([&] { hipLaunchKernelGGL((opt_kernel<0, 2>), dim3(nchunks / 2), dim3(256), 0, 0, g, w, m, w16, nchunks); }, 10);
  printf("opt_2x    %.3f ms  %.2f TB/s\n", t, gb_opt / t);
  return 0;
}
