// CU-thief: N workgroups that sit on N compute units for a given time (tools/cu_thief.py). A stand-in for the
// RCCL all-reduce kernels of an 8-GPU run on a one-GPU box: an RCCL channel is a resident workgroup with a small
// LDS footprint; next to it a 160 KB-LDS ping-pong convolution tile does not fit on that CU, so for the duration
// of a bucket the convolution launches see 256 - N compute units. (What it does not model: the HBM / fabric
// traffic of the collective.)
#include <hip/hip_runtime.h>
#include <stdint.h>

// 96 KB of LDS per workgroup: more than half a CU's 160 KB, so the dispatcher cannot put two thieves on one CU
// (N workgroups = N compute units; a 16 KB footprint let it stack them). where[blockIdx.x] = XCC id << 16 | HW_ID
// (se / sh / cu fields) so the tool can count the distinct CUs that were actually held.
__global__ __launch_bounds__(256) void cu_thief_kernel(long long ticks_100mhz, int* sink, int* where) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && where) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    where[blockIdx.x] = (int)(((xcc & 0xf) << 16) | ((hw >> 8) & 0xff));   // cu_id[11:8], sh_id[12], se_id[15:13]
  }
  const long long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < ticks_100mhz) {
    acc += lds[(threadIdx.x + acc) & 1023];
    __builtin_amdgcn_s_sleep(16);
  }
  if (acc == 0x7fffffff) *sink = acc;
}

extern "C" int cu_thief_launch(void* stream, int workgroups, double microseconds, int* sink, int* where) {
  if (workgroups <= 0) return 0;
  static bool once = [] {
    return hipFuncSetAttribute((const void*)cu_thief_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  }();
  (void)once;
  hipLaunchKernelGGL(cu_thief_kernel, dim3(workgroups), dim3(256), 96 * 1024, (hipStream_t)stream,
                     (long long)(microseconds * 100.0), sink, where);
  return (int)hipGetLastError();
}
