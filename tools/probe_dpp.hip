// Probe: DPP-based wave64 sum / max vs shuffle reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ float dpp_f(float old, float x, int ctrl_dummy);
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppmov(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x += dppmov<0xB1, 0xf>(0.f, x);   // quad_perm [1,0,3,2]
  x += dppmov<0x4E, 0xf>(0.f, x);   // quad_perm [2,3,0,1]
  x += dppmov<0x124, 0xf>(0.f, x);  // row_ror:4
  x += dppmov<0x128, 0xf>(0.f, x);  // row_ror:8
  x += dppmov<0x142, 0xa>(0.f, x);  // row_bcast:15 -> rows 1,3
  x += dppmov<0x143, 0xc>(0.f, x);  // row_bcast:31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max_dpp(float x) {
  const float ninf = -INFINITY;
  x = fmaxf(x, dppmov<0xB1, 0xf>(ninf, x));
  x = fmaxf(x, dppmov<0x4E, 0xf>(ninf, x));
  x = fmaxf(x, dppmov<0x124, 0xf>(ninf, x));
  x = fmaxf(x, dppmov<0x128, 0xf>(ninf, x));
  x = fmaxf(x, dppmov<0x142, 0xa>(ninf, x));
  x = fmaxf(x, dppmov<0x143, 0xc>(ninf, x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  float s = wave_sum_dpp(v), m = wave_max_dpp(v);
  float rs = v, rm = v;
  for (int o = 32; o > 0; o >>= 1) { rs += __shfl_xor(rs, o, 64); rm = fmaxf(rm, __shfl_xor(rm, o, 64)); }
  out[threadIdx.x * 4 + 0] = s; out[threadIdx.x * 4 + 1] = rs;
  out[threadIdx.x * 4 + 2] = m; out[threadIdx.x * 4 + 3] = rm;
}
int main() {
  float h[64], o[256]; float *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 101) - 50.f + 0.25f * i;
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (fabsf(o[i*4] - o[i*4+1]) > 1e-3f || o[i*4+2] != o[i*4+3]) ++bad;
  printf("dpp sum %f ref %f max %f ref %f bad lanes %d\n", o[0], o[1], o[2], o[3], bad);
  return bad != 0;
}
