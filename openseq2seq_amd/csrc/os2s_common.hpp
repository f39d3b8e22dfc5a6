// Shared device/host helpers for the os2s HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/os2s.h"

// hipGetLastError() is sticky per thread and also reports benign errors left
// behind by other HIP users in the process (e.g. PyTorch's event queries), so
// every launch is bracketed: clear first, then check.
#define OS2S_LAUNCH(kernel, grid, block, smem, stream, ...)              \
  do {                                                                   \
    (void)hipGetLastError();                                             \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);  \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      os2s_record_hip_error((int)e__, #kernel);                          \
      return OS2S_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

extern "C" void os2s_record_hip_error(int hip_error, const char* where);

#define OS2S_REQUIRE(cond) \
  do {                     \
    if (!(cond)) return OS2S_ERR_INVALID_ARG; \
  } while (0)

namespace os2s {

// Named test / measurement options behind the ONE entry point os2s_set_option (os2s_api.hip). A translation
// unit registers its knobs next to the state they set: `static OptionReg r("conv1d.variant", [](double v) {...});`
// (host-side only; the registry lives behind a function-local static so the order in which the translation
// units are initialised does not matter). Nothing registered here is read from the environment.
typedef void (*OptionSetter)(double value);
struct OptionReg { OptionReg(const char* name, OptionSetter fn); };
// debug time-stamp buffers of instrumented kernels (os2s_set_debug_stamps): fn(stamps, mode)
typedef void (*StampSetter)(void* stamps, int mode);
struct StampReg { StampReg(const char* name, StampSetter fn); };

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_t;  // raw storage type used across the C ABI

// round-to-nearest-even fp32 -> bf16 (NaN stays a quiet NaN): the gfx950 conversion instruction
// v_cvt_pk_bf16_f32, two values per instruction. (The integer formulation — add 0x7fff + lsb, NaN
// test per value — is ~12 VALU instructions and two exec-mask branches per value; it was 2000 of
// the instructions of a 256 x 256 convolution epilogue.)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
__device__ __forceinline__ float bflo(uint32_t p) {
  return __builtin_bit_cast(float, p << 16);
}
__device__ __forceinline__ float bfhi(uint32_t p) {
  return __builtin_bit_cast(float, p & 0xffff0000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// DPP-based wave64 reductions for FULLY ACTIVE waves (checked against the shuffle versions by
// tools/probe_dpp.hip): six VALU DPP moves instead of six ds_bpermute round trips through
// the LDS crossbar (~10x lower latency — it matters in the sequential decoder kernels). The
// result is wave-uniform (read from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL,
                                                    ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x += dpp_mov<0xB1, 0xf>(0.f, x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E, 0xf>(0.f, x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x124, 0xf>(0.f, x);   // row_ror:4
  x += dpp_mov<0x128, 0xf>(0.f, x);   // row_ror:8
  x += dpp_mov<0x142, 0xa>(0.f, x);   // row_bcast:15 into rows 1, 3
  x += dpp_mov<0x143, 0xc>(0.f, x);   // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max_dpp(float x) {
  const float ninf = -__builtin_inff();
  x = fmaxf(x, dpp_mov<0xB1, 0xf>(ninf, x));
  x = fmaxf(x, dpp_mov<0x4E, 0xf>(ninf, x));
  x = fmaxf(x, dpp_mov<0x124, 0xf>(ninf, x));
  x = fmaxf(x, dpp_mov<0x128, 0xf>(ninf, x));
  x = fmaxf(x, dpp_mov<0x142, 0xa>(ninf, x));
  x = fmaxf(x, dpp_mov<0x143, 0xc>(ninf, x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// Counter-based RNG (philox-like mixing, cheap): deterministic per (seed, idx).
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// 256 B of zeros: LDS-DMA source for padding rows / masked frames (one copy per
// translation unit; device code is not linked across TUs).
static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

// 4 keep/drop decisions (elements idx4*4 .. idx4*4 + 3) from the mixing state
// z0 = idx4 * kDropGolden + seed: one 64-bit mix -> four 16-bit uniforms, keep iff u16 < thr
// (thr = keep * 65536). Callers that walk idx4 in constant steps add step * kDropGolden to z0
// instead of redoing the first multiply.
constexpr unsigned long long kDropGolden = 0x9E3779B97F4A7C15ull;
__device__ __forceinline__ uint32_t dropout_bits4_z(unsigned long long z, uint32_t thr) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  uint32_t bits = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t u = (uint32_t)(z >> (16 * e)) & 0xffffu;
    bits |= (u < thr ? 1u : 0u) << e;
  }
  return bits;
}
// 8 keep/drop decisions for the 8 channels starting at element index idx8*8:
// two 64-bit mixes (idx4 = idx8*2, idx8*2 + 1) -> eight 16-bit uniforms, keep iff u16 < keep*65536.
__device__ __forceinline__ uint32_t dropout_bits8(unsigned long long seed,
                                                  unsigned long long idx8,
                                                  float keep_prob) {
  const uint32_t thr = (uint32_t)(keep_prob * 65536.0f);
  const unsigned long long z0 = (idx8 * 2) * kDropGolden + seed;
  return dropout_bits4_z(z0, thr) | (dropout_bits4_z(z0 + kDropGolden, thr) << 4);
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace os2s
