// Small helpers of the convolution layers that are data movement, not arithmetic.
#include "os2s_common.hpp"

namespace os2s {

// y[b, t * stride, :] = x[b, t, :], every other row of y zero: the zero-upsampled output gradient of a strided
// convolution — its data gradient is then the stride-1 data gradient of the upsampled tensor
// (dx[t] = sum_k dy_up[t - k dil + padL] w[k] with dy_up[t' stride] = dy[t']).
__global__ __launch_bounds__(256) void upsample_rows_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y,
                                                            int T, int Tup, int c8, int stride, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // over B * Tup * c8 16-byte pieces of y
  if (i >= total) return;
  const int c = (int)(i % c8);
  const long long r = i / c8;
  const int tu = (int)(r % Tup);
  const long long b = r / Tup;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (tu % stride == 0 && tu / stride < T) v = x[(b * T + tu / stride) * c8 + c];
  y[i] = v;
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_upsample_rows_bf16(os2s_stream_t stream, const uint16_t* x, int B, int T, int C, int stride,
                                       int Tup, uint16_t* y) {
  OS2S_REQUIRE(x && y && B >= 0 && T >= 1 && C >= 8 && C % 8 == 0 && stride >= 1 && Tup >= (T - 1) * stride + 1);
  if (B == 0) return OS2S_OK;
  const long long total = (long long)B * Tup * (C / 8);
  OS2S_LAUNCH(upsample_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
              reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), T, Tup, C / 8, stride, total);
  return OS2S_OK;
}
