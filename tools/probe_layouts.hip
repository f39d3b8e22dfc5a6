// Probe: verifies the MFMA 32x32x16 bf16 fragment layouts and ds_read_tr16_b64
// semantics assumed by the conv/GEMM kernels. Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void mfma_probe(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 31) * 16 + (l >> 5) * 8 + j];   // A[m=l&31][k=(l>>5)*8+j]
    b[j] = (__bf16)B[((l >> 5) * 8 + j) * 32 + (l & 31)]; // B[k][n=l&31]
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    D[row * 32 + col] = acc[r];
  }
}

// tr16_b64: LDS holds S[k][n] row-major bf16, 16 k-rows x 32 n-cols (row stride 64 B).
// We want B-fragment: lane l gets B[k=(l>>5)*8+j][n=l&31], j=0..7 via two tr reads.
__global__ void tr_probe(const float* S /*16x32*/, float* out /*64x8*/) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[16 * 32];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 32; i += 64) lds[i] = (__bf16)S[i];
  __syncthreads();
  // 16-lane group g = l>>4 : n block = (g&1)*16, k block = (g>>1)*8 (+4 for 2nd read)
  // within group, lane i supplies address of row (i>>2), cols (i&3)*4 .. +3
  int g = l >> 4, i = l & 15;
  int nb = (g & 1) * 16, kb = (g >> 1) * 8;
  const __bf16* p0 = &lds[(kb + (i >> 2)) * 32 + nb + (i & 3) * 4];
  const __bf16* p1 = &lds[(kb + 4 + (i >> 2)) * 32 + nb + (i & 3) * 4];
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
  for (int j = 0; j < 4; ++j) { out[l * 8 + j] = (float)r0[j]; out[l * 8 + 4 + j] = (float)r1[j]; }
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32 * 16; ++i) A[i] = (float)((i * 7 + 3) % 13 - 6);
  for (int i = 0; i < 16 * 32; ++i) B[i] = (float)((i * 5 + 1) % 11 - 5);
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n]; R[m * 32 + n] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 32 * 32; ++i) if (D[i] != R[i]) ++bad;
  printf("mfma_32x32x16 layout: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);

  std::vector<float> S(16 * 32), O(64 * 8);
  for (int i = 0; i < 16 * 32; ++i) S[i] = (float)i;  // value = k*32+n, exact in bf16? up to 511: bf16 has 8 bits mantissa -> not exact
  for (int i = 0; i < 16 * 32; ++i) S[i] = (float)((i / 32) * 32 + (i % 32)) ;
  // use small exact values: encode k*32+n <= 511 -> not exact in bf16 (8 bit mantissa => exact up to 256). Use k in high, n low separately:
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) S[k * 32 + n] = (float)(k * 8) + (float)n / 4.0f; // k*8 + n/4: max 127.75, needs 9 bits.. use two probes
  float *dS, *dO; hipMalloc(&dS, S.size() * 4); hipMalloc(&dO, O.size() * 4);
  int badk = 0, badn = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) S[k * 32 + n] = pass == 0 ? (float)k : (float)n;
    hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dS, dO);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      float want = pass == 0 ? (float)((l >> 5) * 8 + j) : (float)(l & 31);
      if (O[l * 8 + j] != want) { if (pass == 0) ++badk; else ++badn; }
    }
    if (pass == 0 && badk) { printf("tr k-map sample lane0: "); for (int j = 0; j < 8; ++j) printf("%g ", O[j]); printf(" lane17: "); for (int j = 0; j < 8; ++j) printf("%g ", O[17 * 8 + j]); printf(" lane40: "); for (int j = 0; j < 8; ++j) printf("%g ", O[40 * 8 + j]); printf("\n"); }
    if (pass == 1 && badn) { printf("tr n-map sample lane0: "); for (int j = 0; j < 8; ++j) printf("%g ", O[j]); printf(" lane17: "); for (int j = 0; j < 8; ++j) printf("%g ", O[17 * 8 + j]); printf(" lane40: "); for (int j = 0; j < 8; ++j) printf("%g ", O[40 * 8 + j]); printf("\n"); }
  }
  printf("ds_read_tr16_b64 B-fragment mapping: %s (badk=%d badn=%d)\n", (badk || badn) ? "FAIL" : "PASS", badk, badn);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s CUs=%d clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  return 0;
}
