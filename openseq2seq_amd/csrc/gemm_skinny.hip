// Skinny GEMM for the incremental decoding step, gfx950:
//   Y[M, N] = act(X[M, K] . W[N, K]^T + bias) (+ residual),  bf16 in/out, fp32 accumulate,
// with M = batch*beam rows (a few hundred). The dense layers of the Transformer decoder step
// (tf.layers.Dense calls under TransformerDecoder._get_symbols_to_logits_fn,
// open_seq2seq/decoders/transformer_decoder.py:232-285) have this shape: the weight matrix is
// read once (2-8 MB), the FLOPs are negligible, and what matters is the number of dependent
// memory round trips. A 128x128 LDS-tiled kernel runs 16 workgroups for [256 x 1024] and
// takes ~40 us; here every 32x32 output tile is one workgroup whose 8 waves split K, each
// wave issuing all loads of its K-share before its first MFMA (rnn_tile.hpp), so the whole
// product is ~one round trip + an LDS reduction. Bound: launch + L2 latency (≈5 us), then
// HBM for W.
#include <stdlib.h>

#include "os2s_common.hpp"
#include "rnn_tile.hpp"

namespace os2s {

constexpr int kSkWaves = 8;

struct SkinnyArgs {
  const bf16_t* x; long long ldx;     // [M, K]
  const bf16_t* w; long long ldw;     // [N, K]
  const float* bias;                  // [N] or null
  const bf16_t* res; long long ldr;   // [M, N] or null
  bf16_t* y; long long ldy;
  int M, N, K, relu;
};

__global__ __launch_bounds__(64 * kSkWaves) void gemm_skinny_kernel(SkinnyArgs p) {
  __shared__ float red[kSkWaves * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nrow = n0 + l31, mrow = m0 + l31;
  const bf16_t* wrow = nrow < p.N ? p.w + (long long)nrow * p.ldw : nullptr;
  const bf16_t* irow = mrow < p.M ? p.x + (long long)mrow * p.ldx : nullptr;
  tile_gemm_prefetch<kSkWaves, 8, true>(wrow, irow, p.K, acc, p.w);
  float out[4];
  tile_reduce_rows<kSkWaves>(acc, red, out);
  if (wave >= 4) return;
  const int m = m0 + l31, n = n0 + 8 * wave + 4 * lhi;
  if (m >= p.M || n >= p.N) return;
  // N % 4 == 0: the four columns n..n+3 are all valid
  if (p.bias) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] += b[e];
  }
  if (p.relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = fmaxf(out[e], 0.f);
  }
  if (p.res) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p.res + (long long)m * p.ldr + n);
    out[0] += bflo(r[0]); out[1] += bfhi(r[0]); out[2] += bflo(r[1]); out[3] += bfhi(r[1]);
  }
  u32x2 o;
  o[0] = pack2bf(out[0], out[1]);
  o[1] = pack2bf(out[2], out[3]);
  *reinterpret_cast<u32x2*>(p.y + (long long)m * p.ldy + n) = o;
}


// ---- LDS-staged variant ----------------------------------------------------------------------------
// The register-direct kernel above loads MFMA operands straight from global memory: one wave
// load touches 32 rows x 32 bytes, i.e. a quarter of 32 different 128-byte lines, and the 16 KB
// L1 cannot hold the lines until their other three quarters are used — the L2 ends up
// delivering ~4x the useful bytes (measured ≈7.5 TB/s useful at the L2 limit). Here the
// workgroup copies whole lines (two rows x 512 B per wave load) into LDS and the waves read
// their fragments from there.
// TN x TM 32x32 sub-tiles per workgroup, K in chunks of CHUNK staged through LDS (double
// buffered via registers); the 8 waves are (sub-tile, K-share of the chunk).
template <int TN, int TM, int CHUNK, int WN = 1, int WM = 1>
struct LdsCfg {
  static constexpr int kRows = 32 * (TN + TM);
  static constexpr int kPitch = CHUNK * 2 + 16;                 // +16 B: bank spread
  static constexpr int kStage = kRows * kPitch;
  static constexpr int kPieces = kRows * (CHUNK / 8) / 512;     // 16-byte pieces per thread
  static constexpr int kColsPerRow = CHUNK / 8;                 // pieces per row
  static constexpr int kRowsPerPass = 512 / kColsPerRow;
  static constexpr int kGroups = (TN / WN) * (TM / WM);         // wave groups = output sub-blocks
  static constexpr int kSplit = 8 / kGroups;                    // waves sharing one sub-block
  static constexpr int kSlices = CHUNK / 16 / kSplit;           // MFMA k-slices per wave per chunk
  static constexpr int kRed = (kSplit - 1) * TN * TM * 4096;
  static constexpr int kSmem = 2 * kStage > kRed ? 2 * kStage : kRed;
};

template <int TN, int TM, int CHUNK, int WN = 1, int WM = 1>
__global__ __launch_bounds__(512) void gemm_skinny_lds_kernel(SkinnyArgs p) {
  using C = LdsCfg<TN, TM, CHUNK, WN, WM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32 * TN, m0 = blockIdx.y * 32 * TM;
  const int sub = wave % C::kGroups, ks = wave / C::kGroups;
  const int tn = (sub % (TN / WN)) * WN, tm = (sub / (TN / WN)) * WM;   // first sub-tile of this wave
  // ---- global -> register staging: piece j = row (j*kRowsPerPass + tid/kColsPerRow), 16-B column
  const int prow = tid / C::kColsPerRow, pcol = tid % C::kColsPerRow;
  const bf16_t* src[C::kPieces];
  bool ok[C::kPieces];
#pragma unroll
  for (int j = 0; j < C::kPieces; ++j) {
    const int r = j * C::kRowsPerPass + prow;
    if (r < 32 * TN) { ok[j] = n0 + r < p.N; src[j] = p.w + (long long)min(n0 + r, p.N - 1) * p.ldw; }
    else { ok[j] = m0 + r - 32 * TN < p.M; src[j] = p.x + (long long)min(m0 + r - 32 * TN, p.M - 1) * p.ldx; }
  }
  const int nchunks = (p.K + CHUNK - 1) / CHUNK;
  u32x4 st[C::kPieces];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int j = 0; j < C::kPieces; ++j) {
      const int k = c * CHUNK + pcol * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(src[j] + min(k, p.K - 8));
      const u32x4 z = {0u, 0u, 0u, 0u};
      st[j] = (ok[j] && k < p.K) ? v : z;
    }
  };
  auto store_chunk = [&](int stage) {
#pragma unroll
    for (int j = 0; j < C::kPieces; ++j)
      *reinterpret_cast<u32x4*>(smem + stage * C::kStage + (j * C::kRowsPerPass + prow) * C::kPitch + pcol * 16) = st[j];
  };
  f32x16 acc[WN][WM];
#pragma unroll
  for (int i = 0; i < WN; ++i)
#pragma unroll
    for (int j = 0; j < WM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_chunk(c + 1);
    const char* base = smem + (c & 1) * C::kStage;
    const char* ap = base + (tn * 32 + l31) * C::kPitch + ks * C::kSlices * 32 + lhi * 16;
    const char* bp = base + (32 * TN + tm * 32 + l31) * C::kPitch + ks * C::kSlices * 32 + lhi * 16;
#pragma unroll
    for (int s2 = 0; s2 < C::kSlices; ++s2) {
      bf16x8 a[WN], b[WM];
#pragma unroll
      for (int i = 0; i < WN; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * C::kPitch + s2 * 32);
#pragma unroll
      for (int j = 0; j < WM; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bp + j * 32 * C::kPitch + s2 * 32);
#pragma unroll
      for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (c + 1 < nchunks) store_chunk((c + 1) & 1);
    __syncthreads();
  }
  // ---- combine the K-shares, epilogue ------------------------------------------------------------------
  if constexpr (C::kSplit > 1) {
    static_assert(WN == 1 && WM == 1, "K-split variants hold one sub-tile per wave");
    float* red = reinterpret_cast<float*>(smem);      // [kSplit-1][groups][16][64]
    if (ks > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((ks - 1) * C::kGroups + sub) * 16 + r) * 64 + lane] = acc[0][0][r];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int q = 0; q < C::kSplit - 1; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += red[((q * C::kGroups + sub) * 16 + r) * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    const int m = m0 + (tm + j) * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (tn + i) * 32 + 8 * g + 4 * lhi;
        if (n >= p.N) continue;
        float out[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (p.bias) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] += bb[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] = fmaxf(out[e], 0.f);
        }
        if (p.res) {
          const u32x2 r = *reinterpret_cast<const u32x2*>(p.res + (long long)m * p.ldr + n);
          out[0] += bflo(r[0]); out[1] += bfhi(r[0]); out[2] += bflo(r[1]); out[3] += bfhi(r[1]);
        }
        u32x2 o;
        o[0] = pack2bf(out[0], out[1]);
        o[1] = pack2bf(out[2], out[3]);
        *reinterpret_cast<u32x2*>(p.y + (long long)m * p.ldy + n) = o;
      }
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_gemm_skinny(os2s_stream_t stream, const uint16_t* x, long long ldx,
                                const uint16_t* w, long long ldw, const float* bias,
                                const uint16_t* residual, long long ldr, int M, int N, int K,
                                int relu, uint16_t* y, long long ldy) {
  OS2S_REQUIRE(x && w && y && M >= 1 && N >= 4 && N % 4 == 0 && K >= 8 && K % 8 == 0);
  OS2S_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0));
  SkinnyArgs a;
  a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.bias = bias; a.res = residual; a.ldr = ldr;
  a.y = y; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.relu = relu;
  // variant: 128x256 tiles when even those fill the chip (the vocabulary projection), 64x64
  // tiles when they do, else 32x32 tiles (4x the workgroups, each streaming half the bytes);
  // "reg" = the register-direct kernel. OS2S_SKINNY_VARIANT ("reg" | "l64" | "l32" | "wide")
  // overrides (tools / tests).
  static const char* const force = getenv("OS2S_SKINNY_VARIANT");     // read once per process
  const long long blocks64 = (long long)((N + 63) / 64) * ((M + 63) / 64);
  int variant = blocks64 >= 192 ? 2 : 1;
  if (M > 128 && (long long)((N + 127) / 128) * ((M + 255) / 256) >= 192) variant = 3;
  if (force) variant = force[0] == 'r' ? 0 : (force[0] == 'w' ? 3 : (force[1] == '6' ? 2 : 1));
  if (variant == 3) {     // wide: 128 (n) x 256 (m) per workgroup, no K-split — least L2 traffic
    using C = LdsCfg<4, 8, 64, 4, 1>;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm_skinny_lds_kernel<4, 8, 64, 4, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::kSmem) != hipSuccess)
        return OS2S_ERR_LAUNCH;
      attr_set = true;
    }
    OS2S_LAUNCH((gemm_skinny_lds_kernel<4, 8, 64, 4, 1>), dim3((N + 127) / 128, (M + 255) / 256),
                dim3(512), C::kSmem, (hipStream_t)stream, a);
    return OS2S_OK;
  }
  if (variant == 2) {
    using C = LdsCfg<2, 2, 256>;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm_skinny_lds_kernel<2, 2, 256>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::kSmem) != hipSuccess)
        return OS2S_ERR_LAUNCH;
      attr_set = true;
    }
    OS2S_LAUNCH((gemm_skinny_lds_kernel<2, 2, 256>), dim3((N + 63) / 64, (M + 63) / 64), dim3(512),
                C::kSmem, (hipStream_t)stream, a);
    return OS2S_OK;
  }
  if (variant == 1) {
    using C = LdsCfg<1, 1, 512>;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm_skinny_lds_kernel<1, 1, 512>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::kSmem) != hipSuccess)
        return OS2S_ERR_LAUNCH;
      attr_set = true;
    }
    OS2S_LAUNCH((gemm_skinny_lds_kernel<1, 1, 512>), dim3((N + 31) / 32, (M + 31) / 32), dim3(512),
                C::kSmem, (hipStream_t)stream, a);
    return OS2S_OK;
  }
  OS2S_LAUNCH(gemm_skinny_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(64 * kSkWaves), 0,
              (hipStream_t)stream, a);
  return OS2S_OK;
}
