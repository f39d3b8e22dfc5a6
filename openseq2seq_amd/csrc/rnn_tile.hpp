// Small-batch "recurrent GEMM" tile shared by the RNN and attention-decoder step kernels.
//
// A recurrent step multiplies a [B, K] state (B <= a few hundred rows) with a [N, K] weight
// matrix. It is latency-, not throughput-bound: one 32x32 output tile per workgroup, the
// reduction dimension split over the NW waves of the workgroup (each wave issues all loads
// of UNR k-slices before its MFMAs, so the L2 latency is paid once per UNR slices instead
// of once per slice), partial tiles summed through LDS. The orientation is swapped
// (rows = output units, columns = batch) so that all G gates of one (unit, sample) pair end
// up in the same lane and the cell non-linearity is lane-local.
#pragma once
#include "os2s_common.hpp"

namespace os2s {

// acc[g] += W[g*gate_stride + j, 0:K] . In[b, 0:K]  for the 32 rows j = j0.. and 32 columns
// b = b0.. of this workgroup; this wave covers k-slices wave, wave+NW, ...
// MFMA 32x32x16 operand layout: lane supplies 8 consecutive k at (lane>>5)*8 for row lane&31.
template <int G, int NW, int UNR = 4>
__device__ __forceinline__ void tile_gemm_splitk(const bf16_t* __restrict__ w, long long ldw,
                                                 long long gate_stride, int j0, int n_rows,
                                                 const bf16_t* __restrict__ in, long long ldin,
                                                 int b0, int n_batch, int K, f32x16 (&acc)[G]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int jrow = min(j0 + l31, n_rows - 1);
  const int brow = b0 + l31;
  const bool bvalid = brow < n_batch;
  const bf16_t* ip = in + (long long)min(brow, n_batch - 1) * ldin + lhi * 8;
  const bf16_t* wp = w + (long long)jrow * ldw + lhi * 8;
  const int niter = (K + 15) >> 4;
  for (int it0 = wave; it0 < niter; it0 += NW * UNR) {
    bf16x8 a[UNR][G], bb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = it0 + u * NW;
      const int ko = it * 16 + lhi * 8;
      const bool kv = (it < niter) && (ko < K);
      if (kv && bvalid) bb[u] = *reinterpret_cast<const bf16x8*>(ip + it * 16);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[u][e] = (__bf16)0.f;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (kv) a[u][g] = *reinterpret_cast<const bf16x8*>(wp + (long long)g * gate_stride * ldw + it * 16);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) a[u][g][e] = (__bf16)0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][g], bb[u], acc[g], 0, 0, 0);
  }
}

// Sum the NW partial tiles; wave q (< 4) receives rows 8q'+... of its quarter:
// out[g][e] = sum_w acc_w[g][4*wave + e]. `red` is NW*16*64 floats of LDS.
// Accumulator layout: acc[r] is row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31, so the
// quarter q = r>>2 of wave q is rows j0 + 8q + 4*(lane>>5) + e.
template <int G, int NW>
__device__ __forceinline__ void tile_reduce_quarters(f32x16 (&acc)[G], float* red,
                                                     float (&out)[G][4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[g][r];
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) s += red[(w2 * 16 + 4 * wave + e) * 64 + lane];
        out[g][e] = s;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

}  // namespace os2s

namespace os2s {

// ---- latency-optimised variant -------------------------------------------------------
// The step kernels are bound by DEPENDENT memory round trips, not by bandwidth or FLOPs, so
// a wave issues every load of its share of the reduction (KS k-slices: 2*KS 16-byte loads)
// before the first MFMA — one round trip per tile. `wrow` is the weight row this lane
// supplies to the A operand (tile row lane&31) or nullptr for a padding row, `irow` its
// input row (column lane&31) or nullptr for a padding column; `safe` is any readable row of
// K elements (padding lanes load it and discard the data). K % 8 == 0; NW * KS * 16 >= K for a single batch.
// CONTIG: wave w takes KS consecutive k-slices (whole 128-byte lines of a row per wave)
// instead of slices w, w+NW, ... (each line shared by 4 waves).
template <int NW, int KS, bool CONTIG = false>
__device__ __forceinline__ void tile_gemm_prefetch(const bf16_t* __restrict__ wrow,
                                                   const bf16_t* __restrict__ irow, int K,
                                                   f32x16& acc, const bf16_t* __restrict__ safe) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lhi = lane >> 5;
  const int niter = (K + 15) >> 4;
  const bf16_t* wsafe = wrow ? wrow : safe;
  const bf16_t* isafe = irow ? irow : safe;
  for (int base = 0; base < niter; base += NW * KS) {
    // Loads are UNCONDITIONAL (addresses clamped, padding zeroed afterwards with a select): a
    // load inside a branch is waited for at the join, which serialises the round trips.
    u32x4 a[KS], bb[KS];
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int it = CONTIG ? base + wave * KS + i : base + wave + NW * i;
      const int ko = min(it * 16 + lhi * 8, K - 8);
      a[i] = *reinterpret_cast<const u32x4*>(wsafe + ko);
      bb[i] = *reinterpret_cast<const u32x4*>(isafe + ko);
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int it = CONTIG ? base + wave * KS + i : base + wave + NW * i;
      const bool kv = it < niter && it * 16 + lhi * 8 < K;
      const u32x4 av = (kv && wrow) ? a[i] : zero;
      const u32x4 bv = (kv && irow) ? bb[i] : zero;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                    acc, 0, 0, 0);
    }
  }
}

// Sum NW partial tiles through LDS (`red`: NW*16*64 floats). Wave q < 4 receives, for every
// accumulator group g = r >> 2, the element r = 4g + q:  out[g] = sum_w acc_w[4g + q].
// With tile rows laid out as row = 8*gate + unit this hands wave q all four gates of unit
// (q + 4*(lane>>5)) of the tile for batch column lane&31.
template <int NW>
__device__ __forceinline__ void tile_reduce_units(const f32x16& acc, float* red, float (&out)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) s += red[(w2 * 16 + 4 * g + wave) * 64 + lane];
      out[g] = s;
    }
  }
}

// Same for a plain 32-row tile: wave q < 4 receives rows 8q' ... as tile_reduce_quarters
// (single accumulator group).
template <int NW>
__device__ __forceinline__ void tile_reduce_rows(const f32x16& acc, float* red, float (&out)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) s += red[(w2 * 16 + 4 * wave + e) * 64 + lane];
      out[e] = s;
    }
  }
}

// Sum the first EIGHT rows of NW partial tiles (a tile with 8 real rows: accumulator elements
// 0..3 are rows 4*(lane>>5) + e): wave 0 receives out[e] = sum_w acc_w[e]. `red`: NW*4*64 floats.
template <int NW>
__device__ __forceinline__ void tile_reduce_rows8(const f32x16& acc, float* red, float (&out)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(wave * 4 + e) * 64 + lane] = acc[e];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) s += red[(w2 * 4 + e) * 64 + lane];
      out[e] = s;
    }
  }
}

// Same with the WEIGHT operand stored as OCP e4m3 (1 byte per element, one fp32 scale per row
// applied by the caller to the finished dot product): 8 bytes per lane and k-slice instead of 16
// — the step kernels stream the whole weight matrix once per time step, so halving its bytes
// halves the L2/MALL traffic of the recurrence. The bytes are widened to bf16 in registers
// (v_cvt_scalef32_pk_bf16_fp8, exact: every e4m3 value is a bf16 value) and go through the same
// bf16 MFMA; activations stay bf16.
template <int NW, int KS>
__device__ __forceinline__ void tile_gemm_prefetch_w8(const uint8_t* __restrict__ wrow,
                                                      const bf16_t* __restrict__ irow, int K,
                                                      f32x16& acc, const uint8_t* __restrict__ wsafe_,
                                                      const bf16_t* __restrict__ isafe_) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lhi = lane >> 5;
  const int niter = (K + 15) >> 4;
  const uint8_t* wsafe = wrow ? wrow : wsafe_;
  const bf16_t* isafe = irow ? irow : isafe_;
  for (int base = 0; base < niter; base += NW * KS) {
    u32x2 a[KS];
    u32x4 bb[KS];
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int it = base + wave + NW * i;
      const int ko = min(it * 16 + lhi * 8, K - 8);
      a[i] = *reinterpret_cast<const u32x2*>(wsafe + ko);
      bb[i] = *reinterpret_cast<const u32x4*>(isafe + ko);
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int it = base + wave + NW * i;
      const bool kv = it < niter && it * 16 + lhi * 8 < K;
      u32x4 av;
      av[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a[i][0], 1.0f, false));
      av[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a[i][0], 1.0f, true));
      av[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a[i][1], 1.0f, false));
      av[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a[i][1], 1.0f, true));
      av = (kv && wrow) ? av : zero;
      const u32x4 bv = (kv && irow) ? bb[i] : zero;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                    acc, 0, 0, 0);
    }
  }
}

}  // namespace os2s
