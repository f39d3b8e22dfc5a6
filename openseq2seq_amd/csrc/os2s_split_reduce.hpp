// Deterministic in-launch reduction of fp32 partial tiles ("split units") for the 512-thread
// ping-pong kernels (conv1d_pp_kernel, conv1d_wgrad_pp_kernel) and the 256-thread one-wave-per-SIMD
// weight-gradient kernels (conv1d_wgrad_sw.hpp).
//
// A unit of work whose reduction dimension was cut into f pieces is finished like this:
//   * every piece stores its 256 x 256-element fp32 partial tile (128 registers per thread,
//     lane-linear 16-B stores) into its slab of the caller's workspace;
//   * release at agent scope, then ONE relaxed atomic ticket per piece
//     (cdna_hip_programming.md, split-K recipe: stores -> s_waitcnt vmcnt(0) -> barrier ->
//     lane 0: fence(release, agent) -> s_waitcnt vmcnt(0) -> fetch_add);
//   * the piece that draws ticket f-1 acquires, re-reads ALL f slabs in piece order (its own
//     included, so the fp32 summation order never depends on which piece came last) and
//     continues into the epilogue; every other piece returns. Nobody spins, so nothing is
//     assumed about which workgroups are co-resident.
// The ticket is reset to zero by the reducer: the workspace's ticket area is zero before and
// after every launch.
#pragma once
#include "os2s_common.hpp"

namespace os2s {

constexpr int kSplitSlabFloats = 256 * 256;      // one fp32 partial tile (256 KB)
constexpr size_t kSplitTicketBytes = 4096;       // 1024 int32 tickets at the workspace start

// at(v) -> f32x16& for v = 0..NV-1 (the thread's NV accumulator tiles), statically indexed; NT threads per
// workgroup, NT * NV * 16 = 256 x 256 floats (512 x 8: the ping-pong kernels; 256 x 16: the one-wave-per-SIMD
// weight-gradient kernels). Returns true in the reducing workgroup (accumulators then hold the full sums).
template <int NT = 512, int NV = 8, class At>
__device__ __forceinline__ bool split_publish_and_reduce(At&& at, float* slab0, int* ticket,
                                                        int piece, int f, char* smem, int tid) {
  static_assert(NT * NV * 16 == kSplitSlabFloats, "a partial tile is 256 x 256 floats");
  float* const mine = slab0 + (size_t)piece * kSplitSlabFloats;
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x16& a = at(v);
      f32x4 x = {a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]};
      *reinterpret_cast<f32x4*>(mine + (((v * 4 + g4) * NT + tid) << 2)) = x;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile int*>(smem) = old;
  }
  __syncthreads();
  const int old = *reinterpret_cast<volatile int*>(smem);
  if (old != f - 1) return false;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // piece order (deterministic), 8 independent 16-B loads in flight per thread
#pragma unroll
  for (int h = 0; h < NV / 2; ++h) {
    f32x4 sum[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) sum[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jj = 0; jj < f; ++jj) {
      const float* const sl = slab0 + (size_t)jj * kSplitSlabFloats;
      f32x4 t[8];
#pragma unroll
      for (int v = 0; v < 8; ++v)
        t[v] = *reinterpret_cast<const f32x4*>(sl + ((((h * 2 + (v >> 2)) * 4 + (v & 3)) * NT + tid) << 2));
#pragma unroll
      for (int v = 0; v < 8; ++v) sum[v] += t[v];
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      f32x16& a = at(h * 2 + (v >> 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * (v & 3) + e] = sum[v][e];
    }
  }
  return true;
}

// The same protocol in two halves for kernels whose accumulators are pinned to the accumulation registers (256 of
// them: nothing can be summed back INTO them without moving everything through vector registers): split_publish
// stores the partial tile and draws the ticket (true in the last arriver, after its acquire), split_reduce_emit
// sums the slabs in piece order, 8 register groups (g = 8 h + k: tile v = g >> 2, elements 4 (g & 3) .. + 3) at a
// time, and hands each batch to emit8(h, sums) — the kernel's epilogue — instead of writing it back.
template <int NT, int NV, class At>
__device__ __forceinline__ bool split_publish(At&& at, float* slab0, int* ticket, int piece, int f, char* smem, int tid) {
  static_assert(NT * NV * 16 == kSplitSlabFloats, "a partial tile is 256 x 256 floats");
  float* const mine = slab0 + (size_t)piece * kSplitSlabFloats;
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x16& a = at(v);
      f32x4 x = {a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]};
      *reinterpret_cast<f32x4*>(mine + (((v * 4 + g4) * NT + tid) << 2)) = x;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile int*>(smem) = old;
  }
  __syncthreads();
  const int old = *reinterpret_cast<volatile int*>(smem);
  if (old != f - 1) return false;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return true;
}
template <int NT, int NV, class Emit>
__device__ __forceinline__ void split_reduce_emit(const float* slab0, int f, int tid, Emit&& emit8) {
#pragma unroll
  for (int h = 0; h < NV / 2; ++h) {
    f32x4 sum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jj = 0; jj < f; ++jj) {
      const float* const sl = slab0 + (size_t)jj * kSplitSlabFloats;
      f32x4 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        t[k] = *reinterpret_cast<const f32x4*>(sl + (((h * 8 + k) * NT + tid) << 2));
#pragma unroll
      for (int k = 0; k < 8; ++k) sum[k] += t[k];
    }
    emit8(h, sum);
  }
}

// Tail-split decision shared by the ping-pong kernels: U units on G CUs, r = U mod G units left
// for the last round; cut them f ways when the model says the tail gets shorter. Costs in
// microseconds, fitted on MI355X (tools/bench_conv_split.py, tools/bench_wgrad_shapes.py; refitted
// after the epilogue work of round 2): a split unit adds ~20 us (pipeline fill, partial-tile store
// + release, reducer epilogue), ~1.5 us per partial tile the reducer reads back and 0.17 us per
// piece of aggregate workspace traffic.
__host__ __device__ __forceinline__ int split_factor(int r, int G, float round_us, int fmax, int max_pieces) {
  int f = 1;
  float best = round_us;
  for (int ff = 2; ff <= fmax; ++ff) {
    if (r * ff > max_pieces) break;
    const float t = (float)((r * ff + G - 1) / G) * round_us / ff + 20.f + 1.5f * ff + 0.17f * (r * ff);
    if (t < 0.95f * best) { best = t; f = ff; }
  }
  return f;
}

}  // namespace os2s
