// hipBLASLt comparison back end — NOT part of the product library. Since round 2 every matmul of
// the models runs on the in-tree MFMA kernels (os2s_gemm_nt, conv1d_wgrad1x1_pp_kernel, the conv /
// recurrent kernels); this wrapper is kept for A/B measurements only (OS2S_GEMM=lt, tools/bench_*):
// tools/lt/build.sh builds tools/lt/libos2s_lt.so, which openseq2seq_amd.capi.matmul_lt loads on
// demand. libos2s_hip.so does not link the vendor library.
//
// Row-major in, row-major out; descriptors + heuristics are cached per problem.
#include <hipblaslt/hipblaslt.h>

#include <array>
#include <map>
#include <mutex>

#include "../../openseq2seq_amd/csrc/os2s_common.hpp"

#include <cstdio>

namespace {

void lt_note(int status, const char* where) { fprintf(stderr, "[os2s_lt] %s: status %d\n", where, status); }

struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

hipblasLtHandle_t g_handle = nullptr;
constexpr size_t kWsBytes = 64ull << 20;
// one workspace per stream: matmuls enqueued on different streams may run concurrently. A
// stream that is being captured into a graph must not allocate: it gets the workspace created
// with the handle (graph replays are serialised on their launch stream).
void* g_ws_capture = nullptr;
std::map<hipStream_t, void*> g_ws_by_stream;

void* stream_workspace(hipStream_t s) {
  auto it = g_ws_by_stream.find(s);
  if (it != g_ws_by_stream.end()) return it->second;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return g_ws_capture;
  void* p = nullptr;
  if (hipMalloc(&p, kWsBytes) != hipSuccess) return nullptr;
  g_ws_by_stream[s] = p;
  return p;
}
std::mutex g_mu;
std::map<std::array<long long, 10>, LtPlan> g_plans;

}  // namespace

// C[M,N] (row-major, bf16 or fp32) = op(A)[M,K] . op(B)[K,N] + beta * C, bf16 inputs, fp32
// accumulation. A is stored [M,K] (a_is_T = 0) or [K,M] (1); B is stored [K,N] (0) or [N,K] (1).
extern "C" int os2s_lt_matmul(os2s_stream_t stream, const uint16_t* A, int a_is_T, long long lda,
                              const uint16_t* B, int b_is_T, long long ldb, void* C, int c_f32,
                              long long ldc, int M, int N, int K, float beta) {
  OS2S_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && lda >= 1 && ldb >= 1 && ldc >= N);
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_handle) {
    if (hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return OS2S_ERR_LAUNCH;
    if (hipMalloc(&g_ws_capture, kWsBytes) != hipSuccess) return OS2S_ERR_WORKSPACE;
  }
  void* const g_ws = stream_workspace((hipStream_t)stream);
  if (!g_ws) return OS2S_ERR_WORKSPACE;
  const std::array<long long, 10> key = {M, N, K, a_is_T, b_is_T, lda, ldb, ldc, c_f32, beta != 0.f};
  LtPlan& pl = g_plans[key];
  if (!pl.ok) {
    // row-major C = op(A) op(B)  <=>  column-major C^T[N,M] = op(B)^T op(A)^T
    if (hipblasLtMatmulDescCreate(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS)
      return OS2S_ERR_LAUNCH;
    const hipblasOperation_t ta = b_is_T ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    const hipblasOperation_t tb = a_is_T ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    hipblasLtMatmulDescSetAttribute(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    // first operand = the B buffer: stored [K,N] -> column-major N x K; stored [N,K] -> K x N
    if (hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, b_is_T ? K : N, b_is_T ? N : K, ldb) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&pl.lb, HIP_R_16BF, a_is_T ? M : K, a_is_T ? K : M, lda) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&pl.lc, c_f32 ? HIP_R_32F : HIP_R_16BF, N, M, ldc) != HIPBLAS_STATUS_SUCCESS)
      return OS2S_ERR_LAUNCH;
    hipblasLtMatmulPreference_t pref;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return OS2S_ERR_LAUNCH;
    uint64_t ws = kWsBytes;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    constexpr int kCand = 32;
    hipblasLtMatmulHeuristicResult_t res[kCand];
    int found = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, pl.desc, pl.la, pl.lb, pl.lc, pl.lc,
                                                               pref, kCand, res, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) {
      lt_note((int)st, "hipblasLtMatmulAlgoGetHeuristic");
      return OS2S_ERR_UNSUPPORTED;
    }
    // The heuristic's first choice is often not the fastest for tall reductions into small
    // outputs (weight gradients: 128x128 tiles on a 1024x1024 output use 64 of 256 CUs): time the
    // candidates once per problem into a scratch D and keep the best (lazily built plan cache).
    int best = 0;
    if (found > 1) {
      void* scratch = nullptr;
      const size_t dbytes = (size_t)M * (size_t)ldc * (c_f32 ? 4 : 2);
      hipEvent_t e0, e1;
      if (hipMalloc(&scratch, dbytes) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
          hipEventCreate(&e1) == hipSuccess) {
        const float one = 1.f, zero = 0.f;
        float best_ms = 1e30f;
        hipDeviceSynchronize();      // no other stream's work under the timings
        for (int c = 0; c < found; ++c) {
          if (res[c].state != HIPBLAS_STATUS_SUCCESS || res[c].workspaceSize > kWsBytes) continue;
          auto run = [&]() {
            return hipblasLtMatmul(g_handle, pl.desc, &one, B, pl.la, A, pl.lb, &zero, scratch, pl.lc, scratch,
                                   pl.lc, &res[c].algo, g_ws, res[c].workspaceSize, (hipStream_t)stream);
          };
          if (run() != HIPBLAS_STATUS_SUCCESS) continue;      // warm-up
          hipEventRecord(e0, (hipStream_t)stream);
          run(); run();
          hipEventRecord(e1, (hipStream_t)stream);
          hipEventSynchronize(e1);
          float ms = 0.f;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms > 0.f && ms < best_ms) { best_ms = ms; best = c; }
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
      }
      if (scratch) hipFree(scratch);
    }
    pl.algo = res[best].algo;
    pl.ws = res[best].workspaceSize;
    pl.ok = true;
  }
  const float alpha = 1.f;
  const hipblasStatus_t st = hipblasLtMatmul(g_handle, pl.desc, &alpha, B, pl.la, A, pl.lb, &beta, C, pl.lc, C,
                                             pl.lc, &pl.algo, g_ws, pl.ws, (hipStream_t)stream);
  if (st != HIPBLAS_STATUS_SUCCESS) {
    lt_note((int)st, "hipblasLtMatmul");
    return OS2S_ERR_LAUNCH;
  }
  return OS2S_OK;
}
