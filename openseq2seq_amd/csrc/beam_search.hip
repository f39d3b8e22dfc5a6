// Transformer beam search on the device, gfx950.
//
// Reference: SequenceBeamSearch (open_seq2seq/parts/transformer/beam_search.py:62-383), a
// tf.while_loop whose body (a) turns the [batch*beam, V] logits of the alive beams into
// log-probabilities and takes the 2*beam best extensions per batch item (_grow_alive_seq
// :238-296), (b) keeps the best `beam` that did not emit EOS (_get_new_alive_state :298-327),
// (c) merges those that did, scored log_prob / ((5+len)/6)^alpha, into the finished set
// (_get_new_finished_state :329-383), and stops (_continue_search :163-203) once no alive
// beam can beat the worst finished one.
//
// Device formulation. The loop state lives in HBM for the whole search (ping-pong sequence
// buffers, log-probs, scores, flags) and three launches advance it by one step:
//   lse       one workgroup per beam row: logsumexp of the V logits (one read of the row)
//   chunk     one workgroup per 4096-candidate chunk: k = 2*beam rounds of arg-max over
//             64-bit keys (order-preserving float bits : ~flat index), i.e. tf.nn.top_k
//             order — descending value, lower index first among equals
//   select    one workgroup per batch item: merge the chunk winners, then the (tiny) alive /
//             finished bookkeeping and the sequence gathers, and the loop condition
// The loop condition is evaluated on the device (status[0] = running): once it clears every
// kernel of later steps is a no-op, so the host may enqueue steps ahead and poll the flag
// only every few steps without changing the result. Caches are NOT gathered here: the
// select kernel emits the parent row of every new alive beam and the caller gathers whatever
// it keeps per beam (os2s_gather_rows) — for the Transformer that is only an int32 ancestry
// table, the K/V caches never move (decode_attention.hip).
// The pass is HBM/latency bound: the logits row is read twice (lse, chunk).
#include "os2s_common.hpp"

namespace os2s {

constexpr float kBeamInf = 32768.0f;      // beam_search.py:26
constexpr int kChunkThreads = 256;
constexpr int kChunkPer = 16;             // candidates per thread
constexpr int kChunk = kChunkThreads * kChunkPer;
constexpr int kMaxKeep = 64;              // 2*beam <= 64

// status words (int32): 0 running, 1 cur_index, 2 ticket, 3 max_decode_length
__device__ __forceinline__ unsigned long long make_key(float v, uint32_t flat) {
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(~flat);
}
__device__ __forceinline__ float key_value(unsigned long long key) {
  uint32_t u = (uint32_t)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t key_index(unsigned long long key) { return ~(uint32_t)key; }

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(k, o, 64);
    k = other > k ? other : k;
  }
  return k;
}

template <typename T> __device__ __forceinline__ float load_logit(const T* p);
template <> __device__ __forceinline__ float load_logit<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_logit<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__global__ __launch_bounds__(256) void beam_lse_kernel(const T* __restrict__ logits, long long ld,
                                                       int V, const int32_t* __restrict__ status,
                                                       float* __restrict__ lse) {
  if (!status[0]) return;
  __shared__ float red[4];
  const T* row = logits + (long long)blockIdx.x * ld;
  float m = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, load_logit(row + v));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float ms = (m == -INFINITY || m == INFINITY) ? 0.f : m;
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += expf(load_logit(row + v) - ms);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) lse[blockIdx.x] = logf((red[0] + red[1]) + (red[2] + red[3])) + ms;
}

// k rounds of block-wide arg-max over keys held in registers; winners to out[0..k)
template <int PER>
__device__ __forceinline__ void block_topk_regs(unsigned long long (&key)[PER], int k,
                                                unsigned long long* __restrict__ out,
                                                unsigned long long* red) {
  const int wave = threadIdx.x >> 6;
  unsigned long long best = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) best = key[e] > best ? key[e] : best;
  for (int r = 0; r < k; ++r) {
    const unsigned long long w = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[(r & 1) * 4 + wave] = w;
    __syncthreads();
    const unsigned long long* rr = red + (r & 1) * 4;
    unsigned long long win = rr[0];
    win = rr[1] > win ? rr[1] : win;
    win = rr[2] > win ? rr[2] : win;
    win = rr[3] > win ? rr[3] : win;
    if (threadIdx.x == 0) out[r] = win;
    if (win != 0 && best == win) {          // keys are unique (they embed the index)
      best = 0;
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        if (key[e] == win) key[e] = 0;
        best = key[e] > best ? key[e] : best;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kChunkThreads) void beam_chunk_topk_kernel(
    const T* __restrict__ logits, long long ld, int V, int beam, int k,
    const float* __restrict__ lse, const float* __restrict__ alive_lp,
    const int32_t* __restrict__ status, unsigned long long* __restrict__ cand) {
  if (!status[0]) return;
  __shared__ unsigned long long red[8];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int bm = n % beam;
  const T* row = logits + (long long)n * ld;
  const float l = lse[n], a = alive_lp[n];
  unsigned long long key[kChunkPer];
#pragma unroll
  for (int e = 0; e < kChunkPer; ++e) {
    const int v = chunk * kChunk + e * kChunkThreads + threadIdx.x;
    key[e] = 0;
    if (v < V) {
      const float cand_lp = load_logit(row + v) - l;      // _log_prob_from_logits
      key[e] = make_key(cand_lp + a, (uint32_t)(bm * V + v));
    }
  }
  block_topk_regs<kChunkPer>(key, k, cand + ((long long)n * gridDim.x + chunk) * k, red);
}

struct BeamStepArgs {
  const unsigned long long* cand;   // [B, beam*chunks, k]
  int ncand;                        // beam * chunks * k
  int B, beam, k, V, L1, eos;
  const float* lnorm;               // [max_len + 1] length normalisation per length
  int32_t* status;
  int32_t* alive_seq;               // [2, B, beam, L1]
  int32_t* fin_seq;                 // [2, B, beam, L1]
  float* alive_lp;                  // [B, beam]
  float* fin_scores;                // [B, beam]
  int32_t* fin_flags;               // [B, beam]
  int32_t* parent;                  // [B*beam] flat parent row of each new alive beam
  int32_t* stop;                    // [B]
  float* topk_lp;                   // [B, k]  (debug / tests; may be null)
  int32_t* topk_idx;                // [B, k]
};

__global__ __launch_bounds__(256) void beam_select_kernel(BeamStepArgs p) {
  if (!p.status[0]) return;
  extern __shared__ __attribute__((aligned(16))) unsigned long long pool[];   // ncand keys
  __shared__ unsigned long long red[8];
  __shared__ unsigned long long win[kMaxKeep];
  __shared__ float lp[kMaxKeep];
  __shared__ int pb[kMaxKeep], id[kMaxKeep], fin[kMaxKeep];
  __shared__ int a_sel[kMaxKeep];          // alive: candidate index per new beam
  __shared__ int f_sel[kMaxKeep];          // finished: source (< beam: old slot, else beam + c)
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const int i = p.status[1];
  const int beam = p.beam, k = p.k, L1 = p.L1;
  const int cur = i & 1, nxt = cur ^ 1;
  for (int e = tid; e < p.ncand; e += 256) pool[e] = p.cand[(long long)b * p.ncand + e];
  __syncthreads();
  // ---- merge: k rounds over the pool -------------------------------------------------------
  for (int r = 0; r < k; ++r) {
    unsigned long long best = 0;
    for (int e = tid; e < p.ncand; e += 256) best = pool[e] > best ? pool[e] : best;
    best = wave_max_u64(best);
    if ((tid & 63) == 0) red[(r & 1) * 4 + wave] = best;
    __syncthreads();
    const unsigned long long* rr = red + (r & 1) * 4;
    unsigned long long w = rr[0];
    w = rr[1] > w ? rr[1] : w;
    w = rr[2] > w ? rr[2] : w;
    w = rr[3] > w ? rr[3] : w;
    if (tid == 0) win[r] = w;
    for (int e = tid; e < p.ncand; e += 256)
      if (pool[e] == w) pool[e] = 0;
    __syncthreads();
  }
  // ---- bookkeeping (serial, k <= 64) -------------------------------------------------------
  if (tid == 0) {
    for (int c = 0; c < k; ++c) {
      const uint32_t flat = key_index(win[c]);
      lp[c] = key_value(win[c]);
      pb[c] = (int)(flat / (uint32_t)p.V);
      id[c] = (int)(flat % (uint32_t)p.V);
      fin[c] = id[c] == p.eos;
      if (p.topk_lp) { p.topk_lp[b * k + c] = lp[c]; p.topk_idx[b * k + c] = (int)flat; }
    }
    // alive: top `beam` of lp + fin * -INF (stable)
    float av[kMaxKeep];
    bool used[kMaxKeep];
    for (int c = 0; c < k; ++c) { av[c] = lp[c] + (fin[c] ? 1.f : 0.f) * -kBeamInf; used[c] = false; }
    for (int s = 0; s < beam; ++s) {
      int bi = -1;
      for (int c = 0; c < k; ++c)
        if (!used[c] && (bi < 0 || av[c] > av[bi])) bi = c;
      used[bi] = true;
      a_sel[s] = bi;
    }
    // finished: [old (beam) ; new (k)] by score (stable)
    const float ln = p.lnorm[i + 1];
    float fs[kMaxKeep * 2];
    int ff[kMaxKeep * 2];
    bool fu[kMaxKeep * 2];
    for (int j = 0; j < beam; ++j) { fs[j] = p.fin_scores[b * beam + j]; ff[j] = p.fin_flags[b * beam + j]; fu[j] = false; }
    for (int c = 0; c < k; ++c) {
      fs[beam + c] = __fdiv_rn(lp[c], ln) + (1.f - (fin[c] ? 1.f : 0.f)) * -kBeamInf;
      ff[beam + c] = fin[c];
      fu[beam + c] = false;
    }
    float nfs[kMaxKeep]; int nff[kMaxKeep];
    for (int s = 0; s < beam; ++s) {
      int bi = -1;
      for (int c = 0; c < beam + k; ++c)
        if (!fu[c] && (bi < 0 || fs[c] > fs[bi])) bi = c;
      fu[bi] = true;
      f_sel[s] = bi;
      nfs[s] = fs[bi];
      nff[s] = ff[bi];
    }
    float alp0 = 0.f;
    for (int s = 0; s < beam; ++s) {
      const float v = av[a_sel[s]];
      if (s == 0) alp0 = v;
      p.alive_lp[b * beam + s] = v;
      p.parent[b * beam + s] = b * beam + pb[a_sel[s]];
      p.fin_scores[b * beam + s] = nfs[s];
      p.fin_flags[b * beam + s] = nff[s];
    }
    // _continue_search for the NEXT iteration
    const float best_alive = __fdiv_rn(alp0, p.lnorm[p.status[3]]);
    float lowest = INFINITY;
    bool any = false;
    for (int s = 0; s < beam; ++s) {
      lowest = fminf(lowest, nfs[s] * (nff[s] ? 1.f : 0.f));
      any = any || nff[s];
    }
    lowest += (1.f - (any ? 1.f : 0.f)) * -kBeamInf;
    p.stop[b] = lowest > best_alive;
  }
  __syncthreads();
  // ---- sequences ---------------------------------------------------------------------------
  const long long plane = (long long)p.B * beam * L1;
  const int32_t* a_in = p.alive_seq + cur * plane + (long long)b * beam * L1;
  int32_t* a_out = p.alive_seq + nxt * plane + (long long)b * beam * L1;
  const int32_t* f_in = p.fin_seq + cur * plane + (long long)b * beam * L1;
  int32_t* f_out = p.fin_seq + nxt * plane + (long long)b * beam * L1;
  for (int e = tid; e < beam * (i + 2); e += 256) {
    const int s = e / (i + 2), t = e - s * (i + 2);
    const int c = a_sel[s];
    a_out[s * L1 + t] = t <= i ? a_in[pb[c] * L1 + t] : id[c];
    const int src = f_sel[s];
    int v;
    if (src < beam) v = t <= i ? f_in[src * L1 + t] : 0;
    else v = t <= i ? a_in[pb[src - beam] * L1 + t] : id[src - beam];
    f_out[s * L1 + t] = v;
  }
  // ---- loop condition: last workgroup to arrive closes the step -------------------------------
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const int ticket = atomicAdd(&p.status[2], 1);
    if (ticket == p.B - 1) {
      __threadfence();
      bool all = true;
      for (int bb = 0; bb < p.B; ++bb) all = all && (__atomic_load_n(&p.stop[bb], __ATOMIC_RELAXED) != 0);
      p.status[2] = 0;
      p.status[1] = i + 1;
      __threadfence();
      p.status[0] = (i + 1 < p.status[3]) && !all;
    }
  }
}

__global__ void beam_init_kernel(int B, int beam, int L1, int max_len, const int32_t* __restrict__ initial_ids,
                                 int32_t* status, int32_t* alive_seq, int32_t* fin_seq,
                                 float* alive_lp, float* fin_scores, int32_t* fin_flags) {
  const long long plane = (long long)B * beam * L1;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long e = gid; e < 2 * plane; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e % plane;
    const int t = (int)(r % L1);
    const int b = (int)(r / ((long long)beam * L1));
    alive_seq[e] = (t == 0 && e < plane) ? initial_ids[b] : 0;
    fin_seq[e] = 0;
  }
  for (long long e = gid; e < (long long)B * beam; e += (long long)gridDim.x * blockDim.x) {
    alive_lp[e] = (e % beam) == 0 ? 0.f : -INFINITY;
    fin_scores[e] = -kBeamInf;
    fin_flags[e] = 0;
  }
  if (gid == 0) {
    status[0] = max_len > 0;     // _continue_search on the initial state: nothing finished yet
    status[1] = 0;
    status[2] = 0;
    status[3] = max_len;
  }
}

__global__ void beam_finalize_kernel(int B, int beam, int L1, const int32_t* __restrict__ status,
                                     const int32_t* __restrict__ alive_seq,
                                     const int32_t* __restrict__ fin_seq,
                                     const float* __restrict__ alive_lp,
                                     const float* __restrict__ fin_scores,
                                     const int32_t* __restrict__ fin_flags, int32_t* __restrict__ out_seq,
                                     float* __restrict__ out_scores) {
  const int b = blockIdx.x;
  const int cur = status[1] & 1;
  bool any = false;
  for (int s = 0; s < beam; ++s) any = any || fin_flags[b * beam + s];
  const long long plane = (long long)B * beam * L1;
  const int32_t* src = (any ? fin_seq : alive_seq) + cur * plane + (long long)b * beam * L1;
  for (int e = threadIdx.x; e < beam * L1; e += blockDim.x) out_seq[(long long)b * beam * L1 + e] = src[e];
  for (int s = threadIdx.x; s < beam; s += blockDim.x)
    out_scores[b * beam + s] = any ? fin_scores[b * beam + s] : alive_lp[b * beam + s];
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* __restrict__ src,
                                                          const int32_t* __restrict__ idx,
                                                          long long rows, long long row_words,
                                                          const int32_t* __restrict__ enable,
                                                          uint32_t* __restrict__ dst) {
  const bool on = !enable || enable[0];
  const long long total = rows * row_words;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long r = e / row_words, w = e - r * row_words;
    dst[e] = src[(on ? (long long)idx[r] : r) * row_words + w];
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_beam_chunks(int V) { return (V + kChunk - 1) / kChunk; }

extern "C" long long os2s_beam_workspace_bytes(int B, int beam, int V) {
  const long long k = 2LL * beam;
  const long long cand = (long long)B * beam * os2s_beam_chunks(V) * k * 8;
  const long long lse = (long long)B * beam * 4;
  const long long stop = (long long)B * 4;
  return cand + ((lse + 7) / 8) * 8 + ((stop + 7) / 8) * 8;
}

extern "C" int os2s_beam_init(os2s_stream_t stream, int B, int beam, int max_decode_length,
                              const int32_t* initial_ids, int32_t* status, int32_t* alive_seq,
                              int32_t* fin_seq, float* alive_lp, float* fin_scores,
                              int32_t* fin_flags) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && 2 * beam <= kMaxKeep && max_decode_length >= 0);
  OS2S_REQUIRE(initial_ids && status && alive_seq && fin_seq && alive_lp && fin_scores && fin_flags);
  const int L1 = max_decode_length + 1;
  const long long n = 2LL * B * beam * L1;
  OS2S_LAUNCH(beam_init_kernel, dim3((unsigned)min((long long)1024, (n + 255) / 256)), dim3(256), 0,
              (hipStream_t)stream, B, beam, L1, max_decode_length, initial_ids, status, alive_seq,
              fin_seq, alive_lp, fin_scores, fin_flags);
  return OS2S_OK;
}

extern "C" int os2s_beam_step(os2s_stream_t stream, const void* logits, int logits_f32, long long ld,
                              int B, int beam, int V, int max_decode_length, int eos_id,
                              const float* lnorm, int32_t* status, int32_t* alive_seq,
                              int32_t* fin_seq, float* alive_lp, float* fin_scores,
                              int32_t* fin_flags, int32_t* parent, float* topk_lp,
                              int32_t* topk_idx, void* workspace) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && 2 * beam <= kMaxKeep && V >= 2 * beam && ld >= V);
  OS2S_REQUIRE((long long)beam * V < (1LL << 31));
  OS2S_REQUIRE(logits && lnorm && status && alive_seq && fin_seq && alive_lp && fin_scores &&
               fin_flags && parent && workspace);
  const int k = 2 * beam, chunks = os2s_beam_chunks(V), N = B * beam;
  unsigned long long* cand = (unsigned long long*)workspace;
  const long long cand_bytes = (long long)N * chunks * k * 8;
  float* lse = (float*)((char*)workspace + cand_bytes);
  int32_t* stop = (int32_t*)((char*)lse + (((long long)N * 4 + 7) / 8) * 8);
  hipStream_t s = (hipStream_t)stream;
  if (logits_f32) {
    OS2S_LAUNCH(beam_lse_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)logits, ld, V, status, lse);
    OS2S_LAUNCH(beam_chunk_topk_kernel<float>, dim3(chunks, N), dim3(kChunkThreads), 0, s,
                (const float*)logits, ld, V, beam, k, lse, alive_lp, status, cand);
  } else {
    OS2S_LAUNCH(beam_lse_kernel<bf16_t>, dim3(N), dim3(256), 0, s, (const bf16_t*)logits, ld, V, status, lse);
    OS2S_LAUNCH(beam_chunk_topk_kernel<bf16_t>, dim3(chunks, N), dim3(kChunkThreads), 0, s,
                (const bf16_t*)logits, ld, V, beam, k, lse, alive_lp, status, cand);
  }
  BeamStepArgs a;
  a.cand = cand; a.ncand = beam * chunks * k;
  a.B = B; a.beam = beam; a.k = k; a.V = V; a.L1 = max_decode_length + 1; a.eos = eos_id;
  a.lnorm = lnorm; a.status = status; a.alive_seq = alive_seq; a.fin_seq = fin_seq;
  a.alive_lp = alive_lp; a.fin_scores = fin_scores; a.fin_flags = fin_flags; a.parent = parent;
  a.stop = stop; a.topk_lp = topk_lp; a.topk_idx = topk_idx;
  const size_t smem = (size_t)a.ncand * 8;
  OS2S_REQUIRE(smem <= 48 * 1024);
  OS2S_LAUNCH(beam_select_kernel, dim3(B), dim3(256), smem, s, a);
  return OS2S_OK;
}

extern "C" int os2s_beam_finalize(os2s_stream_t stream, int B, int beam, int max_decode_length,
                                  const int32_t* status, const int32_t* alive_seq,
                                  const int32_t* fin_seq, const float* alive_lp,
                                  const float* fin_scores, const int32_t* fin_flags,
                                  int32_t* out_seq, float* out_scores) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && status && alive_seq && fin_seq && alive_lp && fin_scores &&
               fin_flags && out_seq && out_scores);
  OS2S_LAUNCH(beam_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, beam,
              max_decode_length + 1, status, alive_seq, fin_seq, alive_lp, fin_scores, fin_flags,
              out_seq, out_scores);
  return OS2S_OK;
}

extern "C" int os2s_gather_rows(os2s_stream_t stream, const void* src, const int32_t* idx,
                                long long rows, long long row_bytes, const int32_t* enable,
                                void* dst) {
  OS2S_REQUIRE(src && idx && dst && rows >= 0 && row_bytes >= 4 && row_bytes % 4 == 0 && src != dst);
  if (rows == 0) return OS2S_OK;
  const long long total = rows * (row_bytes / 4);
  OS2S_LAUNCH(gather_rows_kernel, dim3((unsigned)min((long long)4096, (total + 255) / 256)), dim3(256),
              0, (hipStream_t)stream, (const uint32_t*)src, idx, rows, row_bytes / 4, enable,
              (uint32_t*)dst);
  return OS2S_OK;
}
