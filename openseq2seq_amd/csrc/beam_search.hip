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
//   row       one 1024-thread workgroup per beam row keeps the row in registers (one 16-byte
//             read per 8 logits): logsumexp, candidate log-probs, then k = 2*beam rounds of
//             wave-level arg-max (DPP) over 64-bit keys (order-preserving float bits : ~flat
//             index) = tf.nn.top_k order: descending value, lower index first among equals;
//             the 16 waves' winners are merged by wave 0.  (V > 65536: lse + chunk kernels.)
//   select    one workgroup per batch item: merge the row winners, then the alive / finished
//             bookkeeping (stable ranks computed in parallel), the sequence gathers and the
//             loop condition
// The loop condition is evaluated on the device (status[0] = running): once it clears every
// kernel of later steps is a no-op, so the host may enqueue steps ahead and poll the flag
// only every few steps without changing the result. Caches are NOT gathered here: the
// select kernel emits the parent row of every new alive beam and the caller gathers whatever
// it keeps per beam (os2s_gather_rows) — for the Transformer that is only an int32 ancestry
// table, the K/V caches never move (decode_attention.hip).
// The pass is HBM/latency bound: the logits are read exactly once.
#include "os2s_common.hpp"

namespace os2s {

constexpr float kBeamInf = 32768.0f;      // beam_search.py:26
constexpr int kChunkThreads = 256;
constexpr int kChunkPer = 16;             // candidates per thread
constexpr int kChunk = kChunkThreads * kChunkPer;
constexpr int kMaxKeep = 64;              // 2*beam <= 64
constexpr int kListCap = 4096;            // >= (kMaxKeep - 1) * 64 + 1 candidates above the threshold

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

// ---- wave-level top-k machinery -----------------------------------------------------------------
// u32 max over a fully active wave with DPP moves (see wave_max_dpp in os2s_common.hpp)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t x) {
  x = max(x, dpp_mov_u32<0xB1, 0xf>(x));
  x = max(x, dpp_mov_u32<0x4E, 0xf>(x));
  x = max(x, dpp_mov_u32<0x124, 0xf>(x));
  x = max(x, dpp_mov_u32<0x128, 0xf>(x));
  x = max(x, dpp_mov_u32<0x142, 0xa>(x));
  x = max(x, dpp_mov_u32<0x143, 0xc>(x));
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// arg-max of 64-bit keys over the wave: high words first, then low words among the leaders
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long best) {
  const uint32_t hi = (uint32_t)(best >> 32), lo = (uint32_t)best;
  const uint32_t hmax = wave_max_u32_dpp(hi);
  const uint32_t lmax = wave_max_u32_dpp(hi == hmax ? lo : 0u);
  return ((unsigned long long)hmax << 32) | lmax;
}

// k rounds over keys held in LDS, `per` keys per lane at pool[e * 64 + lane] (0 = empty);
// the keys are consumed (zeroed). Winners go to out[0..k) in descending order. One wave.
__device__ __forceinline__ void wave_topk_lds(unsigned long long* pool, int per, int k,
                                              unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  unsigned long long best = 0;
  int best_e = 0;
  for (int e = 0; e < per; ++e) {
    const unsigned long long kk = pool[e * 64 + lane];
    if (kk > best) { best = kk; best_e = e; }
  }
  for (int r = 0; r < k; ++r) {
    const unsigned long long win = wave_max_key(best);
    if (lane == 0) out[r] = win;
    if (win != 0 && best == win) {
      pool[best_e * 64 + lane] = 0;
      best = 0;
      for (int e = 0; e < per; ++e) {
        const unsigned long long kk = pool[e * 64 + lane];
        if (kk > best) { best = kk; best_e = e; }
      }
    }
  }
}

// ---- fused row kernel: logsumexp + candidate log-probs + top-k of one beam row --------------------
// 1024 threads, PER columns per thread (8 consecutive columns per 16-byte load for bf16).
template <typename T> struct RowVec;
template <> struct RowVec<bf16_t> { static constexpr int W = 8; };
template <> struct RowVec<float> { static constexpr int W = 4; };

// tf.contrib.seq2seq.BeamSearchDecoder scoring of a row (RNN decoders): finished beams put all
// mass on END, candidates are ranked by total log-prob / ((5 + length) / 6)^weight, at time 0
// only beam 0 competes. finished == nullptr selects the Transformer scoring (alive log-prob +
// log-softmax).
struct TfRowMode {
  const int32_t* finished;
  const int32_t* lengths;
  float lpw;
  int eos, time;
  float* lse_out;
};

__device__ __forceinline__ float tf_length_penalty(int len, float w) {
  return w == 0.f ? 1.f : powf((5.f + (float)len) / 6.f, w);
}

template <typename T, int PER>
__global__ __launch_bounds__(1024) void beam_row_topk_kernel(
    const T* __restrict__ logits, long long ld, int V, int beam, int k,
    const float* __restrict__ alive_lp, const int32_t* __restrict__ status,
    unsigned long long* __restrict__ cand, TfRowMode tf) {
  if (!status[0]) return;
  if (tf.finished && tf.time == 0 && (blockIdx.x % beam) != 0) {     // identical start beams
    if ((int)threadIdx.x < k) cand[(long long)blockIdx.x * k + threadIdx.x] = 0ull;
    return;
  }
  constexpr int W = RowVec<T>::W;
  __shared__ float red[16];
  __shared__ unsigned long long wk[16 * kMaxKeep];       // per-wave top-k of the thread maxima
  __shared__ unsigned long long pool[16 * 64];
  __shared__ unsigned long long thr[kMaxKeep];
  __shared__ unsigned long long list[kListCap];          // elements >= threshold
  __shared__ int count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  const T* row = logits + (long long)n * ld;
  const bool vec_ok = ((((unsigned long long)row) | ((unsigned long long)ld * sizeof(T))) & 15ull) == 0;
  float val[PER];
  // element e = it*W + j  <->  column (it*1024 + tid)*W + j
#pragma unroll
  for (int it = 0; it < PER / W; ++it) {
    const int v0 = (it * 1024 + tid) * W;
    if (vec_ok && v0 + W <= V) {
      if constexpr (W == 8) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(row + v0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { val[it * 8 + 2 * j] = bflo(q[j]); val[it * 8 + 2 * j + 1] = bfhi(q[j]); }
      } else {
        const f32x4 q = *reinterpret_cast<const f32x4*>(row + v0);
#pragma unroll
        for (int j = 0; j < 4; ++j) val[it * 4 + j] = q[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < W; ++j) val[it * W + j] = v0 + j < V ? load_logit(row + v0 + j) : -INFINITY;
    }
  }
  auto col_of = [&](int e) { return ((e / W) * 1024 + tid) * W + (e % W); };
  // ---- logsumexp ---------------------------------------------------------------------------------
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < PER; ++e) m = fmaxf(m, val[e]);
  m = wave_max_dpp(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  const float ms = (m == -INFINITY || m == INFINITY) ? 0.f : m;
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < PER; ++e) sum += col_of(e) < V ? __expf(val[e] - ms) : 0.f;
  sum = wave_sum_dpp(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) sum += red[w];
  const float lse = logf(sum) + ms;
  const float a = alive_lp[n];
  const uint32_t fbase = (uint32_t)((n % beam) * V);
  const bool tf_mode = tf.finished != nullptr;
  bool row_fin = false;
  float lp_other = 1.f, lp_eos = 1.f;
  if (tf_mode) {
    row_fin = tf.finished[n] != 0;
    const int len = tf.lengths[n];
    lp_eos = tf_length_penalty(len, tf.lpw);                     // END adds no length
    lp_other = tf_length_penalty(len + (row_fin ? 0 : 1), tf.lpw);
    if (tid == 0 && tf.lse_out) tf.lse_out[n] = lse;
  }
  // ---- candidate keys; threshold = k-th largest of the 1024 per-thread maxima ---------------------------
  // Every row-level top-k element is >= that threshold, and at most (k-1)*PER + 1 elements are
  // (only the k-1 threads whose maximum beats it can hold more than one), so they fit a small
  // LDS list that wave 0 then ranks exactly.
  uint32_t ordv[PER];                       // order-preserving bits of the candidate log-prob
  unsigned long long tbest = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int v = col_of(e);
    float cand_v = (val[e] - lse) + a;                    // _log_prob_from_logits + alive log-prob
    bool valid = v < V;
    if (tf_mode) {
      if (row_fin) { valid = valid && v == tf.eos; cand_v = __fdiv_rn(a, lp_eos); }
      else cand_v = __fdiv_rn(cand_v, v == tf.eos ? lp_eos : lp_other);
    }
    uint32_t u = __float_as_uint(cand_v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    ordv[e] = valid ? u : 0u;
    const unsigned long long kk = valid ? (((unsigned long long)u << 32) | (unsigned long long)(~(fbase + (uint32_t)v))) : 0ull;
    tbest = kk > tbest ? kk : tbest;
  }
  {
    unsigned long long best = tbest;
    for (int r = 0; r < k; ++r) {
      const unsigned long long win = wave_max_key(best);
      if (lane == 0) wk[wave * k + r] = win;
      if (best == win) best = 0;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int total = 16 * k, per = (total + 63) >> 6;
    for (int e = 0; e < per; ++e) pool[e * 64 + lane] = e * 64 + lane < total ? wk[e * 64 + lane] : 0ull;
    wave_topk_lds(pool, per, k, thr);
    if (lane == 0) count = 0;
  }
  __syncthreads();
  const unsigned long long tau = thr[k - 1];
  const uint32_t tau_hi = (uint32_t)(tau >> 32);
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    if (ordv[e] >= tau_hi && ordv[e] != 0u) {
      const unsigned long long kk = ((unsigned long long)ordv[e] << 32) | (unsigned long long)(~(fbase + (uint32_t)col_of(e)));
      if (kk >= tau) {
        const int slot = atomicAdd(&count, 1);
        if (slot < kListCap) list[slot] = kk;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int total = min(count, kListCap), per = (total + 63) >> 6;
    for (int e = total + lane; e < per * 64; e += 64) list[e] = 0ull;       // pad the last stripe
    wave_topk_lds(list, per, k, cand + (long long)n * k);
  }
}

struct BeamStepArgs {
  const unsigned long long* cand;   // [B, ncand]
  int ncand;                        // candidates per batch item (beam * k, or beam * chunks * k)
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
  int32_t* last_ids;                // [B*beam] last token of every new alive beam (may be null)
  int32_t* pos;                     // [B*beam] its position = cur_index + 1   (may be null)
};

constexpr int kSelThreads = 128;          // >= beam + 2*beam entries of the finished merge

__global__ __launch_bounds__(kSelThreads) void beam_select_kernel(BeamStepArgs p) {
  if (!p.status[0]) return;
  extern __shared__ __attribute__((aligned(16))) unsigned long long pool[];   // 64 * per keys
  __shared__ unsigned long long win[kMaxKeep];
  __shared__ float lp[kMaxKeep], av[kMaxKeep], fs[kMaxKeep + kMaxKeep / 2];
  __shared__ int pb[kMaxKeep], id[kMaxKeep], fin[kMaxKeep], ff[kMaxKeep + kMaxKeep / 2];
  __shared__ int a_sel[kMaxKeep];          // alive: candidate index per new beam
  __shared__ int f_sel[kMaxKeep];          // finished: source (< beam: old slot, else beam + c)
  __shared__ float nfs[kMaxKeep];
  __shared__ int nff[kMaxKeep];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = p.status[1];
  const int beam = p.beam, k = p.k, L1 = p.L1;
  const int cur = i & 1, nxt = cur ^ 1;
  // ---- merge the row winners: top k of the batch item (wave 0) --------------------------------------
  const int per = (p.ncand + 63) >> 6;
  if (wave == 0) {
    for (int e = 0; e < per; ++e)
      pool[e * 64 + lane] = e * 64 + lane < p.ncand ? p.cand[(long long)b * p.ncand + e * 64 + lane] : 0ull;
    wave_topk_lds(pool, per, k, win);
  }
  __syncthreads();
  // ---- per-candidate quantities (thread c < k) --------------------------------------------------------
  const float ln = p.lnorm[i + 1];
  if (tid < k) {
    const uint32_t flat = key_index(win[tid]);
    const float l = key_value(win[tid]);
    const int tok = (int)(flat % (uint32_t)p.V);
    const float f = tok == p.eos ? 1.f : 0.f;
    lp[tid] = l;
    pb[tid] = (int)(flat / (uint32_t)p.V);
    id[tid] = tok;
    fin[tid] = tok == p.eos;
    av[tid] = l + f * -kBeamInf;                                    // _get_new_alive_state
    fs[beam + tid] = __fdiv_rn(l, ln) + (1.f - f) * -kBeamInf;      // _get_new_finished_state
    ff[beam + tid] = tok == p.eos;
    if (p.topk_lp) { p.topk_lp[b * k + tid] = l; p.topk_idx[b * k + tid] = (int)flat; }
  }
  if (tid < beam) {
    fs[tid] = p.fin_scores[b * beam + tid];
    ff[tid] = p.fin_flags[b * beam + tid];
  }
  __syncthreads();
  // ---- stable descending ranks: entry e goes to slot #{e' : v[e'] > v[e] or (== and e' < e)} ----------
  if (tid < k) {
    const float v = av[tid];
    int rank = 0;
    for (int c = 0; c < k; ++c) rank += (av[c] > v || (av[c] == v && c < tid)) ? 1 : 0;
    if (rank < beam) a_sel[rank] = tid;
  }
  if (tid < beam + k) {
    const float v = fs[tid];
    int rank = 0;
    for (int c = 0; c < beam + k; ++c) rank += (fs[c] > v || (fs[c] == v && c < tid)) ? 1 : 0;
    if (rank < beam) { f_sel[rank] = tid; nfs[rank] = v; nff[rank] = ff[tid]; }
  }
  __syncthreads();
  if (tid < beam) {
    const int c = a_sel[tid];
    p.alive_lp[b * beam + tid] = av[c];
    p.parent[b * beam + tid] = b * beam + pb[c];
    p.fin_scores[b * beam + tid] = nfs[tid];
    p.fin_flags[b * beam + tid] = nff[tid];
    if (p.last_ids) p.last_ids[b * beam + tid] = id[c];
    if (p.pos) p.pos[b * beam + tid] = i + 1;
  }
  if (tid == 0) {      // _continue_search for the NEXT iteration
    const float best_alive = __fdiv_rn(av[a_sel[0]], p.lnorm[p.status[3]]);
    float lowest = INFINITY;
    bool any = false;
    for (int s = 0; s < beam; ++s) {
      lowest = fminf(lowest, nfs[s] * (nff[s] ? 1.f : 0.f));
      any = any || nff[s];
    }
    lowest += (1.f - (any ? 1.f : 0.f)) * -kBeamInf;
    p.stop[b] = lowest > best_alive;
  }
  // ---- sequences ---------------------------------------------------------------------------
  const long long plane = (long long)p.B * beam * L1;
  const int32_t* a_in = p.alive_seq + cur * plane + (long long)b * beam * L1;
  int32_t* a_out = p.alive_seq + nxt * plane + (long long)b * beam * L1;
  const int32_t* f_in = p.fin_seq + cur * plane + (long long)b * beam * L1;
  int32_t* f_out = p.fin_seq + nxt * plane + (long long)b * beam * L1;
  for (int e = tid; e < beam * (i + 2); e += kSelThreads) {
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
                                 float* alive_lp, float* fin_scores, int32_t* fin_flags,
                                 int32_t* last_ids, int32_t* pos) {
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
    if (last_ids) last_ids[e] = initial_ids[e / beam];
    if (pos) pos[e] = 0;
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
                              int32_t* fin_flags, int32_t* last_ids, int32_t* pos) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && 2 * beam <= kMaxKeep && max_decode_length >= 0);
  OS2S_REQUIRE(initial_ids && status && alive_seq && fin_seq && alive_lp && fin_scores && fin_flags);
  const int L1 = max_decode_length + 1;
  const long long n = 2LL * B * beam * L1;
  OS2S_LAUNCH(beam_init_kernel, dim3((unsigned)min((long long)1024, (n + 255) / 256)), dim3(256), 0,
              (hipStream_t)stream, B, beam, L1, max_decode_length, initial_ids, status, alive_seq,
              fin_seq, alive_lp, fin_scores, fin_flags, last_ids, pos);
  return OS2S_OK;
}

extern "C" int os2s_beam_step(os2s_stream_t stream, const void* logits, int logits_f32, long long ld,
                              int B, int beam, int V, int max_decode_length, int eos_id,
                              const float* lnorm, int32_t* status, int32_t* alive_seq,
                              int32_t* fin_seq, float* alive_lp, float* fin_scores,
                              int32_t* fin_flags, int32_t* parent, float* topk_lp,
                              int32_t* topk_idx, int32_t* last_ids, int32_t* pos, void* workspace) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && 2 * beam <= kMaxKeep && V >= 2 * beam && ld >= V);
  OS2S_REQUIRE(3 * beam <= kSelThreads);
  OS2S_REQUIRE((long long)beam * V < (1LL << 31));
  OS2S_REQUIRE(logits && lnorm && status && alive_seq && fin_seq && alive_lp && fin_scores &&
               fin_flags && parent && workspace);
  const int k = 2 * beam, chunks = os2s_beam_chunks(V), N = B * beam;
  unsigned long long* cand = (unsigned long long*)workspace;
  const long long cand_bytes = (long long)N * chunks * k * 8;
  float* lse = (float*)((char*)workspace + cand_bytes);
  int32_t* stop = (int32_t*)((char*)lse + (((long long)N * 4 + 7) / 8) * 8);
  hipStream_t s = (hipStream_t)stream;
  int ncand;
  if (V <= 65536) {      // fused: one workgroup per beam row
    ncand = beam * k;
#define OS2S_ROW_TOPK(T, PER)                                                                       \
  OS2S_LAUNCH((beam_row_topk_kernel<T, PER>), dim3(N), dim3(1024), 0, s, (const T*)logits, ld, V, \
              beam, k, alive_lp, status, cand, TfRowMode{nullptr, nullptr, 0.f, 0, 0, nullptr})
    if (logits_f32) {
      if (V <= 8192) OS2S_ROW_TOPK(float, 8);
      else if (V <= 32768) OS2S_ROW_TOPK(float, 32);
      else OS2S_ROW_TOPK(float, 64);
    } else {
      if (V <= 8192) OS2S_ROW_TOPK(bf16_t, 8);
      else if (V <= 32768) OS2S_ROW_TOPK(bf16_t, 32);
      else OS2S_ROW_TOPK(bf16_t, 64);
    }
#undef OS2S_ROW_TOPK
  } else {               // very large vocabularies: separate logsumexp, 4096-candidate chunks
    ncand = beam * chunks * k;
    if (logits_f32) {
      OS2S_LAUNCH(beam_lse_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)logits, ld, V, status, lse);
      OS2S_LAUNCH(beam_chunk_topk_kernel<float>, dim3(chunks, N), dim3(kChunkThreads), 0, s,
                  (const float*)logits, ld, V, beam, k, lse, alive_lp, status, cand);
    } else {
      OS2S_LAUNCH(beam_lse_kernel<bf16_t>, dim3(N), dim3(256), 0, s, (const bf16_t*)logits, ld, V, status, lse);
      OS2S_LAUNCH(beam_chunk_topk_kernel<bf16_t>, dim3(chunks, N), dim3(kChunkThreads), 0, s,
                  (const bf16_t*)logits, ld, V, beam, k, lse, alive_lp, status, cand);
    }
  }
  BeamStepArgs a;
  a.cand = cand; a.ncand = ncand;
  a.B = B; a.beam = beam; a.k = k; a.V = V; a.L1 = max_decode_length + 1; a.eos = eos_id;
  a.lnorm = lnorm; a.status = status; a.alive_seq = alive_seq; a.fin_seq = fin_seq;
  a.alive_lp = alive_lp; a.fin_scores = fin_scores; a.fin_flags = fin_flags; a.parent = parent;
  a.stop = stop; a.topk_lp = topk_lp; a.topk_idx = topk_idx; a.last_ids = last_ids; a.pos = pos;
  const size_t smem = (size_t)((a.ncand + 63) / 64) * 64 * 8;
  OS2S_REQUIRE(smem <= 48 * 1024);
  OS2S_LAUNCH(beam_select_kernel, dim3(B), dim3(kSelThreads), smem, s, a);
  return OS2S_OK;
}

namespace os2s {

// Merge the per-row winners of one batch item and advance the BeamSearchDecoder state.
struct TfSelectArgs {
  const unsigned long long* cand;   // [B*beam, k]
  const void* logits; int logits_f32; long long ld;
  const float* lse;                 // [N]
  int B, beam, V, eos;
  float* log_probs; int32_t* finished; int32_t* lengths;     // [N] in/out
  int32_t* word_ids; int32_t* parent; float* scores;         // [N] out
};

__global__ __launch_bounds__(64) void tf_beam_select_kernel(TfSelectArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long pool[];
  __shared__ unsigned long long win[kMaxKeep];
  __shared__ float old_lp[kMaxKeep];
  __shared__ int old_fin[kMaxKeep], old_len[kMaxKeep];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int beam = p.beam, k = beam, ncand = beam * k;
  const int per = (ncand + 63) >> 6;
  for (int e = 0; e < per; ++e)
    pool[e * 64 + lane] = e * 64 + lane < ncand ? p.cand[(long long)b * ncand + e * 64 + lane] : 0ull;
  if (lane < beam) {
    old_lp[lane] = p.log_probs[b * beam + lane];
    old_fin[lane] = p.finished[b * beam + lane];
    old_len[lane] = p.lengths[b * beam + lane];
  }
  wave_topk_lds(pool, per, k, win);
  __syncthreads();
  if (lane < beam) {
    const uint32_t flat = key_index(win[lane]);
    const int pb = (int)(flat / (uint32_t)p.V), v = (int)(flat % (uint32_t)p.V);
    const int prow = b * beam + pb;
    const bool pf = old_fin[pb] != 0;
    float total = old_lp[pb];
    if (!pf) {
      const float x = p.logits_f32 ? reinterpret_cast<const float*>(p.logits)[(long long)prow * p.ld + v]
                                   : bf2f(reinterpret_cast<const bf16_t*>(p.logits)[(long long)prow * p.ld + v]);
      total += x - p.lse[prow];
    }
    const int n = b * beam + lane;
    p.log_probs[n] = total;
    p.finished[n] = pf || v == p.eos;
    p.lengths[n] = old_len[pb] + (pf ? 0 : 1);
    p.word_ids[n] = v;
    p.parent[n] = prow;
    p.scores[n] = key_value(win[lane]);
  }
}

}  // namespace os2s

extern "C" long long os2s_tf_beam_workspace_bytes(int B, int beam) {
  const long long N = (long long)B * beam;
  return N * beam * 8 + ((N * 4 + 7) / 8) * 8 + 16;
}

extern "C" int os2s_tf_beam_step(os2s_stream_t stream, const void* logits, int logits_f32, long long ld,
                                 int B, int beam, int V, int eos_id, int time,
                                 float length_penalty_weight, float* log_probs, int32_t* finished,
                                 int32_t* lengths, int32_t* word_ids, int32_t* parent, float* scores,
                                 void* workspace) {
  OS2S_REQUIRE(B >= 1 && beam >= 1 && beam <= kMaxKeep && V >= beam && V <= 65536 && ld >= V && time >= 0);
  OS2S_REQUIRE(logits && log_probs && finished && lengths && word_ids && parent && scores && workspace);
  OS2S_REQUIRE((long long)beam * V < (1LL << 31));
  const int N = B * beam, k = beam;
  unsigned long long* cand = (unsigned long long*)workspace;
  float* lse = (float*)((char*)workspace + (long long)N * k * 8);
  int32_t* one = (int32_t*)((char*)lse + (((long long)N * 4 + 7) / 8) * 8);     // "running" flag = 1
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(one, 1, 16, s) != hipSuccess) return OS2S_ERR_LAUNCH;     // non-zero = running
  const TfRowMode tf = {finished, lengths, length_penalty_weight, eos_id, time, lse};
#define OS2S_ROW_TOPK_TF(T, PER)                                                                   \
  OS2S_LAUNCH((beam_row_topk_kernel<T, PER>), dim3(N), dim3(1024), 0, s, (const T*)logits, ld, V, \
              beam, k, log_probs, one, cand, tf)
  if (logits_f32) {
    if (V <= 8192) OS2S_ROW_TOPK_TF(float, 8);
    else if (V <= 32768) OS2S_ROW_TOPK_TF(float, 32);
    else OS2S_ROW_TOPK_TF(float, 64);
  } else {
    if (V <= 8192) OS2S_ROW_TOPK_TF(bf16_t, 8);
    else if (V <= 32768) OS2S_ROW_TOPK_TF(bf16_t, 32);
    else OS2S_ROW_TOPK_TF(bf16_t, 64);
  }
#undef OS2S_ROW_TOPK_TF
  TfSelectArgs a;
  a.cand = cand; a.logits = logits; a.logits_f32 = logits_f32; a.ld = ld; a.lse = lse;
  a.B = B; a.beam = beam; a.V = V; a.eos = eos_id;
  a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.word_ids = word_ids; a.parent = parent; a.scores = scores;
  const size_t smem = (size_t)((beam * k + 63) / 64) * 64 * 8;
  OS2S_LAUNCH(tf_beam_select_kernel, dim3(B), dim3(64), smem, s, a);
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
