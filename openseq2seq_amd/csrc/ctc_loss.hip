// CTC loss (forward + gradient w.r.t. the logits) for gfx950.
//
// Reference semantics: tf.nn.ctc_loss(labels, inputs=logits[T,B,V] (time-major,
// pre-softmax), sequence_length, ignore_longer_outputs_than_inputs=True) with
// TF defaults ctc_merge_repeated=True, preprocess_collapse_repeated=False,
// blank = V-1, as called by open_seq2seq/losses/ctc_loss.py:77-82; then
// mask_nans (:84-85, utils.py:366-370) and the batch mean (:88).
//   * a sample whose label does not fit (label_len + #adjacent repeats > T_b) is
//     ignored: loss 0, gradient 0;
//   * the loss returned per sample is -log p(l|x); non-finite values -> 0.
// In TF1 this op runs on the CPU (a host round trip per step); here it stays on
// the device:
//   1. row log-softmax (parallel);
//   2. alpha and beta recursions in log space — sequential in t, one workgroup
//      per (sample, direction), the state row lives in LDS, log-probs are
//      prefetched in chunks of time steps;
//   3. gradient (parallel over (sample, time)): with e_s = exp(alpha_t(s) +
//      beta_t(s) - logp_t(l'_s) - ll) (a posterior, <= 1) the per-class sums are
//      a dense one-hot product over the extended label — deterministic, no
//      atomics:  dlogit[t,v] = softmax_t(v) - sum_{s: l'_s = v} e_s.
#include "os2s_common.hpp"

namespace os2s {

constexpr float kNegInf = -1e30f;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m <= kNegInf * 0.5f) return kNegInf;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m <= kNegInf * 0.5f) return kNegInf;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// logp[b][t][v] = log_softmax(logits[t][b][:]) for t < len[b]
__global__ __launch_bounds__(256) void ctc_log_softmax_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ in_len, int T, int B,
    int V, float* __restrict__ logp) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)T * B) return;
  const int t = (int)(g / B), b = (int)(g - (long long)t * B);
  if (t >= in_len[b]) return;
  const float* row = logits + g * V;
  float m = row[0];
  for (int v = 1; v < V; ++v) m = fmaxf(m, row[v]);
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += expf(row[v] - m);
  const float lz = m + logf(s);
  float* out = logp + ((long long)b * T + t) * V;
  for (int v = 0; v < V; ++v) out[v] = row[v] - lz;
}

constexpr int kTC = 32;  // time steps of log-probs prefetched per chunk

// grid = 2*B: block (b, dir). dir 0 = alpha (forward), 1 = beta (backward).
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(
    const float* __restrict__ logp, const int32_t* __restrict__ labels, int Lmax,
    const int32_t* __restrict__ label_len, const int32_t* __restrict__ in_len, int T,
    int B, int V, int blank, int Smax, float* __restrict__ alpha,
    float* __restrict__ beta, double* __restrict__ coff_a, double* __restrict__ coff_b,
    double* __restrict__ loglik, int32_t* __restrict__ valid) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
  int L = label_len[b];
  L = L < 0 ? 0 : (L > Lmax ? Lmax : L);
  int Tb = in_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  const int S = 2 * L + 1;
  int32_t* lab = reinterpret_cast<int32_t*>(smem_raw);           // [Smax] extended labels
  float* st = reinterpret_cast<float*>(smem_raw + Smax * 4);      // [2][Smax]
  float* lp = st + 2 * Smax;                                      // [kTC][V]
  __shared__ int s_rep;
  __shared__ float s_wmax[4];
  if (threadIdx.x == 0) s_rep = 0;
  __syncthreads();
  const int32_t* labb = labels + (long long)b * Lmax;
  int rep = 0;
  for (int s = threadIdx.x; s < S; s += 256) {
    const int l = (s & 1) ? labb[s >> 1] : blank;
    lab[s] = l;
    if ((s & 1) && s >= 3 && labb[(s >> 1) - 1] == l) ++rep;
  }
  if (rep) atomicAdd(&s_rep, rep);
  __syncthreads();
  const bool ok = (Tb > 0) && (L + s_rep <= Tb);
  if (!ok) {
    if (dir == 0 && threadIdx.x == 0) { loglik[b] = 0.0; valid[b] = 0; }
    return;
  }
  float* outp = (dir == 0 ? alpha : beta) + (long long)b * T * Smax;
  double* coff = (dir == 0 ? coff_a : coff_b) + (long long)b * T;
  // The stored rows are alpha_t(s) - C_t with C_t a running normaliser kept in
  // double: fp32 alphas that grow like -3.4*t lose ~1e-4 absolute per step.
  double C = 0.0;
  const float* lpb = logp + (long long)b * T * V;

  for (int c0 = 0; c0 < Tb; c0 += kTC) {
    // chunk of time steps [c0, c0+n) in processing order
    const int n = min(kTC, Tb - c0);
    __syncthreads();  // previous chunk's readers are done with lp
    for (int i = threadIdx.x; i < n * V; i += 256) {
      const int j = i / V, v = i - j * V;
      const int t = dir == 0 ? (c0 + j) : (Tb - 1 - (c0 + j));
      lp[i] = lpb[(long long)t * V + v];
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
      const int step = c0 + j;
      const int t = dir == 0 ? step : (Tb - 1 - step);
      const float* prev = st + ((step + 1) & 1) * Smax;
      float* cur = st + (step & 1) * Smax;
      const float* lpr = lp + j * V;
      for (int s = threadIdx.x; s < S; s += 256) {
        const int l = lab[s];
        float v;
        if (step == 0) {
          if (dir == 0) v = (s <= 1) ? lpr[l] : kNegInf;
          else v = (s >= S - 2) ? lpr[l] : kNegInf;
        } else if (dir == 0) {
          const float a0 = prev[s];
          const float a1 = s >= 1 ? prev[s - 1] : kNegInf;
          const float a2 = (s >= 2 && l != blank && lab[s - 2] != l) ? prev[s - 2] : kNegInf;
          v = lse3(a0, a1, a2) + lpr[l];
        } else {
          const float a0 = prev[s];
          const float a1 = s + 1 < S ? prev[s + 1] : kNegInf;
          const float a2 = (s + 2 < S && l != blank && lab[s + 2] != l) ? prev[s + 2] : kNegInf;
          v = lse3(a0, a1, a2) + lpr[l];
        }
        if (v < kNegInf) v = kNegInf;
        cur[s] = v;
        outp[(long long)t * Smax + s] = v;
      }
      if (threadIdx.x == 0) coff[t] = C;
      __syncthreads();
      if ((step & 7) == 7 && step + 1 < Tb) {
        // renormalise: subtract the row maximum, remember it in C
        float m = kNegInf;
        for (int s = threadIdx.x; s < S; s += 256) m = fmaxf(m, cur[s]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
        if (m > kNegInf * 0.5f) {
          for (int s = threadIdx.x; s < S; s += 256) {
            const float x = cur[s];
            cur[s] = x > kNegInf * 0.5f ? x - m : kNegInf;
          }
          C += (double)m;
        }
        __syncthreads();
      }
    }
  }
  if (dir == 0 && threadIdx.x == 0) {
    const float* last = st + ((Tb - 1) & 1) * Smax;
    const float l2 = S >= 2 ? lse2(last[S - 1], last[S - 2]) : last[S - 1];
    const double ll = C + (double)l2;
    loglik[b] = ll;
    valid[b] = (l2 > kNegInf * 0.5f && isfinite(l2)) ? 1 : 0;
  }
}

constexpr int kGR = 8;  // time rows per gradient block

// grid = (ceil(T/kGR), B). thread -> (row r = tid/32, class lane v = tid%32 (+32k))
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logp, const int32_t* __restrict__ labels, int Lmax,
    const int32_t* __restrict__ label_len, const int32_t* __restrict__ in_len, int T,
    int B, int V, int blank, int Smax, const float* __restrict__ alpha,
    const float* __restrict__ beta, const double* __restrict__ coff_a,
    const double* __restrict__ coff_b, const double* __restrict__ loglik,
    const int32_t* __restrict__ valid, float grad_scale_host,
    const float* __restrict__ grad_scale_dev, float* __restrict__ dlogits,
    bf16_t* __restrict__ dlogits_bf16, int Vpad) {
  const float grad_scale = grad_scale_dev ? grad_scale_host * (*grad_scale_dev) : grad_scale_host;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kGR;
  int L = label_len[b];
  L = L < 0 ? 0 : (L > Lmax ? Lmax : L);
  int Tb = in_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  const int S = 2 * L + 1;
  const bool ok = valid[b] != 0;
  int32_t* lab = reinterpret_cast<int32_t*>(smem_raw);        // [Smax]
  float* e = reinterpret_cast<float*>(smem_raw + Smax * 4);   // [kGR][Smax]
  const int r = threadIdx.x >> 5, vl = threadIdx.x & 31;
  const int t = t0 + r;
  const bool live = ok && t < Tb;
  if (ok) {
    const int32_t* labb = labels + (long long)b * Lmax;
    for (int s = threadIdx.x; s < S; s += 256) lab[s] = (s & 1) ? labb[s >> 1] : blank;
    __syncthreads();
    const double ll = loglik[b];
    for (int rr = 0; rr < kGR; ++rr) {
      const int tt = t0 + rr;
      if (tt < Tb) {
        const float off = (float)(coff_a[(long long)b * T + tt] + coff_b[(long long)b * T + tt] - ll);
        const float* al = alpha + ((long long)b * T + tt) * Smax;
        const float* be = beta + ((long long)b * T + tt) * Smax;
        const float* lpr = logp + ((long long)b * T + tt) * V;
        for (int s = threadIdx.x; s < S; s += 256) {
          const float x = (al[s] + be[s] - lpr[lab[s]]) + off;
          e[rr * Smax + s] = x > -80.f ? __expf(x) : 0.f;
        }
      }
    }
    __syncthreads();
  }
  if (t < T) {
    for (int v = vl; v < (dlogits_bf16 ? max(V, Vpad) : V); v += 32) {
      float g = 0.f;
      if (live && v < V) {
        float occ = 0.f;
        const float* er = e + r * Smax;
        // blank lives on even s, labels on odd s
        if (v == blank) {
          for (int s = 0; s < S; s += 2) occ += er[s];
        } else {
          for (int s = 1; s < S; s += 2) occ += (lab[s] == v) ? er[s] : 0.f;
        }
        const float pr = __expf(logp[((long long)b * T + t) * V + v]);
        g = (pr - occ) * grad_scale;
      }
      if (v < V && dlogits) dlogits[((long long)t * B + b) * V + v] = g;
      if (dlogits_bf16 && v < Vpad) dlogits_bf16[((long long)b * T + t) * Vpad + v] = f2bf(g);
    }
  }
}

// loss[b] = valid ? -ll : 0 (mask_nans), mean over the WHOLE batch.
__global__ void ctc_finish_kernel(const double* __restrict__ loglik,
                                  const int32_t* __restrict__ valid, int B,
                                  float* __restrict__ loss_per_sample,
                                  float* __restrict__ loss_mean) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
      float l = valid[b] ? (float)(-loglik[b]) : 0.f;
      if (!isfinite(l)) l = 0.f;
      if (loss_per_sample) loss_per_sample[b] = l;
      tot += l;
    }
    if (loss_mean) *loss_mean = tot / (float)B;
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" size_t os2s_ctc_loss_workspace_bytes(int T, int B, int V, int Lmax) {
  const size_t Smax = 2 * (size_t)Lmax + 1;
  size_t n = (size_t)B * T * V * 4;        // logp
  n += 2 * (size_t)B * T * Smax * 4;       // alpha, beta
  n += 2 * (size_t)B * T * 8;              // running normalisers (double)
  n += (size_t)B * 16;                     // loglik (double), valid
  return n + 256;
}

extern "C" int os2s_ctc_loss(os2s_stream_t stream_, const float* logits,
                             const int32_t* in_len, const int32_t* labels,
                             const int32_t* label_len, int T, int B, int V, int Lmax,
                             int blank, float grad_scale, const float* grad_scale_dev,
                             float* loss_per_sample, float* loss_mean, float* dlogits,
                             uint16_t* dlogits_bf16,
                             int Vpad, void* workspace, size_t workspace_bytes) {
  OS2S_REQUIRE(logits && in_len && labels && label_len && workspace);
  OS2S_REQUIRE(T >= 1 && B >= 1 && V >= 2 && Lmax >= 0 && blank >= 0 && blank < V);
  if (dlogits_bf16) OS2S_REQUIRE(Vpad >= V && Vpad % 8 == 0);
  if (workspace_bytes < os2s_ctc_loss_workspace_bytes(T, B, V, Lmax)) return OS2S_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int Smax = 2 * Lmax + 1;
  char* ws = (char*)workspace;
  float* logp = (float*)ws; ws += (size_t)B * T * V * 4;
  float* alpha = (float*)ws; ws += (size_t)B * T * Smax * 4;
  float* beta = (float*)ws; ws += (size_t)B * T * Smax * 4;
  double* coff_a = (double*)ws; ws += (size_t)B * T * 8;
  double* coff_b = (double*)ws; ws += (size_t)B * T * 8;
  double* ll = (double*)ws; ws += (size_t)B * 8;
  int32_t* valid = (int32_t*)ws;

  OS2S_LAUNCH(ctc_log_softmax_kernel, dim3(ceil_div((long long)T * B, 256)), dim3(256), 0,
              stream, logits, in_len, T, B, V, logp);
  const size_t smem_ab = (size_t)Smax * 4 * 3 + (size_t)kTC * V * 4;
  const size_t smem_g = (size_t)Smax * 4 * (1 + kGR);
  if (smem_ab > 64 * 1024 || smem_g > 64 * 1024) return OS2S_ERR_UNSUPPORTED;
  OS2S_LAUNCH(ctc_alpha_beta_kernel, dim3(2 * B), dim3(256), smem_ab, stream, logp, labels,
              Lmax, label_len, in_len, T, B, V, blank, Smax, alpha, beta, coff_a, coff_b, ll, valid);
  if (dlogits || dlogits_bf16) {
    OS2S_LAUNCH(ctc_grad_kernel, dim3(ceil_div(T, kGR), B), dim3(256), smem_g, stream, logp,
                labels, Lmax, label_len, in_len, T, B, V, blank, Smax, alpha, beta, coff_a, coff_b,
                ll, valid, grad_scale, grad_scale_dev, dlogits, dlogits_bf16, Vpad);
  }
  OS2S_LAUNCH(ctc_finish_kernel, dim3(1), dim3(64), 0, stream, ll, valid, B, loss_per_sample,
              loss_mean);
  return OS2S_OK;
}
