// Global-style-token pieces of the Tacotron2 encoder (_embed_style,
// open_seq2seq/encoders/tacotron2_encoder.py:341-505), gfx950.
//
//  * tf.nn.rnn_cell.GRUCell under dynamic_rnn(sequence_length): the reference encoder's
//    recurrent summary of the down-sampled mel (T' <= 16 steps, 128 units):
//        [r, u] = sigmoid([x, h] Wg + bg),  c = tanh([x, r*h] Wc + bc),  h' = u*h + (1-u)*c
//    (the reset gate multiplies h BEFORE the candidate matmul — unlike the cuDNN form of
//    rnn.hip). Samples are independent and the layer is tiny, so ONE workgroup per sample
//    runs the whole time loop in a single launch (state in LDS, recurrent weights fp32 from L2).
//  * the multi-head "bahdanau" token attention (parts/transformer/attention_layer.py:171-186):
//        w[b,h,n] = softmax_n sum_d tanh(att_v[d] * tanh(k[n,h,d] + q[b,h,d])),  out = w . v
//    over N (32) style tokens: one wave per (sample, head), lane = depth.
#include "os2s_common.hpp"

namespace os2s {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------ GRUCell forward
// gxg [B,T,2H] bf16 = x Wg_x + bg (r | u), gxc [B,T,H] bf16 = x Wc_x + bc;
// wgh [H,2H], wch [H,H] fp32 (TF kernel rows of the state part, [in, out]).
__global__ __launch_bounds__(256) void gru_tf_fwd_kernel(
    const bf16_t* __restrict__ gxg, const bf16_t* __restrict__ gxc, const float* __restrict__ wgh,
    const float* __restrict__ wch, const int32_t* __restrict__ lens, int T, int H,
    float* __restrict__ h_seq /* [B,T+1,H] */, float* __restrict__ r_seq, float* __restrict__ u_seq,
    float* __restrict__ c_seq /* [B,T,H] */, bf16_t* __restrict__ hprev16, bf16_t* __restrict__ rh16,
    float* __restrict__ h_final) {
  extern __shared__ float sm[];
  float* h = sm;          // [H]
  float* rh = sm + H;     // [H]
  float* ru = sm + 2 * H; // [2H]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lens ? min(max(lens[b], 0), T) : T;
  for (int j = tid; j < H; j += 256) { h[j] = 0.f; h_seq[((long long)b * (T + 1)) * H + j] = 0.f; }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const long long row = (long long)b * T + t;
    const bool live = t < len;
    if (live) {
      for (int o = tid; o < 2 * H; o += 256) {
        float a = bf2f(gxg[row * 2 * H + o]);
        for (int k = 0; k < H; ++k) a += h[k] * wgh[(long long)k * 2 * H + o];
        ru[o] = sigm(a);
      }
      __syncthreads();
      for (int j = tid; j < H; j += 256) rh[j] = ru[j] * h[j];
      __syncthreads();
      float cn[2] = {0.f, 0.f};
      int n = 0;
      for (int j = tid; j < H; j += 256, ++n) {
        float a = bf2f(gxc[row * H + j]);
        for (int k = 0; k < H; ++k) a += rh[k] * wch[(long long)k * H + j];
        cn[n & 1] = tanhf(a);
        r_seq[row * H + j] = ru[j];
        u_seq[row * H + j] = ru[H + j];
        c_seq[row * H + j] = cn[n & 1];
        hprev16[row * H + j] = f2bf(h[j]);
        rh16[row * H + j] = f2bf(rh[j]);
      }
      __syncthreads();
      n = 0;
      for (int j = tid; j < H; j += 256, ++n) h[j] = ru[H + j] * h[j] + (1.f - ru[H + j]) * cn[n & 1];
    } else {
      for (int j = tid; j < H; j += 256) {
        r_seq[row * H + j] = 0.f; u_seq[row * H + j] = 1.f; c_seq[row * H + j] = 0.f;
        hprev16[row * H + j] = f2bf(0.f); rh16[row * H + j] = f2bf(0.f);
      }
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) h_seq[((long long)b * (T + 1) + t + 1) * H + j] = h[j];
  }
  for (int j = tid; j < H; j += 256) h_final[(long long)b * H + j] = h[j];
}

// ------------------------------------------------------------------ GRUCell backward
// dh_final [B,H] fp32 -> dgxg [B,T,2H], dgxc [B,T,H] bf16 (zero for t >= len); wghT [2H,H],
// wchT [H,H] fp32 transposed state kernels.
__global__ __launch_bounds__(256) void gru_tf_bwd_kernel(
    const float* __restrict__ dh_final, const float* __restrict__ wghT, const float* __restrict__ wchT,
    const int32_t* __restrict__ lens, int T, int H, const float* __restrict__ h_seq,
    const float* __restrict__ r_seq, const float* __restrict__ u_seq, const float* __restrict__ c_seq,
    bf16_t* __restrict__ dgxg, bf16_t* __restrict__ dgxc) {
  extern __shared__ float sm[];
  float* dh = sm;            // [H]
  float* dcp = sm + H;       // [H]   d(candidate pre-activation)
  float* drh = sm + 2 * H;   // [H]   d(r*h)
  float* dg = sm + 3 * H;    // [2H]  d(r pre | u pre)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lens ? min(max(lens[b], 0), T) : T;
  for (int j = tid; j < H; j += 256) dh[j] = dh_final[(long long)b * H + j];
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const long long row = (long long)b * T + t;
    if (t >= len) {
      for (int j = tid; j < H; j += 256) {
        dgxg[row * 2 * H + j] = f2bf(0.f); dgxg[row * 2 * H + H + j] = f2bf(0.f); dgxc[row * H + j] = f2bf(0.f);
      }
      continue;
    }
    const float* hp = h_seq + ((long long)b * (T + 1) + t) * H;
    for (int j = tid; j < H; j += 256) {
      const float u = u_seq[row * H + j], c = c_seq[row * H + j];
      const float d = dh[j];
      const float dcv = d * (1.f - u) * (1.f - c * c);
      dcp[j] = dcv;
      dg[H + j] = d * (hp[j] - c) * u * (1.f - u);
      dgxc[row * H + j] = f2bf(dcv);
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {   // d(r*h)[j] = sum_o dcp[o] Wch[j,o]
      float a = 0.f;
      for (int o = 0; o < H; ++o) a += dcp[o] * wchT[(long long)o * H + j];
      drh[j] = a;
      const float r = r_seq[row * H + j];
      dg[j] = a * hp[j] * r * (1.f - r);
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {
      const float u = u_seq[row * H + j], r = r_seq[row * H + j];
      float a = dh[j] * u + drh[j] * r;
      for (int o = 0; o < 2 * H; ++o) a += dg[o] * wghT[(long long)o * H + j];
      dgxg[row * 2 * H + j] = f2bf(dg[j]);
      dgxg[row * 2 * H + H + j] = f2bf(dg[H + j]);
      drh[j] = a;     // next dh (drh is free now)
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) dh[j] = drh[j];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ token attention
// q [B, heads*dh] bf16, k/v [N, heads*dh] bf16, att_v [dh] fp32 -> out [B, heads*dh] bf16,
// w [B, heads, N] fp32 (saved). dh == 64 (one lane per depth), N <= 64.
__global__ __launch_bounds__(64) void gst_attn_fwd_kernel(const bf16_t* __restrict__ q,
                                                          const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v,
                                                          const float* __restrict__ att_v, int heads,
                                                          int N, bf16_t* __restrict__ out,
                                                          float* __restrict__ w) {
  const int b = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
  const int D = heads * 64;
  const float qv = bf2f(q[(long long)b * D + h * 64 + d]), av = att_v[d];
  float sc[64];
  float mx = -INFINITY;
#pragma unroll
  for (int n = 0; n < 64; ++n) {
    sc[n] = -INFINITY;
    if (n < N) {
      const float t1 = tanhf(bf2f(k[(long long)n * D + h * 64 + d]) + qv);
      sc[n] = wave_sum_dpp(tanhf(av * t1));
      mx = fmaxf(mx, sc[n]);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int n = 0; n < 64; ++n) if (n < N) { sc[n] = __expf(sc[n] - mx); den += sc[n]; }
  float o = 0.f;
#pragma unroll
  for (int n = 0; n < 64; ++n) {
    if (n < N) {
      const float wn = sc[n] / den;
      o += wn * bf2f(v[(long long)n * D + h * 64 + d]);
      if (d == 0) w[((long long)b * heads + h) * N + n] = wn;
    }
  }
  out[(long long)b * D + h * 64 + d] = f2bf(o);
}

// dout [B,D] bf16 -> dq [B,D] bf16, dk/dv [N,D] fp32 (atomic accumulate), datt_v [64] fp32
__global__ __launch_bounds__(64) void gst_attn_bwd_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
    const bf16_t* __restrict__ v, const float* __restrict__ att_v, const float* __restrict__ w,
    int B, int heads, int N, bf16_t* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
    float* __restrict__ datt_v) {
  // grid (B, heads) — or (1, 1) in deterministic mode: then every address gets its adds from ONE thread,
  // in (head, sample) order
  const int d = threadIdx.x;
  const int D = heads * 64;
  for (int h = blockIdx.y; h < heads; h += gridDim.y)
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
  const long long off = (long long)h * 64 + d;
  const float qv = bf2f(q[(long long)b * D + off]), av = att_v[d], dov = bf2f(dout[(long long)b * D + off]);
  float dw[64];
  float dot = 0.f;
#pragma unroll
  for (int n = 0; n < 64; ++n) {
    dw[n] = 0.f;
    if (n < N) {
      const float wn = w[((long long)b * heads + h) * N + n];
      dw[n] = wave_sum_dpp(dov * bf2f(v[(long long)n * D + off]));
      dot += wn * dw[n];
      __hip_atomic_fetch_add(dv + (long long)n * D + off, wn * dov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  float dqa = 0.f, dava = 0.f;
#pragma unroll
  for (int n = 0; n < 64; ++n) {
    if (n < N) {
      const float wn = w[((long long)b * heads + h) * N + n];
      const float ds = wn * (dw[n] - dot);                       // softmax backward
      const float t1 = tanhf(bf2f(k[(long long)n * D + off]) + qv);
      const float t2 = tanhf(av * t1);
      const float dz = ds * (1.f - t2 * t2);                      // d(att_v * t1)
      dava += dz * t1;
      const float dpre = dz * av * (1.f - t1 * t1);               // d(k + q)
      dqa += dpre;
      __hip_atomic_fetch_add(dk + (long long)n * D + off, dpre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  dq[(long long)b * D + off] = f2bf(dqa);
  __hip_atomic_fetch_add(datt_v + d, dava, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace os2s

using namespace os2s;

extern "C" int os2s_gru_tf_fwd(os2s_stream_t stream, const uint16_t* gxg, const uint16_t* gxc,
                               const float* wgh, const float* wch, const int32_t* lens, int B, int T,
                               int H, float* h_seq, float* r_seq, float* u_seq, float* c_seq,
                               uint16_t* hprev16, uint16_t* rh16, float* h_final) {
  OS2S_REQUIRE(gxg && gxc && wgh && wch && h_seq && r_seq && u_seq && c_seq && hprev16 && rh16 && h_final);
  OS2S_REQUIRE(B >= 1 && T >= 1 && H >= 1 && H <= 512);
  OS2S_LAUNCH(gru_tf_fwd_kernel, dim3(B), dim3(256), (size_t)4 * H * sizeof(float), (hipStream_t)stream,
              (const bf16_t*)gxg, (const bf16_t*)gxc, wgh, wch, lens, T, H, h_seq, r_seq, u_seq, c_seq,
              (bf16_t*)hprev16, (bf16_t*)rh16, h_final);
  return OS2S_OK;
}

extern "C" int os2s_gru_tf_bwd(os2s_stream_t stream, const float* dh_final, const float* wghT,
                               const float* wchT, const int32_t* lens, int B, int T, int H,
                               const float* h_seq, const float* r_seq, const float* u_seq,
                               const float* c_seq, uint16_t* dgxg, uint16_t* dgxc) {
  OS2S_REQUIRE(dh_final && wghT && wchT && h_seq && r_seq && u_seq && c_seq && dgxg && dgxc);
  OS2S_REQUIRE(B >= 1 && T >= 1 && H >= 1 && H <= 512);
  OS2S_LAUNCH(gru_tf_bwd_kernel, dim3(B), dim3(256), (size_t)5 * H * sizeof(float), (hipStream_t)stream,
              dh_final, wghT, wchT, lens, T, H, h_seq, r_seq, u_seq, c_seq, (bf16_t*)dgxg, (bf16_t*)dgxc);
  return OS2S_OK;
}

extern "C" int os2s_gst_attention_fwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                                      const uint16_t* v, const float* att_v, int B, int heads, int N,
                                      uint16_t* out, float* w) {
  OS2S_REQUIRE(q && k && v && att_v && out && w && B >= 1 && heads >= 1 && N >= 1 && N <= 64);
  OS2S_LAUNCH(gst_attn_fwd_kernel, dim3(B, heads), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)q,
              (const bf16_t*)k, (const bf16_t*)v, att_v, heads, N, (bf16_t*)out, w);
  return OS2S_OK;
}

extern "C" int os2s_gst_attention_bwd(os2s_stream_t stream, const uint16_t* dout, const uint16_t* q,
                                      const uint16_t* k, const uint16_t* v, const float* att_v,
                                      const float* w, int B, int heads, int N, uint16_t* dq, float* dk,
                                      float* dv, float* datt_v) {
  OS2S_REQUIRE(dout && q && k && v && att_v && w && dq && dk && dv && datt_v && N >= 1 && N <= 64);
  const dim3 grid = os2s_deterministic() ? dim3(1, 1) : dim3(B, heads);
  OS2S_LAUNCH(gst_attn_bwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dout,
              (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, att_v, w, B, heads, N, (bf16_t*)dq, dk,
              dv, datt_v);
  return OS2S_OK;
}
