// Single-query multi-head attention for incremental (beam-search) decoding, gfx950.
//
// Reference: TransformerDecoder._get_symbols_to_logits_fn / predict
// (open_seq2seq/decoders/transformer_decoder.py:232-326) call the decoder stack on ONE new
// position per beam; SelfAttention appends that position's k, v to cache["layer_n"]
// (parts/transformer/attention_layer.py:133-139) and the beam search re-gathers the whole
// [batch*beam, i, hidden] caches by parent beam every step (beam_search.py:277-285); the
// encoder-decoder attention re-projects the (beam-tiled) encoder output every step.
//
// Here nothing is gathered or re-projected:
//  * self-attention: K/V caches [N, Tmax, D] are append-only — row n, slot i holds what beam
//    row n produced at step i — and an int32 ancestry table anc[n, j] names the row holding
//    position j of beam n's history. The beam search permutes only that table
//    (os2s_gather_rows with the parent rows from os2s_beam_step);
//  * encoder-decoder attention: K/V of the packed encoder output are projected once per
//    sentence; beam row n reads sentence n / beam.
// One wave per (beam row, head): 64 lanes score 64 keys at a time against the query held
// in registers (each lane reads one 128-byte key row), softmax over the wave (DPP), then the
// value reduction with lanes split 4 keys x 16 channel-quads (8-byte loads) and a final
// 2-step butterfly. Pure HBM/latency work: bytes = 2 * len * 128 B per (row, head).
#include "os2s_common.hpp"

namespace os2s {

constexpr int kDaHeads = 4;      // heads (waves) per workgroup
constexpr int kDaDh = 64;

struct DecAttnArgs {
  const bf16_t* q; long long ldq;           // [N, H*64]
  const bf16_t* k; const bf16_t* v;         // cache [N, Tmax, D]  or packed [Nk, ld_t]
  long long ld_t;                           // elements between consecutive positions
  long long ld_row;                         // elements between cache rows (self mode)
  const bf16_t* knew; const bf16_t* vnew;   // [N, .] this step's k, v (self mode)
  long long ldnew;
  bf16_t* kw; bf16_t* vw;                   // cache write pointers (self mode)
  int32_t* anc; int anc_ld;                 // [N, anc_ld]
  const int32_t* cu_k; int beam;            // cross mode
  const int32_t* step_dev; int step;        // self mode: position being produced
  int H, max_len;
  float scale;
  bf16_t* o; long long ldo;
};

template <bool SELF>
__global__ __launch_bounds__(kDaHeads * 64) void decode_attention_kernel(DecAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x, h = blockIdx.y * kDaHeads + wave;
  if (h >= p.H) return;
  float* sc = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * p.max_len;
  int* rowi = reinterpret_cast<int*>(sc + p.max_len);
  const int step = SELF ? (p.step_dev ? p.step_dev[1] : p.step) : 0;
  if (SELF && step >= p.max_len) return;     // search already stopped at the last slot
  int len, base = 0;
  if (SELF) len = step + 1;
  else {
    const int b = n / p.beam;
    base = p.cu_k[b];
    len = min(p.cu_k[b + 1] - base, p.max_len);
  }
  const long long hoff = (long long)h * kDaDh;
  // ---- append this step's k, v (the wave owns its 64 channels of row n) ----------------------
  if (SELF) {
    p.kw[((long long)n * p.ld_row) + (long long)step * p.ld_t + hoff + lane] = p.knew[(long long)n * p.ldnew + hoff + lane];
    p.vw[((long long)n * p.ld_row) + (long long)step * p.ld_t + hoff + lane] = p.vnew[(long long)n * p.ldnew + hoff + lane];
    if (h == 0 && lane == 0) p.anc[(long long)n * p.anc_ld + step] = n;
  }
  // ---- query: 64 bf16, identical in every lane ------------------------------------------------
  u32x4 qv[8];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(p.q + (long long)n * p.ldq + hoff);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = qp[e];
  }
  // ---- scores -----------------------------------------------------------------------------------
  float m = -INFINITY;
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    if (j < len) {
      const bf16_t* kp;
      int row = 0;
      if (SELF) {
        row = j == step ? n : p.anc[(long long)n * p.anc_ld + j];
        kp = j == step ? p.knew + (long long)n * p.ldnew + hoff
                       : p.k + (long long)row * p.ld_row + (long long)j * p.ld_t + hoff;
      } else {
        kp = p.k + (long long)(base + j) * p.ld_t + hoff;
      }
      const u32x4* kq = reinterpret_cast<const u32x4*>(kp);
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const u32x4 kv = kq[e];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          acc += bflo(kv[w]) * bflo(qv[e][w]);
          acc += bfhi(kv[w]) * bfhi(qv[e][w]);
        }
      }
      s = acc * p.scale;
      sc[j] = s;
      rowi[j] = row;
    }
    m = fmaxf(m, s);
  }
  m = wave_max_dpp(m);
  float l = 0.f;
  for (int j = lane; j < len; j += 64) {
    const float e = __expf(sc[j] - m);
    sc[j] = e;
    l += e;
  }
  l = wave_sum_dpp(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  // (LDS writes above are read below by other lanes of the same wave)
  __builtin_amdgcn_wave_barrier();
  // ---- weighted values: lane = (key group jg, channel quad dq) --------------------------------
  const int jg = lane >> 4, dq = lane & 15;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int j0 = 0; j0 < len; j0 += 4) {
    const int j = j0 + jg;
    if (j < len) {
      const bf16_t* vp;
      if (SELF) {
        vp = j == step ? p.vnew + (long long)n * p.ldnew + hoff
                       : p.v + (long long)rowi[j] * p.ld_row + (long long)j * p.ld_t + hoff;
      } else {
        vp = p.v + (long long)(base + j) * p.ld_t + hoff;
      }
      const u32x2 vv = *reinterpret_cast<const u32x2*>(vp + dq * 4);
      const float w = sc[j];
      a0 += w * bflo(vv[0]); a1 += w * bfhi(vv[0]);
      a2 += w * bflo(vv[1]); a3 += w * bfhi(vv[1]);
    }
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64);
    a2 += __shfl_xor(a2, o, 64); a3 += __shfl_xor(a3, o, 64);
  }
  if (jg == 0) {
    u32x2 out;
    out[0] = pack2bf(a0 * inv, a1 * inv);
    out[1] = pack2bf(a2 * inv, a3 * inv);
    *reinterpret_cast<u32x2*>(p.o + (long long)n * p.ldo + hoff + dq * 4) = out;
  }
}

}  // namespace os2s

using namespace os2s;

static int launch_decode_attention(hipStream_t s, const DecAttnArgs& a, int N, bool self) {
  OS2S_REQUIRE(N >= 1 && a.H >= 1 && a.max_len >= 1);
  const int gy = (a.H + kDaHeads - 1) / kDaHeads;
  const size_t smem = (size_t)kDaHeads * 2 * a.max_len * 4;
  OS2S_REQUIRE(smem <= 64 * 1024);
  if (self)
    OS2S_LAUNCH(decode_attention_kernel<true>, dim3(N, gy), dim3(kDaHeads * 64), smem, s, a);
  else
    OS2S_LAUNCH(decode_attention_kernel<false>, dim3(N, gy), dim3(kDaHeads * 64), smem, s, a);
  return OS2S_OK;
}

extern "C" int os2s_decode_self_attention(os2s_stream_t stream, const uint16_t* q, long long ldq,
                                          const uint16_t* knew, const uint16_t* vnew,
                                          long long ldnew, uint16_t* kcache, uint16_t* vcache,
                                          int32_t* ancestry, int N, int H, int dh, int Tmax,
                                          int step, const int32_t* status_dev, float scale,
                                          uint16_t* o, long long ldo) {
  OS2S_REQUIRE(q && knew && vnew && kcache && vcache && ancestry && o);
  if (dh != kDaDh) return OS2S_ERR_UNSUPPORTED;
  OS2S_REQUIRE(Tmax >= 1 && step >= 0 && step < Tmax);
  OS2S_REQUIRE(ldq % 8 == 0 && ldnew % 8 == 0 && ldo % 4 == 0);
  DecAttnArgs a = {};
  const long long D = (long long)H * dh;
  a.q = q; a.ldq = ldq; a.k = kcache; a.v = vcache; a.ld_t = D; a.ld_row = D * Tmax;
  a.knew = knew; a.vnew = vnew; a.ldnew = ldnew; a.kw = kcache; a.vw = vcache;
  a.anc = ancestry; a.anc_ld = Tmax; a.step_dev = status_dev; a.step = step;
  a.H = H; a.max_len = Tmax; a.scale = scale; a.o = o; a.ldo = ldo;
  return launch_decode_attention((hipStream_t)stream, a, N, true);
}

extern "C" int os2s_decode_cross_attention(os2s_stream_t stream, const uint16_t* q, long long ldq,
                                           const uint16_t* k, const uint16_t* v, long long ldkv,
                                           const int32_t* cu_k, int beam, int N, int H, int dh,
                                           int max_len, float scale, uint16_t* o, long long ldo) {
  OS2S_REQUIRE(q && k && v && cu_k && o && beam >= 1);
  if (dh != kDaDh) return OS2S_ERR_UNSUPPORTED;
  OS2S_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0);
  DecAttnArgs a = {};
  a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ld_t = ldkv; a.cu_k = cu_k; a.beam = beam;
  a.H = H; a.max_len = max_len; a.scale = scale; a.o = o; a.ldo = ldo;
  return launch_decode_attention((hipStream_t)stream, a, N, false);
}
