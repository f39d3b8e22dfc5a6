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
// One wave per (beam row, head), lanes = 8 key groups x 8 sixteen-byte channel chunks: every
// wave load reads 8 whole 128-byte K (or V) rows, the score is an 8-lane DPP sum, each group
// keeps an online softmax (running max, sum, 8 output channels per lane) over its keys and the
// groups are merged with three butterfly steps — a single pass, no LDS.
// Pure HBM/latency work: bytes = 2 * len * 128 B per (row, head).
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

__device__ __forceinline__ float sum8_dpp(float x) {     // sum over aligned groups of 8 lanes
  x += dpp_mov<0xB1, 0xf>(0.f, x);     // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E, 0xf>(0.f, x);     // quad_perm [2,3,0,1]
  x += dpp_mov<0x141, 0xf>(0.f, x);    // row_half_mirror: lane i <-> 7 - i
  return x;
}

__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) { f[2 * w] = bflo(v[w]); f[2 * w + 1] = bfhi(v[w]); }
}

// One wave per (beam row, head). Lane = (key group g = lane >> 3, 16-byte channel chunk
// c = lane & 7): a wave load covers 8 whole 128-byte key (or value) rows. Each group runs an
// online softmax over its keys (j = 8*it + g); the 8 groups are merged at the end.
template <bool SELF>
__global__ __launch_bounds__(kDaHeads * 64) void decode_attention_kernel(DecAttnArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x, h = blockIdx.y * kDaHeads + wave;
  if (h >= p.H) return;
  const int g = lane >> 3, c = lane & 7;
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
  float q8[8];
  unpack8(*reinterpret_cast<const u32x4*>(p.q + (long long)n * p.ldq + hoff + c * 8), q8);
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int j0 = 0; j0 < len; j0 += 64) {
    int rowv = n;
    if (SELF) {
      const int jj = j0 + lane;
      if (jj < len && jj != step) rowv = p.anc[(long long)n * p.anc_ld + jj];
    }
    const int nit = min(8, (len - j0 + 7) >> 3);
    for (int it = 0; it < nit; ++it) {
      const int j = j0 + it * 8 + g;
      const bool valid = j < len;
      const int jc = valid ? j : len - 1;
      const bf16_t *kp, *vp;
      if (SELF) {
        const int row = __shfl(rowv, it * 8 + g, 64);
        const bool fresh = jc == step;
        const long long off = fresh ? (long long)n * p.ldnew : (long long)row * p.ld_row + (long long)jc * p.ld_t;
        kp = (fresh ? p.knew : p.k) + off + hoff + c * 8;
        vp = (fresh ? p.vnew : p.v) + off + hoff + c * 8;
      } else {
        kp = p.k + (long long)(base + jc) * p.ld_t + hoff + c * 8;
        vp = p.v + (long long)(base + jc) * p.ld_t + hoff + c * 8;
      }
      const u32x4 kq = *reinterpret_cast<const u32x4*>(kp);
      const u32x4 vq = *reinterpret_cast<const u32x4*>(vp);
      float k8[8], v8[8];
      unpack8(kq, k8);
      unpack8(vq, v8);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += k8[e] * q8[e];
      s = sum8_dpp(s) * p.scale;
      if (!valid) s = -INFINITY;
      const float mn = fmaxf(m, s);
      const float corr = m == -INFINITY ? 0.f : __expf(m - mn);
      const float pe = valid ? __expf(s - mn) : 0.f;
      l = l * corr + pe;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = acc[e] * corr + pe * v8[e];
      m = mn;
    }
  }
  // ---- merge the 8 key groups (lanes with equal c) ------------------------------------------------
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) {
    const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, mo);
    const float a = m == -INFINITY ? 0.f : __expf(m - mn);
    const float b = mo == -INFINITY ? 0.f : __expf(mo - mn);
    l = l * a + lo * b;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], off, 64) * b;
    m = mn;
  }
  if (g == 0) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) o[w] = pack2bf(acc[2 * w] * inv, acc[2 * w + 1] * inv);
    *reinterpret_cast<u32x4*>(p.o + (long long)n * p.ldo + hoff + c * 8) = o;
  }
}

}  // namespace os2s

using namespace os2s;

static int launch_decode_attention(hipStream_t s, const DecAttnArgs& a, int N, bool self) {
  OS2S_REQUIRE(N >= 1 && a.H >= 1 && a.max_len >= 1);
  const int gy = (a.H + kDaHeads - 1) / kDaHeads;
  const size_t smem = 0;
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
  OS2S_REQUIRE(ldq % 8 == 0 && ldnew % 8 == 0 && ldo % 8 == 0);
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
  OS2S_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0);
  DecAttnArgs a = {};
  a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ld_t = ldkv; a.cu_k = cu_k; a.beam = beam;
  a.H = H; a.max_len = max_len; a.scale = scale; a.o = o; a.ldo = ldo;
  return launch_decode_attention((hipStream_t)stream, a, N, false);
}
