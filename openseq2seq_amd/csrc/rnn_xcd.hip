// Persistent, XCD-local forward pass of a cuDNN-form GRU layer (DeepSpeech2: 5 bidirectional layers
// of 800 units, open_seq2seq/encoders/ds2_encoder.py:294-328; the cell of csrc/rnn.hip).
//
// rnn.hip runs one launch per time step: every step pays a dependent kernel boundary (~1.5 us), a
// cold stream of the recurrent weights R [3H, H] (3.84 MB per direction at H = 800) and the launch
// ramp of a 100-workgroup grid — 7.2 us per step, 4 000 steps per train step. Here ONE launch runs
// all T steps of both directions:
//
//   * a direction lives on ONE XCD: its 32 compute units are 32 workgroups (256 workgroups of one per CU;
//     a workgroup reads HW_REG_XCC_ID and takes the next free slot of that XCD's direction), so the
//     per-step exchange of the hidden state goes through that XCD's own L2 — plain 8-byte stores,
//     L1-bypassing loads, no agent-scope fence, nothing crosses the fabric;
//   * weights are STATIONARY IN REGISTERS: a workgroup owns H/32 hidden units = 3H/32 rows of R
//     (120 KB at H = 800), held as MFMA A fragments (wave w keeps the k-steps w, w+8, ... of all row
//     tiles: 96 VGPRs), loaded once per launch;
//   * per step a workgroup gathers h_{t-1} [B, H] (bf16) from the exchange buffer into LDS, runs
//     6 x 4 v_mfma_f32_16x16x32_bf16 per wave, reduces the 8 k-slices through LDS, does the gate
//     math for its own (unit, sample) pairs and publishes its slice of h_t;
//   * the exchange needs no flags: a granule is one naturally aligned 16-byte word {16-bit tag, 7 x bf16},
//     tag = step + 1, written by ONE store (MI355X_MICROARCH.md "handoff" rows); two buffer slots
//     suffice because nobody can publish h_{t+1} before everybody has read h_{t-1};
//   * every wait is bounded: a workgroup that does not see its granules within ~2^17 polls raises
//     the abort flag and leaves, every other workgroup follows; the code is latched into the sticky
//     word os2s_gru_xcd_status() returns and the host layer redoes the step on the per-step path
//     (Model.train_step; launches themselves do not fail on a set word — see "sticky abort word").
#include <cstdio>
#include <cstdlib>

#include "os2s_common.hpp"

namespace os2s {

constexpr int kXcdCus = 32;          // workgroups (= compute units) per direction
constexpr int kXcdThreads = 512;
constexpr int kXcdRTMax = 6;         // row tiles of 16 (>= 3 * ceil(H / 32) rows, H <= 1024)
constexpr int kXcdKS = 4;            // k-steps of 32 per wave (8 waves x 4 x 32 = 1024 >= H)
constexpr int kXcdGC = 4;            // granules a thread polls at once (H = 800, B = 16: 3.6 per thread and step)

struct GruXcdDir {
  const bf16_t* gx;        // [B, T, 3H] input projections (+ input bias)
  const bf16_t* wh;        // [3H, H]
  const float* bh;         // [3H] or null
  float* h32;              // [B, H] state in / out
  bf16_t* y;               // row (b, t) at y + (b*T + t) * ldy
  long long ldy;
  bf16_t* gates;           // [B, T, 4H] saved r, z, n, (R_n h + b_Rn) or null
  unsigned long long* xbuf;   // granules [2 slots][32 workgroups][gpc] x 16 bytes
  int reverse;
};
struct GruXcdArgs {
  int B, T, H, ndir;
  const int32_t* lens;
  int* flags;              // [0] abort (1 = timeout, 2 = placement mismatch)
  GruXcdDir d[2];
};

__device__ __forceinline__ void st_plain_b128(void* p, u32x4 v) {
  // ONE 16-byte store that stays in this XCD's L2 (no sc bits: every reader is on the same XCD and
  // bypasses its L1)
  asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast_(float x) { return 1.f - 2.f / (1.f + __expf(2.f * x)); }

// NB = 16-sample column tiles (B <= 16 NB); RT = 16-row tiles of a workgroup's slice of R (3 * upc rows)
template <int NB, int RT>
__global__ __launch_bounds__(kXcdThreads) void gru_xcd_fwd_kernel(GruXcdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // role = (XCD this workgroup runs on, arrival order on that XCD): 256 workgroups of one per CU put 32
  // on every XCD whatever the dispatch order (blockIdx b mostly lands on XCD b % 8, not always)
  int dir, cu;
  {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dir = xcc & 7;
    if (dir >= a.ndir) return;
    int* const slot = reinterpret_cast<int*>(smem);
    if (tid == 0) *slot = __hip_atomic_fetch_add(a.flags + 8 + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    cu = *slot;
    __syncthreads();
    if (cu >= kXcdCus) {                 // more than one workgroup per CU: not the residency this is built on
      if (tid == 0) __hip_atomic_store(a.flags, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  const GruXcdDir& p = a.d[dir];
  const int B = a.B, T = a.T, H = a.H;
  constexpr int BP = NB * 16;                       // padded batch
  const int upc = (H + kXcdCus - 1) / kXcdCus;      // hidden units of a workgroup
  const int u0 = cu * upc;
  const int nu = max(0, min(upc, H - u0));
  const int NK = (H + 31) / 32;                     // k-steps
  const int HS = NK * 32 * 2 + 16;                  // LDS row stride of h (bytes)
  const int gpc = (upc * BP + 6) / 7;               // granules a workgroup publishes per step (7 values each)
  const int ngran = kXcdCus * gpc;                  // granules of one h vector
  constexpr int PR = RT * 16 + 4;               // row pitch of the partial-sum image (+4: the 16-byte
                                                // writes of the 16 sample columns hit distinct banks)
  // LDS: h [BP][HS] | partial sums [8 waves][BP][PR] f32 | state [upc][BP] f32 | hpub [gpc * 7] bf16
  char* const h_l = smem;
  float* const part = reinterpret_cast<float*>(smem + BP * HS);
  float* const hst = part + 8 * BP * PR;
  bf16_t* const hpub = reinterpret_cast<bf16_t*>(hst + upc * BP);

  // ---- weights -> registers: A fragments (row = lane & 15 of the tile, k = 8 (lane >> 4) .. + 8) ----
  bf16x8 af[kXcdKS][RT];
  {
    const int rrow = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < kXcdKS; ++ks) {
      const int k = (wave + 8 * ks) * 32 + kq;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = rt * 16 + rrow;                // row of this workgroup: gate * upc + unit
        const int g = r / upc, u = r - g * upc;
        bf16x8 v = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        if (g < 3 && u < nu && k < H)                // H % 8 == 0: a 16-byte piece is all in or all out
          v = *reinterpret_cast<const bf16x8*>(p.wh + ((long long)g * H + u0 + u) * H + k);
        af[ks][rt] = v;
      }
    }
  }
  // ---- per-thread constants of the gate math: pair e = unit + upc * sample (unit fastest: the reads
  //      of the partial sums are then conflict-free) ------------------------------------------------
  constexpr int ME = (kXcdCus * BP + kXcdThreads - 1) / kXcdThreads;   // (unit, sample) pairs per thread
  int len_i[ME], ub_i[ME];        // ub = unit | sample << 8 of the thread's pairs
  float bias_i[ME][3];
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = tid + i * kXcdThreads, b = e / upc, u = e - b * upc;
    ub_i[i] = u | (b << 8);
    len_i[i] = -1;
    bias_i[i][0] = bias_i[i][1] = bias_i[i][2] = 0.f;
    if (e < upc * BP && u < nu && b < B) {
      len_i[i] = a.lens ? min(max(a.lens[b], 0), T) : T;
      if (p.bh) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bias_i[i][g] = p.bh[g * H + u0 + u];
      }
    }
  }
  // ---- state: fp32 copy of this workgroup's (unit, sample) pairs; zero rows of h past B / H ----------
  for (int e = tid; e < upc * BP; e += kXcdThreads) {
    const int b = e / upc, u = e - b * upc;
    hst[b + BP * u] = (u < nu && b < B) ? p.h32[(long long)b * H + u0 + u] : 0.f;
  }
  for (int i = tid; i < gpc * 7 + 1; i += kXcdThreads) hpub[i] = 0;
  const unsigned gpc_inv = (unsigned)((0x100000000ull + gpc - 1) / gpc);   // gi / gpc = umulhi(gi, gpc_inv), gi < 2^16
  __syncthreads();
  // h_{-1}: every workgroup reads the whole initial state (plain loads: nobody has written it)
  for (int e = tid; e < B * H; e += kXcdThreads) {
    const int b = e / H, k = e - b * H;
    *reinterpret_cast<bf16_t*>(h_l + b * HS + k * 2) = f2bf(p.h32[e]);
  }
  __syncthreads();

  int failed = 0;
#ifdef OS2S_GRU_XCD_TIMERS
  long long tph[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#define XCD_TICK(i) do { const long long n_ = (long long)__builtin_readcyclecounter(); tph[i] += n_ - tlast; tlast = n_; } while (0)
#else
#define XCD_TICK(i) do { } while (0)
#endif
  for (int s = 0; s < T; ++s) {
    // ---- (0) this step's gate pre-activations: issued now, used after the matrix product ---------
    float pre[ME][3];
    int tt[ME];
#pragma unroll
    for (int i = 0; i < ME; ++i) {
      const int u = ub_i[i] & 255, b = ub_i[i] >> 8;
      tt[i] = -1;
      pre[i][0] = pre[i][1] = pre[i][2] = 0.f;
      if (s < len_i[i]) {
        const int t = p.reverse ? len_i[i] - 1 - s : s;
        tt[i] = t;
        const bf16_t* gp = p.gx + ((long long)b * T + t) * (3 * H) + u0 + u;
        pre[i][0] = bf2f(gp[0]); pre[i][1] = bf2f(gp[H]); pre[i][2] = bf2f(gp[2 * H]);
      }
    }
    // ---- (1) gather h_{s-1} from the exchange buffer (step 0: already in LDS). A granule = 16 bytes
    //      {16-bit tag, 7 bf16}: pairs j0 .. j0 + 6 of workgroup c's slice (pair j = sample + BP * unit) ---
    if (s > 0) {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.xbuf + (size_t)((s - 1) & 1) * ngran * 2), 0, ngran * 16, 0x00020000);
      const unsigned want = (unsigned)s & 0xffffu;      // tag of h_{s-1} = (s - 1) + 1
      int polls = 0;
#pragma unroll 1
      for (int g0 = 0; g0 < ngran; g0 += kXcdGC * kXcdThreads) {     // ONE trip at H = 800, B = 16
        unsigned pending = 0;
#pragma unroll
        for (int i = 0; i < kXcdGC; ++i)
          if (g0 + tid + i * kXcdThreads < ngran) pending |= 1u << i;
        while (pending && !failed) {
          u32x4 g[kXcdGC];
#pragma unroll
          for (int i = 0; i < kXcdGC; ++i)
            if (pending & (1u << i))       // aux 16 = sc1: served by the XCD's L2, never by this CU's L1
              g[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (g0 + tid + i * kXcdThreads) * 16, 0, 16);
#pragma unroll
          for (int i = 0; i < kXcdGC; ++i)
            if ((pending & (1u << i)) && (g[i][0] & 0xffffu) == want) {
              const int gi = g0 + tid + i * kXcdThreads;
              const int c = (int)__umulhi((unsigned)gi, gpc_inv);      // producer workgroup
              const int j0 = (gi - c * gpc) * 7;                       // first pair of the granule
              char* const hb = h_l + c * upc * 2;
#pragma unroll
              for (int v = 0; v < 7; ++v) {
                const int j = j0 + v;                       // pair index: sample = j % BP, unit = j / BP
                const unsigned w16 = (g[i][(v + 1) >> 1] >> (16 * ((v + 1) & 1))) & 0xffffu;
                if (j < upc * BP)
                  *reinterpret_cast<bf16_t*>(hb + (j & (BP - 1)) * HS + (j / BP) * 2) = (bf16_t)w16;
              }
              pending &= ~(1u << i);
            }
          if (pending) {
            if (++polls > (1 << 17) ||
                ((polls & 63) == 0 && __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
              failed = 1;
            else
              __builtin_amdgcn_s_sleep(1);
          }
        }
      }
    }
    XCD_TICK(0);
    if (__syncthreads_or(failed)) {                  // uniform exit: nobody is left at a barrier
      if (tid == 0) __hip_atomic_store(a.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    XCD_TICK(1);
    // ---- (2) gates' recurrent part: rows of R (registers) x h (LDS), this wave's k-steps -----------
    f32x4 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const int col = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
      for (int ks = 0; ks < kXcdKS; ++ks) {
        const int kstep = wave + 8 * ks;
        if (kstep < NK) {
          bf16x8 bfr[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            bfr[nb] = *reinterpret_cast<const bf16x8*>(h_l + (nb * 16 + col) * HS + (kstep * 32 + kq) * 2);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[rt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][rt], bfr[nb], acc[rt][nb], 0, 0, 0);
        }
      }
      // C layout: col = lane & 15, rows 4 (lane >> 4) .. + 3 -> one 16-byte write per tile into the
      // [sample][row] image of this wave
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          *reinterpret_cast<f32x4*>(part + ((size_t)wave * BP + nb * 16 + col) * PR + rt * 16 + 4 * (lane >> 4)) = acc[rt][nb];
    }
    __syncthreads();
    XCD_TICK(2);
    // ---- (3) gate math for this workgroup's (unit, sample) pairs --------------------------------
#pragma unroll
    for (int i = 0; i < ME; ++i) {
      const int u = ub_i[i] & 255, b = ub_i[i] >> 8, e = b + BP * u;
      if (tid + i * kXcdThreads < upc * BP) {
        float hn_new = hst[e];
        if (tt[i] >= 0) {
          float rec[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            float sm = bias_i[i][g];
#pragma unroll
            for (int w = 0; w < 8; ++w) sm += part[((size_t)w * BP + b) * PR + g * upc + u];
            rec[g] = sm;
          }
          const float rg = sigm(pre[i][0] + rec[0]);
          const float zg = sigm(pre[i][1] + rec[1]);
          const float ng = tanh_fast_(pre[i][2] + rg * rec[2]);
          hn_new = (1.f - zg) * ng + zg * hn_new;
          const long long row = (long long)b * T + tt[i];
          if (p.gates) {
            bf16_t* gp = p.gates + row * (4 * H) + u0 + u;
            gp[0] = f2bf(rg); gp[H] = f2bf(zg); gp[2 * H] = f2bf(ng); gp[3 * H] = f2bf(rec[2]);
          }
          p.y[row * p.ldy + u0 + u] = f2bf(hn_new);
          hst[e] = hn_new;
        }
        hpub[e] = f2bf(hn_new);
      }
    }
    __syncthreads();
    XCD_TICK(3);
    // ---- (4) publish this workgroup's slice of h_s: gpc granules of 7 values, ONE 16-byte store each ---
    if (s + 1 < T) {
      char* xo = reinterpret_cast<char*>(p.xbuf) + ((size_t)(s & 1) * ngran + (size_t)cu * gpc) * 16;
      for (int e = tid; e < gpc; e += kXcdThreads) {
        const bf16_t* hp = hpub + e * 7;
        u32x4 g;
        g[0] = ((unsigned)(s + 1) & 0xffffu) | ((unsigned)hp[0] << 16);
#pragma unroll
        for (int v = 0; v < 3; ++v) g[1 + v] = (unsigned)hp[1 + 2 * v] | ((unsigned)hp[2 + 2 * v] << 16);
        st_plain_b128(xo + (size_t)e * 16, g);
      }
    }
  }
  for (int e = tid; e < upc * BP; e += kXcdThreads) {
    const int b = e / upc, u = e - b * upc;
    if (u < nu && b < B) p.h32[(long long)b * H + u0 + u] = hst[b + BP * u];
  }
#ifdef OS2S_GRU_XCD_TIMERS
  XCD_TICK(4);
  if (dir == 0 && cu == 5 && tid == 0)   // workgroup 5 of direction 0: cycles per step of each phase
    for (int i = 0; i < 5; ++i) a.flags[2 + i] = (int)(tph[i] / T);
#endif
}

// ---------------------------------------------------------------------------------------------
// Backward through time. dh_s = dy + carry + dg_{s+1} . Wh has a 3H-deep reduction; gathering dg
// [B, 3H] on every workgroup (88 KB per step at H = 800) made a first version of this kernel slower
// than the launch per step (10.96 vs 9.06 us). Here the product is cut along its REDUCTION instead: a
// workgroup keeps the gate gradients of its OWN upc units (it computed them itself one step earlier),
// multiplies them with its 3 * upc rows of Wh — held in registers as columns of Wh^T, [H x 96] — into a
// partial dh for ALL H units, and hands each other workgroup the [upc x B] block of that partial that
// belongs to its units (bf16, 7 values + tag per 16-byte granule). A step gathers 32 such blocks
// (30 KB, the forward pass's volume), sums them in a fixed order (no atomics) and does the gate
// derivatives of rnn.hip:rnn_step_bwd_kernel. B <= 16.
// ---------------------------------------------------------------------------------------------
constexpr int kXcdBT = 7;            // row tiles (16 units of dh) per wave: 8 x 7 x 16 = 896 >= H

struct GruXcdDirB {
  const bf16_t* whT;       // [H, 3H]
  const bf16_t* dy; long long lddy;
  const bf16_t* y; long long ldy;     // forward outputs (h_{t-1})
  const bf16_t* gates;     // [B, T, 4H] saved r, z, n, (R_n h + b_Rn)
  bf16_t* dgx;             // [B, T, 3H]
  bf16_t* dgr;             // [B, T, 3H] or null
  unsigned long long* xbuf;   // granules [2 slots][32 consumers][32 producers][gpp] x 16 bytes
  int reverse;
};
struct GruXcdArgsB {
  int B, T, H, ndir;
  const int32_t* lens;
  int* flags;
  GruXcdDirB d[2];
};

__global__ __launch_bounds__(kXcdThreads) void gru_xcd_bwd_kernel(GruXcdArgsB a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int dir, cu;
  {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dir = xcc & 7;
    if (dir >= a.ndir) return;
    int* const slot = reinterpret_cast<int*>(smem);
    if (tid == 0) *slot = __hip_atomic_fetch_add(a.flags + 8 + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    cu = *slot;
    __syncthreads();
    if (cu >= kXcdCus) {
      if (tid == 0) __hip_atomic_store(a.flags, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  const GruXcdDirB& p = a.d[dir];
  const int B = a.B, T = a.T, H = a.H, GH = 3 * H;
  constexpr int BP = 16;
  const int upc = (H + kXcdCus - 1) / kXcdCus;      // <= 32
  const int u0 = cu * upc;
  const int nu = max(0, min(upc, H - u0));
  const int nval = upc * BP;                        // values of one (producer, consumer) block
  const int gpp = (nval + 6) / 7;                   // granules of a block
  const int VP = gpp * 7 + 2;                       // pitch (bf16) of a block image in LDS
  const int ngran = kXcdCus * gpp;                  // granules a workgroup gathers / publishes per step
  const unsigned gpp_inv = (unsigned)((0x100000000ull + gpp - 1) / gpp);
  constexpr int DS = 96 * 2 + 16;                   // row stride (bytes) of the own gate gradients [BP][96]
  // LDS: stage [32 producers][VP] bf16 | out [32 consumers][VP] bf16 | dgo [BP][DS] | carry [upc][BP] f32
  bf16_t* const stage = reinterpret_cast<bf16_t*>(smem);
  bf16_t* const outi = stage + kXcdCus * VP;
  char* const dgo = reinterpret_cast<char*>(outi + kXcdCus * VP + 8);
  float* const carry = reinterpret_cast<float*>(dgo + BP * DS);

  // ---- Wh^T columns of this workgroup's gate rows -> registers: A[j][k'], k' = 32 gate + unit -------
  bf16x8 af[kXcdBT][3];
  {
    const int rrow = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
    for (int rt = 0; rt < kXcdBT; ++rt) {
      const int j = (wave + 8 * rt) * 16 + rrow;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {               // k-step = gate
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = kq + i;
          e[i] = (j < H && u < nu) ? p.whT[(long long)j * GH + (long long)ks * H + u0 + u] : (unsigned short)0;
        }
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)e[2 * i] | ((unsigned)e[2 * i + 1] << 16);
        af[rt][ks] = __builtin_bit_cast(bf16x8, w);
      }
    }
  }
  const int pb = tid / upc, pu = tid - pb * upc;     // this thread's (unit, sample) pair (upc * BP <= 512)
  const bool mine = tid < upc * BP && pu < nu && pb < B;
  const int len = mine ? (a.lens ? min(max(a.lens[pb], 0), T) : T) : -1;
  for (int i = tid; i < upc * BP; i += kXcdThreads) carry[i] = 0.f;
  for (int i = tid; i < 2 * kXcdCus * VP + 8; i += kXcdThreads) stage[i] = 0;
  for (int i = tid; i < BP * DS / 4; i += kXcdThreads) reinterpret_cast<uint32_t*>(dgo)[i] = 0u;
  __syncthreads();

  int failed = 0;
  for (int it = 0; it < T; ++it) {
    const int s = T - 1 - it;
    // ---- (0) operands of this thread's pair at loop step s ----------------------------------------
    int t = -1;
    float dyv = 0.f, sv0 = 0.f, sv1 = 0.f, sv2 = 0.f, sv3 = 0.f, hprev = 0.f;
    if (s < len) {
      t = p.reverse ? len - 1 - s : s;
      const long long row = (long long)pb * T + t;
      const int j = u0 + pu;
      dyv = bf2f(p.dy[row * p.lddy + j]);
      const bf16_t* gp = p.gates + row * (4 * H) + j;
      sv0 = bf2f(gp[0]); sv1 = bf2f(gp[H]); sv2 = bf2f(gp[2 * H]); sv3 = bf2f(gp[3 * H]);
      if (s > 0) hprev = bf2f(p.y[((long long)pb * T + (p.reverse ? t + 1 : t - 1)) * p.ldy + j]);
    }
    // ---- (1) gather the 32 partial blocks of dg_{s+1} . Wh for this workgroup's units ----------------
    if (it > 0) {
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.xbuf + ((size_t)((it - 1) & 1) * kXcdCus + cu) * ngran * 2), 0, ngran * 16, 0x00020000);
      const unsigned want = (unsigned)it & 0xffffu;
      int polls = 0;
#pragma unroll 1
      for (int g0 = 0; g0 < ngran; g0 += kXcdGC * kXcdThreads) {
        unsigned pending = 0;
#pragma unroll
        for (int i = 0; i < kXcdGC; ++i)
          if (g0 + tid + i * kXcdThreads < ngran) pending |= 1u << i;
        while (pending && !failed) {
          u32x4 g[kXcdGC];
#pragma unroll
          for (int i = 0; i < kXcdGC; ++i)
            if (pending & (1u << i))
              g[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (g0 + tid + i * kXcdThreads) * 16, 0, 16);
#pragma unroll
          for (int i = 0; i < kXcdGC; ++i)
            if ((pending & (1u << i)) && (g[i][0] & 0xffffu) == want) {
              const int gi = g0 + tid + i * kXcdThreads;
              const int pr = (int)__umulhi((unsigned)gi, gpp_inv);        // producer
              bf16_t* const dst = stage + pr * VP + (gi - pr * gpp) * 7;
#pragma unroll
              for (int v = 0; v < 7; ++v)
                dst[v] = (bf16_t)((g[i][(v + 1) >> 1] >> (16 * ((v + 1) & 1))) & 0xffffu);
              pending &= ~(1u << i);
            }
          if (pending) {
            if (++polls > (1 << 17) ||
                ((polls & 63) == 0 && __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
              failed = 1;
            else
              __builtin_amdgcn_s_sleep(1);
          }
        }
      }
    }
    if (__syncthreads_or(failed)) {
      if (tid == 0) __hip_atomic_store(a.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // ---- (2) gate derivatives of this thread's pair (rnn.hip: cuDNN GRU) ------------------------------
    if (tid < upc * BP) {
      float dr_ = 0.f, dz_ = 0.f, drn_ = 0.f;
      if (t >= 0) {
        float dh = dyv + carry[pb + BP * pu];
        if (it > 0) {
          const bf16_t* sp = stage + pu * BP + pb;            // value index = unit * BP + sample
#pragma unroll 8
          for (int q = 0; q < kXcdCus; ++q) dh += bf2f(sp[q * VP]);
        }
        const float rg = sv0, zg = sv1, ng = sv2, hn = sv3;
        const float dn = dh * (1.f - zg);
        const float dz = dh * (hprev - ng);
        const float dnpre = dn * (1.f - ng * ng);
        dr_ = dnpre * hn * rg * (1.f - rg);
        dz_ = dz * zg * (1.f - zg);
        drn_ = dnpre * rg;
        carry[pb + BP * pu] = dh * zg;
        const long long o = ((long long)pb * T + t) * GH + u0 + pu;
        p.dgx[o] = f2bf(dr_); p.dgx[o + H] = f2bf(dz_); p.dgx[o + 2 * H] = f2bf(dnpre);
        if (p.dgr) { p.dgr[o] = f2bf(dr_); p.dgr[o + H] = f2bf(dz_); p.dgr[o + 2 * H] = f2bf(drn_); }
      }
      // recurrent-side gradients of the own units, the B operand of the next product (zeros for a
      // sample past its length)
      bf16_t* dq = reinterpret_cast<bf16_t*>(dgo + pb * DS) + pu;
      dq[0] = f2bf(dr_); dq[32] = f2bf(dz_); dq[64] = f2bf(drn_);
    }
    __syncthreads();
    if (it + 1 < T) {
      // ---- (3) partial dh for ALL units: Wh^T[:, own rows] x dg_own -> bf16 image by consumer ------------
      {
        const int col = lane & 15, kq = (lane >> 4) * 8;
        bf16x8 bfr[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
          bfr[ks] = *reinterpret_cast<const bf16x8*>(dgo + col * DS + (ks * 32 + kq) * 2);
#pragma unroll
        for (int rt = 0; rt < kXcdBT; ++rt) {
          const int j0 = (wave + 8 * rt) * 16;
          if (j0 < H) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][ks], bfr[ks], acc, 0, 0, 0);
            // C layout: sample = lane & 15, units j0 + 4 (lane >> 4) + i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int j = j0 + 4 * (lane >> 4) + i;
              if (j < H) {
                const int cc = j / upc;
                outi[cc * VP + (j - cc * upc) * BP + col] = f2bf(acc[i]);
              }
            }
          }
        }
      }
      __syncthreads();
      // ---- (4) publish: block (consumer cc, producer cu) = gpp granules ------------------------------------
      {
        char* const xo = reinterpret_cast<char*>(p.xbuf) + (size_t)(it & 1) * kXcdCus * ngran * 16;
        for (int e = tid; e < ngran; e += kXcdThreads) {
          const int cc = (int)__umulhi((unsigned)e, gpp_inv), ge = e - cc * gpp;
          const bf16_t* hp = outi + cc * VP + ge * 7;
          u32x4 g;
          g[0] = ((unsigned)(it + 1) & 0xffffu) | ((unsigned)hp[0] << 16);
#pragma unroll
          for (int v = 0; v < 3; ++v) g[1 + v] = (unsigned)hp[1 + 2 * v] | ((unsigned)hp[2 + 2 * v] << 16);
          st_plain_b128(xo + ((size_t)cc * ngran + (size_t)cu * gpp + ge) * 16, g);
        }
      }
    }
  }
}

}  // namespace os2s

using namespace os2s;

static int g_gru_xcd_mode = -1;    // -1 = environment (OS2S_GRU_XCD, default on), 0 = off, 1 = on
static int g_gru_xcd_force_abort = 0;
// 2 (test hook) = on, and the NEXT persistent forward launch starts with its abort flag set: every workgroup
// leaves at its first wait, the outputs are garbage and the sticky word reports a timeout — what a launch that
// does not get its 32 co-resident workgroups per XCD does (tests/test_ds2_gpu.py: the step is redone)
extern "C" void os2s_gru_xcd_set_mode(int m) {
  g_gru_xcd_force_abort = m == 2;
  g_gru_xcd_mode = m == 2 ? 1 : m;
}

static bool gru_xcd_enabled() {
  if (g_gru_xcd_mode >= 0) return g_gru_xcd_mode != 0;
  static const int v = [] { const char* e = getenv("OS2S_GRU_XCD"); return e ? atoi(e) : 1; }();
  return v != 0;
}

// bytes of the exchange buffers + flags per direction (appended to the step path's workspace)
extern "C" size_t os2s_gru_xcd_workspace_bytes(int B, int H) {
  const int BP = B <= 16 ? 16 : 32, upc = (H + 31) / 32, gpc = (upc * BP + 6) / 7;
  return (size_t)2 * 32 * gpc * 16 + 256;
}

static size_t gru_xcd_lds_bytes(int B, int H) {
  const int BP = B <= 16 ? 16 : 32, upc = (H + kXcdCus - 1) / kXcdCus, NK = (H + 31) / 32;
  const int RT = 3 * upc <= 80 ? 5 : 6;
  return (size_t)BP * (NK * 64 + 16) + (size_t)8 * BP * (RT * 16 + 4) * 4 + (size_t)upc * BP * 4 +
         (size_t)(((upc * BP + 6) / 7) * 7 + 1) * 2 + 64;
}

// true when the persistent kernel covers this layer (the caller uses the per-step launches otherwise)
bool gru_xcd_supported(int B, int T, int H, int ndir) {
  if (!gru_xcd_enabled() || B < 1 || B > 32 || H % 8 != 0 || H > 1024 || H < 32 || T < 2 || ndir > 2) return false;
  const int BP = B <= 16 ? 16 : 32, upc = (H + kXcdCus - 1) / kXcdCus;
  if (3 * upc > kXcdRTMax * 16 || T > 65000 || kXcdCus * ((upc * BP + 6) / 7) >= 65536) return false;
  if (gru_xcd_lds_bytes(B, H) > 160 * 1024) return false;   // (B = 32 with H = 1024: the partial sums do not fit)
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n != 256)
    return false;                                        // 8 XCDs x 32 CUs: the placement the kernel checks
  return true;
}

// ---- sticky abort word -------------------------------------------------------------------------------
// A launch that gives up (poll timeout / placement mismatch) leaves partially written outputs. Its abort
// code is latched by a one-thread kernel behind EVERY persistent launch (forward and backward) into a pinned,
// device-mapped host word that no launch ever clears: the next persistent launch of either direction and
// os2s_gru_xcd_status() (called by the host layer at its per-step sync point) fail loudly.
static int* g_sticky_host = nullptr;
static int* g_sticky_dev = nullptr;

static bool gru_xcd_sticky_init() {
  if (g_sticky_host) return true;
  int* h = nullptr;
  if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess) return false;
  *h = 0;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { hipHostFree(h); return false; }
  g_sticky_dev = (int*)d;
  g_sticky_host = h;
  return true;
}

__global__ void gru_xcd_latch_kernel(const int* __restrict__ flags, int* __restrict__ sticky) {
  const int f = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (f != 0) __hip_atomic_fetch_or(sticky, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int gru_xcd_latch(hipStream_t stream, const int* flags) {
  if (!gru_xcd_sticky_init()) return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH(gru_xcd_latch_kernel, dim3(1), dim3(1), 0, stream, flags, g_sticky_dev);
  return OS2S_OK;
}

// Launches do NOT fail on a set word (they did until round 4): in a data-parallel job every rank has to
// enqueue the same sequence of collectives whether or not one of its recurrent launches gave up, so the step
// runs to its end everywhere and the host layer decides THERE (Model.train_step: status word, agreed over
// the ranks, then the step is redone on the launch-per-step kernels).
static long long g_gru_xcd_launches = 0;
static int gru_xcd_check_sticky() {
  if (!gru_xcd_sticky_init()) return OS2S_ERR_LAUNCH;
  ++g_gru_xcd_launches;
  return OS2S_OK;
}
// persistent GRU launches (forward + backward) enqueued by this process so far: the host layer reads the
// abort word only after steps that ran some
extern "C" long long os2s_gru_xcd_launch_count(void) { return g_gru_xcd_launches; }

// 0 = no persistent GRU launch has given up so far (as far as the host can see without synchronising: call
// it after a stream synchronisation for a definite answer); otherwise the OR of the abort codes (1 = poll
// timeout, 2 = placement mismatch). clear != 0 resets the word (after the caller has discarded the step).
extern "C" int os2s_gru_xcd_status(int clear) {
  if (!g_sticky_host) return 0;            // no persistent launch has run in this process
  const int f = *(volatile int*)g_sticky_host;
  if (clear) *(volatile int*)g_sticky_host = 0;
  return f;
}

// one launch for all T steps of ndir directions. gx/wh/bh/y/gates/reverse per direction as in
// os2s_rnn_dir_fwd_t; h32[d] = fp32 state [B, H] (in: initial, out: final); xws[d] = exchange buffer of
// os2s_gru_xcd_workspace_bytes() bytes (zeroed here); flags = 64 zeroed bytes shared by the launch.
int launch_gru_xcd_fwd(hipStream_t stream, int ndir, const os2s_rnn_dir_fwd_t* dirs, float* const* h32,
                       void* const* xws, int* flags, const int32_t* lens, int B, int T, int H) {
  if (int rc = gru_xcd_check_sticky()) return rc;
  GruXcdArgs a;
  a.B = B; a.T = T; a.H = H; a.ndir = ndir; a.lens = lens; a.flags = flags;
  const int BP = B <= 16 ? 16 : 32;
  for (int d = 0; d < ndir; ++d) {
    const os2s_rnn_dir_fwd_t& s = dirs[d];
    GruXcdDir& k = a.d[d];
    k.gx = (const bf16_t*)s.gx; k.wh = (const bf16_t*)s.wh; k.bh = s.bh; k.h32 = h32[d];
    k.y = (bf16_t*)s.y; k.ldy = s.ldy; k.gates = (bf16_t*)s.gates; k.reverse = s.reverse;
    k.xbuf = (unsigned long long*)xws[d];
    if (hipMemsetAsync(xws[d], 0, os2s_gru_xcd_workspace_bytes(B, H), stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  }
  if (ndir == 1) a.d[1] = a.d[0];
  if (hipMemsetAsync(flags, 0, 64, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  if (g_gru_xcd_force_abort) {
    g_gru_xcd_force_abort = 0;
    if (hipMemsetAsync(flags, 1, 1, stream) != hipSuccess) return OS2S_ERR_LAUNCH;     // flags[0] = 1 (timeout)
  }
  const int upc = (H + kXcdCus - 1) / kXcdCus;
  const int RT = 3 * upc <= 80 ? 5 : 6;
  const size_t lds = gru_xcd_lds_bytes(B, H);
  if (lds > 160 * 1024) return OS2S_ERR_UNSUPPORTED;
  const void* fn = BP == 16 ? (RT == 5 ? (const void*)gru_xcd_fwd_kernel<1, 5> : (const void*)gru_xcd_fwd_kernel<1, 6>)
                            : (RT == 5 ? (const void*)gru_xcd_fwd_kernel<2, 5> : (const void*)gru_xcd_fwd_kernel<2, 6>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  const dim3 grid(8 * kXcdCus), blk(kXcdThreads);
  if (BP == 16 && RT == 5) { OS2S_LAUNCH((gru_xcd_fwd_kernel<1, 5>), grid, blk, lds, stream, a); }
  else if (BP == 16) { OS2S_LAUNCH((gru_xcd_fwd_kernel<1, 6>), grid, blk, lds, stream, a); }
  else if (RT == 5) { OS2S_LAUNCH((gru_xcd_fwd_kernel<2, 5>), grid, blk, lds, stream, a); }
  else { OS2S_LAUNCH((gru_xcd_fwd_kernel<2, 6>), grid, blk, lds, stream, a); }
  if (int rc = gru_xcd_latch(stream, flags)) return rc;
#ifdef OS2S_GRU_XCD_TIMERS
  {
    int hf[8];
    hipStreamSynchronize(stream);
    hipMemcpy(hf, flags, sizeof(hf), hipMemcpyDeviceToHost);
    fprintf(stderr, "gru_xcd cycles/step: gather %d  barrier %d  mfma+partials %d  gates %d  publish(+pre) %d (last column: tail)\n",
            hf[2], hf[3], hf[4], hf[5], hf[6]);
  }
#endif
  return OS2S_OK;
}


extern "C" size_t os2s_gru_xcd_bwd_workspace_bytes(int B, int H) {
  const int upc = (H + 31) / 32, gpp = (upc * 16 + 6) / 7;
  (void)B;
  return (size_t)2 * 32 * 32 * gpp * 16 + 256;
}

static size_t gru_xcd_bwd_lds_bytes(int H) {
  const int upc = (H + kXcdCus - 1) / kXcdCus, gpp = (upc * 16 + 6) / 7, VP = gpp * 7 + 2;
  return (size_t)2 * kXcdCus * VP * 2 + 16 + (size_t)16 * (96 * 2 + 16) + (size_t)upc * 16 * 4 + 64;
}

bool gru_xcd_bwd_supported(int B, int T, int H, int ndir) {
  static const int on = [] { const char* e = getenv("OS2S_GRU_XCD_BWD"); return e ? atoi(e) : 1; }();
  if (!on) return false;
  if (!gru_xcd_supported(B, T, H, ndir) || B > 16) return false;
  const int upc = (H + kXcdCus - 1) / kXcdCus;
  if (upc > 32 || upc * 16 > kXcdThreads || H > 8 * kXcdBT * 16) return false;
  return gru_xcd_bwd_lds_bytes(H) <= 160 * 1024;
}

int launch_gru_xcd_bwd(hipStream_t stream, int ndir, const os2s_rnn_dir_bwd_t* dirs, void* const* xws, int* flags,
                       const int32_t* lens, int B, int T, int H) {
  if (int rc = gru_xcd_check_sticky()) return rc;
  GruXcdArgsB a;
  a.B = B; a.T = T; a.H = H; a.ndir = ndir; a.lens = lens; a.flags = flags;
  for (int d = 0; d < ndir; ++d) {
    const os2s_rnn_dir_bwd_t& s = dirs[d];
    GruXcdDirB& k = a.d[d];
    k.whT = (const bf16_t*)s.whT; k.dy = (const bf16_t*)s.dy; k.lddy = s.lddy; k.y = (const bf16_t*)s.y; k.ldy = s.ldy;
    k.gates = (const bf16_t*)s.gates; k.dgx = (bf16_t*)s.dgx; k.dgr = (bf16_t*)s.dgr; k.reverse = s.reverse;
    k.xbuf = (unsigned long long*)xws[d];
    if (hipMemsetAsync(xws[d], 0, os2s_gru_xcd_bwd_workspace_bytes(B, H), stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  }
  if (ndir == 1) a.d[1] = a.d[0];
  if (hipMemsetAsync(flags, 0, 64, stream) != hipSuccess) return OS2S_ERR_LAUNCH;
  const size_t lds = gru_xcd_bwd_lds_bytes(H);
  if (hipFuncSetAttribute((const void*)gru_xcd_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return OS2S_ERR_LAUNCH;
  OS2S_LAUNCH(gru_xcd_bwd_kernel, dim3(8 * kXcdCus), dim3(kXcdThreads), lds, stream, a);
  return gru_xcd_latch(stream, flags);
}
