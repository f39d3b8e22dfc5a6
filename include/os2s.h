/*
 * os2s.h — C ABI of libos2s_hip.so: the MI355X (gfx950) kernels behind
 * OpenSeq2Seq's Encoder / Decoder / Loss / optimizer hot path.
 *
 * Conventions (all entry points):
 *   - plain C types only: device pointers, sizes, a stream handle
 *     (os2s_stream_t == hipStream_t, passed as void*; NULL = default stream);
 *   - the caller owns every buffer (PyTorch tensors in the Python host layer);
 *     no hidden allocation, workspaces are passed in and sized by
 *     os2s_*_workspace_bytes();
 *   - returns OS2S_OK (0) or a negative OS2S_ERR_* code; no exceptions cross
 *     the boundary; kernels are enqueued asynchronously on `stream`;
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits);
 *   - activations are channels-last: [B, T, C] with C contiguous, the layout
 *     the reference's data layers / encoders use
 *     (open_seq2seq/encoders/tdnn_encoder.py:160-164, "B T F").
 *
 * Each declaration cites the reference call site it replaces
 * (paths relative to NVIDIA/OpenSeq2Seq).
 */
#ifndef OS2S_H_
#define OS2S_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* os2s_stream_t;

enum {
  OS2S_OK = 0,
  OS2S_ERR_INVALID_ARG = -1,
  OS2S_ERR_LAUNCH = -2,
  OS2S_ERR_UNSUPPORTED = -3,
  OS2S_ERR_WORKSPACE = -4
};

/* Library identification: returns the ABI version (bumped on any signature change). */
int os2s_abi_version(void);
/* Human-readable description of an error code. */
const char* os2s_strerror(int code);
/* Detail of the last failed kernel launch on this thread (HIP error string). */
const char* os2s_last_error_detail(void);
/* Deterministic mode (default: environment OS2S_DETERMINISTIC, else off): a debugging aid. The kernels that
 * accumulate parameter gradients with fp32 atomics from several workgroups (narrow / K = 1 / stride-2 conv
 * and depthwise weight gradients, the embedding gradient, the style-token attention gradient) are
 * launched in a single-contributor geometry instead: slower, bit-identical run to run. */
int os2s_deterministic(void);
void os2s_set_deterministic(int on);

/* ------------------------------------------------------------------------
 * CTC greedy (best-path) decode.
 * Replaces tf.nn.ctc_greedy_decoder as called by decode_without_lm
 * (open_seq2seq/decoders/fc_decoders.py:244-251) and mirrors
 * decoders/ctc_greedy_decoder.cpp:4-45.
 *   logits      [T, B, V] fp32, time-major (fc_decoders.py:147-148)
 *   seq_len     [B] int32 valid frames per sample
 *   blank       blank id (V-1 in the reference, speech2text.py:123-125)
 *   merge_repeated  collapse repeats before dropping blanks (1 in the reference)
 * Outputs (dense form of the reference's SparseTensor):
 *   out_ids     [B, T] int32, first out_len[b] entries valid, rest = -1
 *   out_len     [B] int32
 *   neg_sum_logits [B] fp32 = -sum_t max_v logits[t,b,v]  (may be NULL)
 *   workspace   device scratch of os2s_ctc_greedy_decode_workspace_bytes(T,B)
 * Ties in the argmax resolve to the lowest class index (first maximum).
 * ---------------------------------------------------------------------- */
size_t os2s_ctc_greedy_decode_workspace_bytes(int T, int B);
int os2s_ctc_greedy_decode(os2s_stream_t stream, const float* logits,
                           const int32_t* seq_len, int T, int B, int V,
                           int blank, int merge_repeated, int32_t* out_ids,
                           int32_t* out_len, float* neg_sum_logits,
                           void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * 1-D convolution as implicit GEMM on the matrix cores (bf16 in, fp32 accumulate).
 * Replaces tf.layers.conv1d(use_bias=False) of conv_bn_actv / conv_bn_res_bn_actv
 * (open_seq2seq/parts/cnns/conv_blocks.py:195-206, 78-85, 118-129) including the
 * sequence mask the TDNN encoder multiplies onto every conv INPUT
 * (open_seq2seq/encoders/tdnn_encoder.py:185-186, 204-205); with K = 1 it is the
 * dense layer of FullyConnectedTimeDecoder (decoders/fc_decoders.py:135-148).
 *
 *   y[b,t,co] = sum_k sum_ci x[b, t*stride + k*dil - padL, ci] * w[k][co][ci] (+ bias[co])
 *
 *   x      [B, Tin, Cin] bf16, channels-last; rows t >= in_len[b] read as zero
 *          when in_len != NULL (fused mask), rows outside [0,Tin) are padding.
 *   w      [K, Cout, Cin] bf16 (Cin contiguous). NOTE: the reference (TF) stores
 *          [K, Cin, Cout]; the host layer transposes on import/export.
 *   y      bf16 (out_f32 = 0) or fp32 (out_f32 = 1), element strides
 *          y_stride_b / y_stride_t (channel stride 1) — e.g. time-major logits
 *          [T, B, V] use y_stride_b = V, y_stride_t = B*V.
 *   padL   left padding; TF "SAME": total = max((ceil(Tin/stride)-1)*stride +
 *          (K-1)*dil + 1 - Tin, 0), padL = total/2 (extra pad goes right).
 *   stats  optional fp32 [os2s_conv1d_num_mtiles(B,Tout), 2, Cout]: per-tile
 *          per-channel (sum, sum of squares) of the bf16-rounded outputs over the
 *          tile's valid rows — the partial sums BatchNorm needs (K5 fused).
 *   accumulate  1: y += result (used to sum data-gradients of several consumers).
 * The data-gradient of a stride-1 conv is this same entry point applied to dY
 * with the tap-flipped, transposed weight copy wT[k'][ci][co] = w[K-1-k'][co][ci]
 * and padL' = (K-1)*dil - padL.
 * Requirements: Cin % 8 == 0; for bf16 output Cout % 8 == 0 and strides % 8 == 0.
 * ---------------------------------------------------------------------- */
int os2s_conv1d_num_mtiles(int B, int Tout);
/* Same with a fused epilogue: y = residual + dropout(act(conv + bias)); act 0 none / 1 relu;
 * dropout uses the (seed, element_index/8) hash shared by all os2s kernels; residual has
 * y's layout (bf16 output only). Covers FeedFowardNetwork's Dense+ReLU+dropout
 * (parts/transformer/ffn_layer.py:51-85) and PrePostProcessingWrapper's dropout + residual
 * (parts/transformer/common.py:99-106). */
int os2s_conv1d_fwd_ex(os2s_stream_t stream, const uint16_t* x, const uint16_t* w, void* y,
                       const int32_t* in_len, const float* bias, float* stats, int B,
                       int Tin, int Cin, int Cout, int K, int stride, int dil, int padL,
                       int Tout, long long y_stride_b, long long y_stride_t, int out_f32,
                       int accumulate, int act, float keep_prob, unsigned long long seed,
                       const uint16_t* residual, const int32_t* out_len);
/* Ragged batches: windows (128 output rows of one sample) whose whole input window lies past
 * in_len[b] cost no matrix work (their output is the exact zero tile, their BatchNorm partials
 * are zero); with out_len != NULL output windows that start at t >= out_len[b] are not computed
 * or stored at all (data gradients: the consumer masks those rows,
 * encoders/tdnn_encoder.py:185-186,204-205). os2s_conv1d_wgrad likewise visits only the
 * (sample, 64-row) chunks whose X window starts before in_len[b].
 *
 * os2s_conv1d_fwd_ws: the same call with a caller-owned workspace of
 * os2s_conv1d_workspace_bytes() bytes. The workspace lets the library split the last partial
 * round of workgroups of a launch over the input channels (fp32 partial tiles, deterministic
 * reduction by the last arriver); its first 4096 bytes are int32 tickets that must be ZERO
 * before the first use (the library leaves them zero). One workspace per stream: launches that
 * may overlap must not share one. Without a workspace (os2s_conv1d_fwd_ex / os2s_conv1d_fwd)
 * results are identical up to fp32 summation order, only the load balance differs.
 * The tile choice is a fixed function of the problem shape; no timing, no hidden state. */
/* Data gradient of a convolution whose input is the output of a conv + BatchNorm + ReLU (+ dropout)
 * layer (parts/cnns/conv_blocks.py:170-232), as the LAST contribution to that output's gradient, with
 * the activation backward and the partial sums of that layer's BatchNorm backward in the epilogue:
 *   dx (+)= conv(dy, wT) [stride 1];  dz = (mask_ref > 0) ? dx * mask_scale : 0, written to dx;
 *   stats[window, 0, c] = sum_rows dz,  stats[window, 1, c] = sum_rows dz * stat_ref   (ZERO on entry)
 * mask_ref = the layer's saved output [B, Tout, Cout], stat_ref = its convolution output (same layout),
 * mask_scale = 1 / keep_prob; wT = the tap-flipped transposed weights [K, Cout, Cin]; out_len as in
 * os2s_conv1d_fwd_ws. Replaces the reduction pass os2s_bn_act_bwd_reduce for single-input BatchNorm
 * layers; os2s_bn_bwd_finalize_raw turns the partials into dgamma / dbeta / c1 / c2. */
int os2s_conv1d_dgrad_bnact_ws(os2s_stream_t stream, const uint16_t* dy, const uint16_t* wT, void* dx,
                               float* stats, int B, int Tin, int Cin, int Cout, int K, int dil, int padL,
                               int Tout, int accumulate, const int32_t* out_len, const uint16_t* mask_ref,
                               float mask_scale, const uint16_t* stat_ref, void* workspace,
                               size_t workspace_bytes);
int os2s_bn_bwd_finalize_raw(os2s_stream_t stream, const float* partial, int nparts, int C, long long count,
                             const float* mean, const float* rstd, float* dgamma, float* dbeta,
                             int accumulate, float* c1, float* c2);
size_t os2s_conv1d_workspace_bytes(void);
int os2s_conv1d_fwd_ws(os2s_stream_t stream, const uint16_t* x, const uint16_t* w, void* y,
                       const int32_t* in_len, const float* bias, float* stats, int B,
                       int Tin, int Cin, int Cout, int K, int stride, int dil, int padL,
                       int Tout, long long y_stride_b, long long y_stride_t, int out_f32,
                       int accumulate, int act, float keep_prob, unsigned long long seed,
                       const uint16_t* residual, const int32_t* out_len, void* workspace,
                       size_t workspace_bytes);
/* Optional hint: a HOST copy of the int32 sequence lengths that the forward launches which follow receive
 * as in_len (lens == NULL or B <= 0 withdraws it). The tile of a ping-pong launch depends on the number of live
 * 128-row windows of the ragged batch; without the hint that number is only known on the device, both
 * ping-pong kernels are enqueued and the one not chosen exits (~8 us per launch); with it the same cost
 * model is evaluated at launch time and one kernel is enqueued. The hint only selects among tiles that are
 * all exact for any data: lengths that differ from the device's cost speed, never results. Process-wide
 * state, like the variant hook: set it around the launches of one batch (the encoder does). Returns OS2S_OK. */
int os2s_conv1d_set_host_lens(const int32_t* lens, int B);
/* Named test / measurement options — the ONE entry point for every tuning knob and forced-variant switch of
 * the library (nothing in the library reads the environment for them; values are process-wide and only
 * select among launch geometries that are all exact for any data):
 *   conv1d.variant: force a convolution tile. -1 (default) = by shape; 0 = 128x128 tile, X window
 *     double-buffered; 3 = 128x128, X window single-buffered when K >= 8 (3 workgroups per CU); 5 = 256x256
 *     lockstep tile; 10 = ping-pong kernels (balanced over live windows), tile chosen on the device from the
 *     live-window count: 2 windows x 256 columns, or 2 / 3 windows x 128 columns; 12 / 13 / 14 = ping-pong
 *     with the 2 x 128 / 3 x 128 / 2 x 256 tile forced
 *   conv1d.split: f > 0 forces the tail split factor of the ping-pong kernel (-1 = cost model)
 *   conv1d.pp_cost_256, conv1d.pp_cost_2x128, conv1d.pp_cost_3x128: fitted microseconds per 64-deep step
 *     of the three ping-pong convolution tiles — the constants of the device-side tile choice; a cost
 *     >= 1e6 removes a narrow tile from the candidates
 *   conv1d.pp_dgrad_penalty: factor on the narrow tiles' cost in data-gradient launches (out_len given:
 *     they share the chip with the weight-gradient stream)
 *   conv1d.pp_prio: 1 = the loading wave of a narrow-tile slot runs at s_setprio 2
 *   conv1d.pp_min_cout: narrowest layer (output channels) the ping-pong kernels take (default 320)
 *   conv1x1.variant: the 1x1 launches (os2s_conv1x1_fwd_grouped and K = 1 layers): 0 and 1 = lockstep 128x128
 *     tile; 2 = 256x256 ping-pong tile over the live windows whenever its envelope allows (Cin % 64 == 0,
 *     B <= 64); -1 (default) = the ping-pong tile for a single K = 1 layer of >= conv1d.pp_min_cout output
 *     channels (the lockstep tile's life there is its steps of exposed load latency: 512 -> 512 channels 33.7 ->
 *     26.3 us, QuartzNet step -0.8 ms), the lockstep tile for the grouped launches (a grouped 1x1 unit is 4 - 12
 *     steps of matrix work behind 128 KB of output: slower on the Jasper shapes)
 *   conv1x1.order: os2s_conv1x1_fwd_grouped's workgroup order: 1 (default) = the column tiles of one row tile run
 *     behind the same XCD at the same time (the activations are fetched from HBM once, not once per column tile;
 *     a 1x1 weight matrix fits every L2), 0 = all row tiles of a column tile first (rounds 1 - 5)
 *   conv1d_wgrad.variant: 0 = lockstep kernel, 1 = ping-pong, 2 = K = 1 ping-pong, 3 = one wave per SIMD with 16
 *     accumulator blocks per wave and a hand-written instruction stream (stride 1, K >= 2, dilation <= 5, channel
 *     counts multiples of 128; opt-in: measured slower than the ping-pong kernel), -1 = by shape;
 *   conv1d_wgrad.xcd_order: 1 (default) = consecutive ranks of the ping-pong / one-wave weight-gradient kernels —
 *     the tap quads or ci tiles of one output tile — run behind one XCD's L2; 0 = rank = blockIdx.x (rounds 2 - 5)
 *   conv1d_wgrad.sw_ablate: read only by measurement builds (-DOS2S_SW_ABLATE, tools/sw_ablate.py): the one-wave
 *     stream with an ingredient removed (1 LDS-DMA, 2 barrier, 4 transpose reads, 8 MFMAs; results are wrong then)
 *   conv1d_wgrad.split: > 0 forces the reduction split factor of the ping-pong kernels (-1 = cost model)
 *   gemm_nt.split: f > 0 forces the tail split factor of os2s_gemm_nt*, 0 disables the split, < 0 = cost model
 *   gemm_nt.tile: 0 (default) = by shape, 256 = always the 256 x 256 tile, 160 = the 160-row x 256-column tile
 *     (eight waves over the columns; bit-identical results) whenever it is legal (no per-window statistics)
 *   depthwise.variant: 0 = the generic depthwise kernels only, 1 = generic + register-window kernels (rounds 4 - 5),
 *     < 0 = by shape (stride 1, dilation 1, K <= 96: the matrix-core kernels of round 6)
 *   depthwise.ablate: measurement only (scratch/bench_depthwise.py): bit mask that switches parts of the matrix-core
 *     depthwise kernels off (1 compute, 2 output / diagonal sums, 4 loads; results are wrong then)
 *   bn.act_fwd.groups, bn.act_fwd.rows, bn.act_bwd_reduce.groups, bn.act_bwd_reduce.rows, bn.bwd_apply.groups,
 *     bn.bwd_apply.rows: tiling of the three BatchNorm kernels (8-channel groups per workgroup 8 .. 256, rows
 *     per workgroup 8 .. 1024; tools/bench_bn_sweep.py); the reduce tiling also sets what
 *     os2s_bn_act_bwd_num_parts returns
 * Returns 0, or -1 for an unknown name. os2s_option_name(i) enumerates the registered names (NULL past the
 * last one). */
int os2s_set_option(const char* name, double value);
const char* os2s_option_name(int index);
/* Profiling aid (tools/pp_timeline.py, ppn_timeline.py, conv1x1_phases.py): a device buffer the instrumented
 * kernel `kernel` ("conv1d", "conv1d_wgrad") writes per-slot time stamps into (NULL withdraws it); `mode` is
 * the kernel's own timing-experiment bit mask. Returns 0, or -1 for an unknown kernel name. */
int os2s_set_debug_stamps(const char* kernel, void* stamps, int mode);
/* Shader-clock probe (bench.py `roofline.shader_clock_mhz`): enqueues, on a private non-blocking stream, ONE wave
 * that spins for `spin_cycles` ticks of the shader clock counter and writes {shader cycles, ticks of the constant
 * 100 MHz reference counter} to the device buffer `out_u64x2`; os2s_clock_probe_wait() blocks until it has
 * finished. Run next to the work being measured it reports the clock the chip sustains under that load — the
 * factor between the 2.5 PFLOP/s peak (quoted at 2.4 GHz) and what the matrix pipes can deliver there. No
 * reference counterpart (measurement aid of SURVEY 8d). */
int os2s_clock_probe(void* out_u64x2, unsigned long long spin_cycles);
int os2s_clock_probe_wait(void);
/* Up to 16 independent 1x1 convolutions over the same batch geometry (B, T, lengths) in ONE
 * launch: y_i[b,t,:] (+)= x_i[b,t,:] . w_i^T, bf16 out, optional BatchNorm partials per group
 * (layout as os2s_conv1d_fwd). Replaces the dense-residual branches of conv_bn_res_bn_actv
 * (parts/cnns/conv_blocks.py:78-85: tf.layers.conv1d(kernel_size=1) per residual input, up to 10
 * per Jasper block) and, with the transposed weights and out_len, their data gradients.
 * `groups` is a HOST array (copied into the launch). */
typedef struct {
  const uint16_t* x;   /* [B, T, Cin]  bf16 */
  const uint16_t* w;   /* [1, Cout, Cin] bf16 */
  void* y;             /* [B, T, Cout] bf16 */
  float* stats;        /* [os2s_conv1d_num_mtiles(B,T), 2, Cout] or NULL */
  int Cin, Cout, accumulate;
} os2s_conv_group_t;
int os2s_conv1x1_fwd_grouped(os2s_stream_t stream, const os2s_conv_group_t* groups, int ngroups,
                             const int32_t* in_len, const int32_t* out_len, int B, int T);
/* The weight gradients of the same branches in ONE launch: for up to 16 groups over one ragged
 * batch (B, T, in_len), dw_i[co][ci] += sum_(b,t) dy_i[b,t,co] * x_i[b,t,ci] (fp32, accumulated:
 * the gradient of tf.layers.conv1d(kernel_size=1) w.r.t. its kernel, parts/cnns/conv_blocks.py:78-85,
 * once per residual input). One launch per branch held 12-36 output tiles and cut the reduction
 * 14-40 ways with fp32 atomics; a block's branches together are 50-216 tiles. `groups` is a HOST
 * array (copied into the launch); x rows may be a channel slice (x_row_stride >= Cin). */
typedef struct {
  const uint16_t* x;        /* [B, T, Cin]  bf16, row stride x_row_stride elements */
  const uint16_t* dy;       /* [B, T, Cout] bf16 */
  float* dw;                /* [1, Cout, Cin] fp32, accumulated into */
  long long x_row_stride;
  int Cin, Cout;
} os2s_wgrad_group_t;
int os2s_conv1x1_wgrad_grouped(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                               const int32_t* in_len, int B, int T);
/* The kernel gradients of up to 8 convolution layers of ONE shape (Cin, Cout, K, stride, dilation, padding) over
 * one ragged batch in one launch: dw_i[K, Cout, Cin] (+)= the weight gradient os2s_conv1d_wgrad_ws computes for
 * (x_i, dy_i). The `repeat` identical tf.layers.conv1d layers of a Jasper block (parts/cnns/conv_blocks.py:61-168,
 * encoders/tdnn_encoder.py:150-180) are 12 - 150 units of work each on 256 CUs; ranked together they fill the chip
 * without cutting every unit's reduction. Deterministic (the ping-pong kernel's one-owner reduction); shapes that
 * kernel does not take are launched one by one. Workspace: as os2s_conv1d_wgrad_ws. */
typedef struct {
  const uint16_t* x;        /* [B, Tin, Cin] bf16, row stride x_row_stride elements (the same for every group) */
  const uint16_t* dy;       /* [B, Tout, Cout] bf16 */
  float* dw;                /* [K, Cout, Cin] fp32 */
  long long x_row_stride;
} os2s_cwgrad_group_t;
int os2s_conv1d_wgrad_grouped_ws(os2s_stream_t stream, const os2s_cwgrad_group_t* groups, int ngroups,
                                 const int32_t* in_len, int B, int Tin, int Cin, int Cout, int K,
                                 int stride, int dil, int padL, int Tout, int accumulate,
                                 void* workspace, size_t workspace_bytes);
/* The same with the conv workspace (os2s_conv1d_workspace_bytes, one per stream): when every group is at
 * least 128 x 128 channels and the batch has >= 2048 rows the launch runs on the K = 1 ping-pong TN-GEMM
 * kernel — 256 x 256 tiles, reduction over the live 64-row chunks only, the tail of the launch cut along
 * the reduction and summed by the last arriver: no atomics, bit-identical run to run — otherwise it is
 * os2s_conv1x1_wgrad_grouped. */
int os2s_conv1x1_wgrad_grouped_ws(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                                  const int32_t* in_len, int B, int T, void* workspace,
                                  size_t workspace_bytes);
/* The weight gradients of up to 16 Dense layers over the same M rows (a packed token batch) in one
 * launch of the K = 1 ping-pong kernel: dw_i[Cout_i, Cin_i] (+)= dy_i[M, Cout_i]^T x_i[M, Cin_i], fp32,
 * deterministic (no atomics). tf.layers.Dense kernels of the Transformer whose outputs are too small
 * to fill the chip alone (1024 x 1024: attention_layer.py:54-62, 219). Cin, Cout >= 128. Workspace:
 * os2s_conv1d_workspace_bytes(), the contract of os2s_conv1d_fwd_ws. */
int os2s_gemm_wgrad_grouped(os2s_stream_t stream, const os2s_wgrad_group_t* groups, int ngroups,
                            long long M, int accumulate, void* workspace, size_t workspace_bytes);
/* Plain GEMM, hand-written for the gfx950 matrix cores (csrc/gemm_pp.hip):
 *   C[M,N] (+)= A[M,K] . W[N,K]^T, C = residual + dropout(act(. + bias)) as in os2s_conv1d_fwd_ex.
 * Replaces tf.layers.Dense of the Transformer (parts/transformer/attention_layer.py:54-62,125-127,
 * 219; ffn_layer.py:51-85; the tied softmax, embedding_layer.py:90-105) and, applied to the
 * transposed weight copy, its data gradient. A has row stride lda (elements), W is contiguous
 * [N, K], K % 64 == 0; bf16 output needs N % 8 == 0 and ldc % 8 == 0. */
int os2s_gemm_nt(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W, void* C,
                 long long ldc, int M, int N, int K, const float* bias, int act, float keep_prob,
                 unsigned long long seed, const uint16_t* residual, int accumulate, int out_f32);
/* The same with a caller-owned workspace (os2s_conv1d_workspace_bytes(), the contract of
 * os2s_conv1d_fwd_ws: tickets zero on entry and on exit, one workspace per stream): the tiles of
 * the last partial round of workgroups are cut along K and reduced deterministically by the last
 * arriver when the cost model says so (few output tiles x a long reduction: the data gradient of
 * a vocabulary projection). Same result up to fp32 summation order. */
int os2s_gemm_nt_ws(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W, void* C,
                    long long ldc, int M, int N, int K, const float* bias, int act, float keep_prob,
                    unsigned long long seed, const uint16_t* residual, int accumulate, int out_f32,
                    void* workspace, size_t workspace_bytes);
/* The data gradient of a Dense layer whose input is the output of a ReLU + dropout layer
 * (parts/transformer/ffn_layer.py:51-85: output_layer(dropout(relu(filter_layer(x))))), with that
 * layer's activation / dropout backward fused into the epilogue:
 *   C[M,N] = (A[M,K] . W[N,K]^T) * (mask_ref[m,n] > 0 ? mask_scale : 0)        (bf16)
 * mask_ref = the saved forward output of the ReLU + dropout layer (row stride ldc), mask_scale =
 * 1 / keep_prob. stats (or NULL): [ceil(M/128), 2, N] fp32 per-128-row-window column sums / sums of
 * squares of C — the partial sums of that layer's bias gradient. Workspace as os2s_gemm_nt_ws. */
int os2s_gemm_nt_mask_ws(os2s_stream_t stream, const uint16_t* A, long long lda, const uint16_t* W, void* C,
                         long long ldc, int M, int N, int K, const uint16_t* mask_ref, float mask_scale,
                         float* stats, void* workspace, size_t workspace_bytes);
int os2s_conv1d_fwd(os2s_stream_t stream, const uint16_t* x, const uint16_t* w,
                    void* y, const int32_t* in_len, const float* bias,
                    float* stats, int B, int Tin, int Cin, int Cout, int K,
                    int stride, int dil, int padL, int Tout,
                    long long y_stride_b, long long y_stride_t, int out_f32,
                    int accumulate);

/* ------------------------------------------------------------------------
 * Weight gradient of os2s_conv1d_fwd (fp32 output, layout [K, Cout, Cin]):
 *   dW[k][co][ci] (+)= sum_b sum_t dy[b,t,co] * x[b, t*stride + k*dil - padL, ci]
 * with x rows >= in_len[b] read as zero (the masked conv input). This is the
 * gradient TF derives for tf.layers.conv1d (conv_blocks.py:195-206); it is
 * emitted in fp32 because MixedPrecisionOptimizerWrapper casts every gradient
 * to fp32 first (optimizers/mp_wrapper.py:79).
 *   accumulate = 0: dW is overwritten.  accumulate = 1: dW += (fp32 atomics; the
 *   batch may be split across workgroups to fill the chip).
 * ---------------------------------------------------------------------- */
int os2s_conv1d_wgrad(os2s_stream_t stream, const uint16_t* x, const uint16_t* dy,
                      float* dw, const int32_t* in_len, int B, int Tin, int Cin,
                      int Cout, int K, int stride, int dil, int padL, int Tout,
                      int accumulate);
/* same with x rows x_row_stride elements apart (x is a channel slice of a wider tensor) */
int os2s_conv1d_wgrad_ex(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                         const uint16_t* dy, float* dw, const int32_t* in_len, int B, int Tin,
                         int Cin, int Cout, int K, int stride, int dil, int padL, int Tout,
                         int accumulate);
/* Same with a caller-owned workspace (os2s_conv1d_workspace_bytes(), zero tickets, one per
 * stream — the contract of os2s_conv1d_fwd_ws). Stride-1 layers with K >= 2 run on a kernel that
 * cuts the reduction over the live (sample, 64-row) chunks when a layer has too few
 * (co, ci, tap) tiles to fill the chip; partial tiles are reduced in a fixed order by one
 * owner per tile, which also adds the previous dW when accumulate = 1: no fp32 atomics, results
 * are run-to-run identical. */
int os2s_conv1d_wgrad_ws(os2s_stream_t stream, const uint16_t* x, long long x_row_stride,
                         const uint16_t* dy, float* dw, const int32_t* in_len, int B, int Tin,
                         int Cin, int Cout, int K, int stride, int dil, int padL, int Tout,
                         int accumulate, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * BatchNorm + residual sum + activation + dropout + sequence mask
 * (conv_bn_actv / conv_bn_res_bn_actv, open_seq2seq/parts/cnns/conv_blocks.py:61-232;
 *  tf.nn.dropout at encoders/tdnn_encoder.py:255; mask at :185-186,204-205).
 * All activations bf16 [B,T,C] channels-last; statistics / affine params fp32 [C].
 * TF fused-BN conventions: normalise with the biased batch variance over all B*T
 * positions (padded frames included); moving = moving*momentum + batch*(1-momentum)
 * with Bessel-corrected batch variance.
 * ---------------------------------------------------------------------- */
/* partial [nparts,2,C] (sum, sumsq) -> mean/rstd (saved for backward), fused
 * scale = gamma*rstd, shift = beta - mean*scale; training=0 uses moving stats. */
int os2s_bn_finalize(os2s_stream_t stream, const float* partial, int nparts, int C,
                     long long count, const float* gamma, const float* beta,
                     float eps, float momentum, int training, float* moving_mean,
                     float* moving_var, float* mean_out, float* rstd_out,
                     float* scale_out, float* shift_out);
/* The same for J <= 16 BatchNorms of one geometry (nparts, C, count) in one launch: the 1x1 residual
 * branches of a conv_bn_res_bn_actv block end (conv_blocks.py:134-168 builds one BatchNorm per branch).
 * Every argument that is a pointer above is an array of J pointers here (the arrays live on the host). */
int os2s_bn_finalize_multi(os2s_stream_t stream, int J, const float* const* partial, int nparts, int C,
                           long long count, const float* const* gamma, const float* const* beta, float eps,
                           float momentum, int training, float* const* moving_mean,
                           float* const* moving_var, float* const* mean_out, float* const* rstd_out,
                           float* const* scale_out, float* const* shift_out);
/* stand-alone statistics partials for producers other than os2s_conv1d_fwd */
int os2s_bn_stats_num_parts(long long rows);
int os2s_bn_stats(os2s_stream_t stream, const uint16_t* y, long long rows, int C,
                  float* partial);
/* out = seqmask * dropout(act(sum_j y_j*scale_j + shift_j)); act: 0 none, 1 relu,
 * 2 tanh; J <= 12 inputs (main conv + dense-residual branches). y/scale/shift are
 * HOST arrays of J device pointers. */
int os2s_bn_act_fwd(os2s_stream_t stream, int J, const uint16_t* const* y,
                    const float* const* scale, const float* const* shift,
                    uint16_t* out, const int32_t* out_len, int B, int T, int C,
                    int act, float keep_prob, unsigned long long seed);
/* backward pass 1: dz = dout * mask * dropout' * act'(out) (bf16) and per-channel
 * partial sums partial[nparts][1+J][C] = {sum dz, sum dz*xhat_j}. */
int os2s_bn_act_bwd_num_parts(long long rows);
int os2s_bn_act_bwd_reduce(os2s_stream_t stream, int J, const uint16_t* dout,
                           const uint16_t* out, const uint16_t* const* y,
                           const float* const* mean, const float* const* rstd,
                           uint16_t* dz, float* partial, const int32_t* out_len,
                           int B, int T, int C, int act, float keep_prob,
                           unsigned long long seed);
/* reduce the partials for input q-1: dgamma (sum dz*xhat), dbeta (sum dz),
 * c1 = mean(dz), c2 = mean(dz*xhat) */
int os2s_bn_bwd_finalize(os2s_stream_t stream, const float* partial, int nparts,
                         int nq, int q, int C, long long count, float* dgamma,
                         float* dbeta, int accumulate, float* c1, float* c2);
/* The same for all J inputs of a residual block end in one launch: partial is
 * [nparts, 1 + J, C], dgamma / dbeta are J pointers (entries may be NULL), c1 / c2 are [J, C]. */
int os2s_bn_bwd_finalize_multi(os2s_stream_t stream, const float* partial, int nparts, int J,
                               int C, long long count, float* const* dgamma,
                               float* const* dbeta, int accumulate, float* c1, float* c2);
/* backward pass 2: dy = gamma*rstd*(dz - c1 - xhat*c2) */
int os2s_bn_bwd_apply(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                      const float* gamma, const float* mean, const float* rstd,
                      const float* c1, const float* c2, uint16_t* dy,
                      long long rows, int C);
/* The same on a ragged [B, T, C] batch: rows t >= out_len[b] + margin are written as zeros without
 * reading dz / y. `margin` = how far past the sequence end the consumers of dy look — (K-1)*dilation
 * for the data- and weight-gradient convolutions of a K-tap layer (dy is not zero there: the batch
 * statistics run over padded frames too). out_len NULL = every row. */
int os2s_bn_bwd_apply_ragged(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                             const float* gamma, const float* mean, const float* rstd,
                             const float* c1, const float* c2, uint16_t* dy,
                             const int32_t* out_len, int margin, int B, int T, int C);
/* ... with a dz that is defined only for rows t < out_len[b] (written by os2s_conv1d_dgrad_bnact_ws,
 * which skips the rest) and is taken as zero beyond, unread */
int os2s_bn_bwd_apply_ragged_dz(os2s_stream_t stream, const uint16_t* dz, const uint16_t* y,
                             const float* gamma, const float* mean, const float* rstd,
                             const float* c1, const float* c2, uint16_t* dy,
                             const int32_t* out_len, int margin, int B, int T, int C);
/* test hook: the keep/drop bits (one byte per 8 consecutive elements) the
 * dropout of os2s_bn_act_fwd uses for (seed, keep_prob). */
int os2s_dropout_mask(os2s_stream_t stream, unsigned long long seed, long long n8,
                      float keep_prob, uint8_t* out);

/* ------------------------------------------------------------------------
 * CTC loss + gradient w.r.t. logits. Replaces tf.nn.ctc_loss(...,
 * ignore_longer_outputs_than_inputs=True) + mask_nans + reduce_mean
 * (open_seq2seq/losses/ctc_loss.py:77-88); blank = V-1 in the reference.
 *   logits [T,B,V] fp32 time-major (pre-softmax), in_len [B], labels [B,Lmax]
 *   int32 (only the first label_len[b] entries are read, ctc_loss.py:12-16).
 *   loss_per_sample [B] (= -log p, 0 for ignored / non-finite samples),
 *   loss_mean [1] (mean over the whole batch), both optional.
 *   dlogits [T,B,V] fp32 and/or dlogits_bf16 [B,T,Vpad] bf16 (zero padded
 *   channels; feeds the FC backward GEMMs) = grad_scale * d(loss_b)/d(logits),
 *   additionally multiplied by *grad_scale_dev when that device pointer is not
 *   NULL (the device-resident loss scale of the mixed-precision optimizer).
 * ---------------------------------------------------------------------- */
size_t os2s_ctc_loss_workspace_bytes(int T, int B, int V, int Lmax);
int os2s_ctc_loss(os2s_stream_t stream, const float* logits, const int32_t* in_len,
                  const int32_t* labels, const int32_t* label_len, int T, int B,
                  int V, int Lmax, int blank, float grad_scale,
                  const float* grad_scale_dev, float* loss_per_sample,
                  float* loss_mean, float* dlogits,
                  uint16_t* dlogits_bf16, int Vpad, void* workspace,
                  size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * Mixed-precision optimizer step over flat fp32 buffers (device-side skip
 * decision, no host sync). Replaces MixedPrecisionOptimizerWrapper
 * (optimizers/mp_wrapper.py:27-122), AutomaticLossScaler Backoff/LogMax
 * (automatic_loss_scaler.py:50-203), post_process_gradients LARC / global-norm
 * clip (optimizers.py:289-482), NovoGrad (novograd.py:93-126, AS WRITTEN — see novograd_ema), TF Momentum /
 * Adam, and the lr policies (lr_policies.py) evaluated at the device-resident
 * global step. Tensors start at multiples of os2s_opt_chunk_elems() elements.
 * ---------------------------------------------------------------------- */
#define OS2S_LR_MAX_BOUNDARIES 16
typedef struct {
  int optimizer;               /* 0 SGD, 1 Momentum, 2 NovoGrad, 3 Adam */
  float beta1, beta2, epsilon, weight_decay;
  int grad_averaging;          /* NovoGrad: g *= (1-beta1) */
  int lr_policy;               /* 0 fixed, 1 poly_decay, 2 exp_decay, 3 transformer_policy, 4 cosine_decay,
                                * 5 piecewise_constant (lr_policies.py:30-57), 6 inv_poly_decay (:203-245) */
  float learning_rate, min_lr, power, decay_rate, max_lr, coefficient;
  long long decay_steps, begin_decay_at, warmup_steps;
  int use_staircase_decay, d_model, has_max_lr;
  int use_larc;
  float larc_eta, larc_min_update, larc_epsilon;
  int larc_mode_scale;         /* 0 'clip' (default), 1 'scale' */
  float clip_global_norm;      /* <= 0: off */
  int scaler;                  /* 0 static, 1 Backoff, 2 LogMax */
  float scale_min, scale_max, step_factor;
  long long step_window;
  float log_max, lm_beta1, lm_beta2, overflow_std_dev;
  int world_size;              /* gradients in the buffer are sums over ranks */
  /* piecewise_constant: lr = learning_rate * pw_rates[i], i = number of boundaries < global_step
   * (tf.train.piecewise_constant: the value changes AFTER a boundary step); pw_rates[0] = 1. */
  int pw_count;                /* number of boundaries, <= OS2S_LR_MAX_BOUNDARIES */
  long long pw_boundaries[16];
  float pw_rates[17];
  int novograd_ema;            /* 0 (reference): v_t = |g_t|^2 every step — the reference graph
                                * never assigns nvgrad2_ema* (novograd.py:107-113: the tf.cond result
                                * only replaces the Python list entry), so beta2 is dead there;
                                * 1: v_t = beta2 v_{t-1} + (1-beta2) |g_t|^2 (the published NovoGrad) */
} os2s_opt_config_t;

int os2s_opt_chunk_elems(void);
size_t os2s_opt_state_bytes(void);
size_t os2s_opt_config_bytes(void);
int os2s_opt_init_state(os2s_stream_t stream, void* state, float loss_scale);
int os2s_opt_step(os2s_stream_t stream, const os2s_opt_config_t* cfg, void* state,
                  const float* grads, float* weights, float* m1, float* m2,
                  uint16_t* w16, int nchunks, int ntensors,
                  const int32_t* chunk_tensor, const int32_t* tensor_chunk_begin,
                  const float* tensor_l2, const float* tensor_wd_mask,
                  float* partial, float* tensor_gnorm2, float* tensor_wnorm2,
                  float* tensor_amax, float* tensor_mult, float* tensor_v);
/* The same step in two halves, so that the update can run on its own stream NEXT TO the following forward pass
 * (optimizers/optimizers.py TrainOp.run_async; the reference's train_op is one sess.run, optimizers.py:107-160 /
 * mp_wrapper.py:84-117): os2s_opt_prepare = everything that needs ALL gradients (statistics, overflow / skip decision,
 * loss-scale update, learning rate, LARC / NovoGrad factors); os2s_opt_apply_range = the update of chunks
 * [chunk_begin, chunk_end) — the caller records an event behind each range and the forward pass waits for the range
 * that holds a variable before it first reads it. zero_grads != 0: each gradient chunk is written back as zeros once
 * it has been read (also on a skipped step), replacing the fill of the gradient buffer that opens the next step.
 * os2s_opt_step == os2s_opt_prepare + os2s_opt_apply_range(0, nchunks, zero_grads = 0). */
int os2s_opt_prepare(os2s_stream_t stream, const os2s_opt_config_t* cfg, void* state,
                     const float* grads, const float* weights, int nchunks, int ntensors,
                     const int32_t* chunk_tensor, const int32_t* tensor_chunk_begin,
                     const float* tensor_l2, float* partial, float* tensor_gnorm2, float* tensor_wnorm2,
                     float* tensor_amax, float* tensor_mult, float* tensor_v);
int os2s_opt_apply_range(os2s_stream_t stream, const os2s_opt_config_t* cfg, const void* state,
                         float* grads, float* weights, float* m1, float* m2, uint16_t* w16,
                         int chunk_begin, int chunk_end, const int32_t* chunk_tensor,
                         const float* tensor_l2, const float* tensor_mult,
                         const float* tensor_wd_mask, int zero_grads);
int os2s_cast_f32_to_bf16(os2s_stream_t stream, const float* src, uint16_t* dst,
                          long long n);
/* batched dgrad copies of conv weights: wT[k'][ci][co] = w[K-1-k'][co][ci];
 * descs: device array of {int64 src_off, int64 dst_off, int32 K, Cout, Cin, tile_begin} */
int os2s_conv_weight_dgrad_copy(os2s_stream_t stream, const uint16_t* w16,
                                uint16_t* wt16, const void* descs, int ndesc,
                                int total_tiles);

/* ------------------------------------------------------------------------
 * ASR log-mel front end ("logfbank", librosa backend). Replaces the NumPy/librosa
 * feature extraction the reference runs on the host inside tf.py_func:
 * get_speech_features_librosa (open_seq2seq/data/speech2text/speech_utils.py:322-441):
 * gain normalise -> dither -> pre-emphasis -> centred reflect-padded STFT (n_fft 512,
 * symmetric Hann of win_length zero-padded to n_fft) -> |.|^2 -> mel -> log(.+floor)
 * -> per-feature (or global) mean/std over time; output is the zero-padded batch
 * [B, Tpad, n_mels] the data layer hands to the encoder (speech2text.py:313-317).
 *   signal      [B, Nmax] float32 or int16 PCM (sample_is_int16), n_samples [B]
 *   window      [n_fft] fp32 device (already centred/zero-padded)
 *   mel tables  compact form of the [n_mels, n_fft/2+1] basis: for filter m the
 *               non-zero run starts at bin mel_start[m], has mel_len[m] bins, weights
 *               mel_wt[j*n_mels + m]
 *   fixed_gain  > 0: use it; <= 0: 1/(max|x| + 1e-5) per utterance
 *   frames per utterance = 1 + n_samples/hop (out_len), rows >= that are zero
 * Only n_fft == 512 and n_mels <= 64 are implemented (OS2S_ERR_UNSUPPORTED otherwise).
 * ---------------------------------------------------------------------- */
/* CRC-32C (Castagnoli) of a host buffer, extendable: crc = os2s_crc32c(0, a, na);
 * crc = os2s_crc32c(crc, b, nb). Used by the TensorBundle checkpoint reader / writer
 * (tf.train.Saver files: open_seq2seq/utils/funcs.py:117-144, utils/helpers.py:462-553). Host only. */
uint32_t os2s_crc32c(uint32_t init, const void* data, size_t n);

/* 'spectrogram' features of the python_speech_features backend (get_speech_features_psf,
 * open_seq2seq/data/speech2text/speech_utils.py:444-535; the DeepSpeech2 configs: 160 bins of a
 * 320-point spectrum): int16 re-quantisation of the gain-normalised signal, frames of n_win samples
 * every n_step (zero-padded tail, symmetric Hann), 10 log10(|rfft|^2 / n_win) clipped at 1e-30,
 * first num_features bins, (x - mean) / std over the utterance INCLUDING the zero frames that round
 * the frame count up to a multiple of pad_to. out_len[b] = that frame count (<= Tpad required);
 * rows past it are zero. */
size_t os2s_psf_spectrogram_workspace_bytes(int B, int Tpad, int num_features);
int os2s_psf_spectrogram(os2s_stream_t stream, const void* signal, int sample_is_int16,
                         const int32_t* n_samples, int B, long long Nmax, int n_win, int n_step,
                         int pad_to, int num_features, int Tpad, uint16_t* out_bf16, float* out_f32,
                         int32_t* out_len, void* workspace, size_t workspace_bytes);
/* 'logfbank' features of the python_speech_features backend (get_speech_features_psf,
 * open_seq2seq/data/speech2text/speech_utils.py:517-535 -> psf.logfbank(nfft = 512, lowfreq = 0,
 * highfreq = sr / 2, preemph = 0.97): int16 normalisation, pre-emphasis, rectangular frames, |rfft|^2 / nfft,
 * the filter table fb [nfilt][nfft/2 + 1] (device, fp32: python_speech_features.get_filterbanks), ln, then
 * (x - mean) / std over the utterance incl. the pad_to frames). Workspace:
 * os2s_psf_spectrogram_workspace_bytes(B, Tpad, nfilt). The toy Wave2Letter / TDNN configurations of the
 * reference's acceptance tests use this path. */
int os2s_psf_logfbank(os2s_stream_t stream, const void* signal, int sample_is_int16, const int32_t* n_samples,
                      int B, long long Nmax, int n_win, int n_step, int pad_to, int nfilt, int nfft,
                      const float* fb, int Tpad, uint16_t* out_bf16, float* out_f32, int32_t* out_len,
                      void* workspace, size_t workspace_bytes);
size_t os2s_logmel_workspace_bytes(int B, int Tmax, int n_mels);
int os2s_logmel(os2s_stream_t stream, const void* signal, const int32_t* n_samples,
                int sample_is_int16, int B, long long Nmax, int n_fft, int hop,
                int n_mels, const float* window, const int32_t* mel_start,
                const int32_t* mel_len, const float* mel_wt, int mel_maxlen,
                float preemph, float dither, unsigned long long seed, float fixed_gain,
                float log_floor, int norm_per_feature, int Tmax, int Tpad,
                uint16_t* out_bf16, float* out_f32, int32_t* out_len, void* workspace,
                size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * Transformer NMT path (packed, token-major [N_tokens, hidden] bf16 tensors; sequences
 * are concatenated without padding, cu_* [B+1] hold token offsets).
 * ---------------------------------------------------------------------- */
/* EmbeddingSharedWeights.call (parts/transformer/embedding_layer.py:59-88): gather * sqrt(d),
 * ids >= V -> pad, pad id 0 -> zero vector; + get_position_encoding (utils.py:28-54) at
 * pos[n]; + tf.nn.dropout (encoders/transformer_encoder.py:157-161,
 * decoders/transformer_decoder.py:204-213). The decoder's shift-right is the caller's
 * index remap. */
int os2s_embed_fwd(os2s_stream_t stream, const int32_t* ids, const int32_t* pos,
                   const uint16_t* table, int V, int D, long long N, float emb_scale,
                   float keep_prob, unsigned long long seed, uint16_t* out, int plain_lookup);
/* its gradient: dtable[id] += emb_scale * dropout'(dout[n]) (fp32 atomics).
 * plain_lookup = 1: tf.nn.embedding_lookup as the RNN encoders/decoders use it
 * (encoders/rnn_encoders.py:283-289, decoders/rnn_decoders.py:262-266): no pad zeroing,
 * no position signal (pos may be NULL). */
int os2s_embed_bwd(os2s_stream_t stream, const int32_t* ids, const uint16_t* dout, int V,
                   int D, long long N, float emb_scale, float keep_prob,
                   unsigned long long seed, float* dtable, int plain_lookup);
/* LayerNormalization "layernorm_L2" (parts/transformer/common.py:41-68), D in {512, 1024} */
int os2s_layernorm_fwd(os2s_stream_t stream, const uint16_t* x, const float* gamma,
                       const float* beta, float eps, long long N, int D, uint16_t* y,
                       float* mean, float* rstd);
int os2s_layernorm_bwd_num_parts(long long N);
/* dx = dres + LN'(dy); partial [num_parts,2,D] = {sum dy, sum dy*xhat}: reduce with
 * os2s_bn_bwd_finalize(partial, nparts, 2, 1, D, ...) -> dbeta, dgamma */
int os2s_layernorm_bwd(os2s_stream_t stream, const uint16_t* dy, const uint16_t* x,
                       const float* gamma, const float* mean, const float* rstd,
                       const uint16_t* dres, long long N, int D, uint16_t* dx,
                       float* partial);
/* mode 0: d = dout * keepmask/keep (hash mask, PrePostProcessingWrapper dropout);
 * mode 1: d = dout * (out > 0)/keep (Dense+ReLU+dropout of FeedFowardNetwork) */
int os2s_dropout_bwd(os2s_stream_t stream, const uint16_t* dout, const uint16_t* out,
                     int mode, float keep_prob, unsigned long long seed, long long n,
                     uint16_t* d);
/* The same on a [rows, C] matrix, with the column sums of d (the gradient of the Dense layer's
 * bias: tf.layers.Dense(use_bias=True) of FeedFowardNetwork, parts/transformer/ffn_layer.py:51-85)
 * from the same pass: partial[os2s_dropout_bwd_colsum_num_parts(rows)][2][C] fp32, plane 0 =
 * per-workgroup column sums (plane 1 zero) — reduce with os2s_bn_bwd_finalize(q = 1). */
int os2s_dropout_bwd_colsum_num_parts(long long rows);
int os2s_dropout_bwd_colsum(os2s_stream_t stream, const uint16_t* dout, const uint16_t* out, int mode,
                            float keep_prob, unsigned long long seed, long long rows, int C,
                            uint16_t* d, float* partial);
int os2s_add_bf16(os2s_stream_t stream, const uint16_t* a, const uint16_t* b, long long n,
                  uint16_t* out);
/* Attention.call "loung" mode (parts/transformer/attention_layer.py:104-220): per
 * (batch, head) softmax(scale * q k^T + bias) -> dropout -> @ v, fp32 softmax. Head h owns
 * channels [h*dh, (h+1)*dh) of each row (split_heads/combine_heads are indexing only).
 * bias = padding mask (absent keys in the packed layout) and, if causal, the decoder's
 * lower-triangular band (utils.py:57-79). lse [Nq, H] is saved for the backward, which
 * recomputes the probabilities. dh == 64. Training (backward, attention dropout) is
 * implemented for max_len <= 64 (the length-filtered training sets of the configs); the
 * forward without dropout takes any max_len (eval / infer batches): one wave per 64-query
 * tile walks the key tiles with an online softmax. OS2S_ERR_UNSUPPORTED otherwise. */
int os2s_attention_fwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                       const uint16_t* v, uint16_t* o, float* lse, const int32_t* cu_q,
                       const int32_t* cu_k, int B, int H, int dh, int max_len,
                       long long ldq, long long ldk, long long ldv, long long ldo,
                       int causal, float scale, float keep_prob, unsigned long long seed);
int os2s_attention_bwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                       const uint16_t* v, const uint16_t* d_o, const float* lse,
                       uint16_t* dq, uint16_t* dk, uint16_t* dv, const int32_t* cu_q,
                       const int32_t* cu_k, int B, int H, int dh, int max_len,
                       long long ldq, long long ldk, long long ldv, long long lddo,
                       long long lddq, long long lddk, long long lddv, int causal,
                       float scale, float keep_prob, unsigned long long seed);
/* PaddedCrossEntropyLossWithSmoothing (losses/sequence_loss.py:257-309) over the N
 * non-pad target rows: row_loss[n] = xent(soft targets) - normalizing constant;
 * loss_mean = grad_scale * sum(row_loss); dlogits = grad_scale * (*grad_scale_dev) *
 * (softmax - soft_target). Pass grad_scale = 1/N for the token mean (Transformer) or
 * 1/batch_size with label_smoothing = 0 for BasicSequenceLoss (losses/sequence_loss.py:53-114:
 * sparse softmax xent summed over time, divided by the batch size).
 * logits/dlogits bf16 [N, ld], V % 8 == 0, V <= 40960; columns >= V_valid are vocabulary
 * padding (treated as -inf logits, zero gradient). */
int os2s_xent_smooth(os2s_stream_t stream, const uint16_t* logits, const int32_t* labels,
                     long long N, int V, int V_valid, long long ld, float label_smoothing,
                     float grad_scale, const float* grad_scale_dev, float* row_loss,
                     float* loss_mean, uint16_t* dlogits);
/* rows with label < 0 are masked positions (weight 0: no loss, zero gradient). */
/* tf.argmax(logits, -1) over the first V_valid columns of bf16 rows (decoders'
 * 'outputs', decoders/rnn_decoders.py:318, GreedyEmbeddingHelper sampling). */
int os2s_argmax_rows(os2s_stream_t stream, const uint16_t* x, long long N, int V_valid,
                     long long ld, int32_t* out);

/* ------------------------------------------------------------------------
 * Speech data-layer augmentation on the device (data/speech2text/speech_utils.py):
 * os2s_augment_signal = normalize_signal (:225-231) -> speed perturbation by band-limited
 * sinc resampling (augment_audio_signal :234-272 calls resampy.resample(..., 'kaiser_best');
 * the resampy interpolation loop is restated, the half-window table interp_win[nwin]
 * (num_zeros * num_table + 1 taps) comes from the caller) -> additive Gaussian noise.
 * Per sample b: n_out[b] = int(n_in[b] * ratio[b]) output samples (ratio 1.0 = copy),
 * noise_amp[b] = 10^(dB/20) (0 = none; noise_amp may be NULL). fixed_gain > 0 overrides the
 * 1/(max|x| + 1e-5) normalisation. out: fp32 [B, nout_max], zero beyond n_out. absmax_scratch:
 * B uint32. os2s_spec_augment zeroes the half-open boxes masks[b][m] = (t0, t1, f0, f1) of
 * bf16 features [B, T, F] (SpecAugment :419-433; the random draws stay on the host).
 * ---------------------------------------------------------------------- */
int os2s_augment_signal(os2s_stream_t stream, const void* signal, int is_int16, int B,
                        long long nmax, const int32_t* n_in, const int32_t* n_out,
                        const double* ratio, const float* noise_amp, float fixed_gain,
                        const float* interp_win, int nwin, int num_table, unsigned long long seed,
                        uint32_t* absmax_scratch, float* out, long long nout_max);
int os2s_spec_augment(os2s_stream_t stream, uint16_t* feats, int B, int T, int F,
                      const int32_t* masks, int n_masks);

/* ------------------------------------------------------------------------
 * Transformer beam-search inference: the device side of
 * SequenceBeamSearch (parts/transformer/beam_search.py:62-383) and of the incremental
 * decoder step (decoders/transformer_decoder.py:232-326).
 *
 * Loop state (caller-allocated device buffers, L1 = max_decode_length + 1):
 *   status     int32[4]  {running, cur_index, ticket, max_decode_length}
 *   alive_seq  int32[2][B][beam][L1]   ping-pong: step i reads plane i&1, writes the other
 *   fin_seq    int32[2][B][beam][L1]
 *   alive_lp   f32[B][beam]  fin_scores f32[B][beam]  fin_flags int32[B][beam]
 * os2s_beam_init = _create_initial_state (:96-161). os2s_beam_step = one _search_step
 * (:205-383) on this step's logits [B*beam, V] (row stride ld; bf16, or fp32 when
 * logits_f32) followed by _continue_search (:163-203) for the next iteration: it clears
 * status[0] when the search is over, and every kernel of a later os2s_beam_step call is
 * then a no-op (the host may enqueue ahead and poll status only every few steps). INF =
 * 32768 (:26); tf.nn.top_k order (descending, lower index first among equals); lnorm[len]
 * = ((5 + len) / 6)^alpha for len = 0..max_decode_length (fp32, host-computed, :424-426).
 * parent[B*beam] receives the flat row (b*beam + old beam) each new alive beam descends
 * from — gather per-beam caches with os2s_gather_rows. topk_lp / topk_idx [B, 2*beam]
 * optionally receive the step's top-2*beam log-probs / flat candidate indices (tests);
 * last_ids / pos [B*beam] (optional) are kept equal to the last token and the position
 * (= cur_index) of every alive beam, so that a decoder step can be driven entirely from
 * device state (no host-side loop index: the step is hipGraph-capturable).
 * Needs 2*beam <= 64 and V >= 2*beam. os2s_beam_finalize = search() epilogue (:85-94).
 * ---------------------------------------------------------------------- */
int os2s_beam_chunks(int V);
long long os2s_beam_workspace_bytes(int B, int beam, int V);
int os2s_beam_init(os2s_stream_t stream, int B, int beam, int max_decode_length,
                   const int32_t* initial_ids, int32_t* status, int32_t* alive_seq,
                   int32_t* fin_seq, float* alive_lp, float* fin_scores, int32_t* fin_flags,
                   int32_t* last_ids, int32_t* pos);
int os2s_beam_step(os2s_stream_t stream, const void* logits, int logits_f32, long long ld, int B,
                   int beam, int V, int max_decode_length, int eos_id, const float* lnorm,
                   int32_t* status, int32_t* alive_seq, int32_t* fin_seq, float* alive_lp,
                   float* fin_scores, int32_t* fin_flags, int32_t* parent, float* topk_lp,
                   int32_t* topk_idx, int32_t* last_ids, int32_t* pos, void* workspace);
int os2s_beam_finalize(os2s_stream_t stream, int B, int beam, int max_decode_length,
                       const int32_t* status, const int32_t* alive_seq, const int32_t* fin_seq,
                       const float* alive_lp, const float* fin_scores, const int32_t* fin_flags,
                       int32_t* out_seq, float* out_scores);
/* One step of tf.contrib.seq2seq.BeamSearchDecoder (_beam_search_step) as driven by
 * BeamSearchRNNDecoderWithAttention (decoders/rnn_decoders.py:324-532): logits [B*beam, V] ->
 * top `beam` continuations per batch item by (log-prob total) / ((5 + length) / 6)^weight, finished
 * beams continue with END at no cost, at time 0 only beam 0 competes. State arrays [B*beam]
 * (log_probs: init {0, -inf, ...}; finished, lengths: init 0) are updated in place; word_ids,
 * parent (flat parent ROW) and scores receive the step outputs (gather decoder state with
 * os2s_gather_rows; trace the final sequences back through parent). V <= 65536, beam <= 64. */
long long os2s_tf_beam_workspace_bytes(int B, int beam);
int os2s_tf_beam_step(os2s_stream_t stream, const void* logits, int logits_f32, long long ld, int B,
                      int beam, int V, int eos_id, int time, float length_penalty_weight,
                      float* log_probs, int32_t* finished, int32_t* lengths, int32_t* word_ids,
                      int32_t* parent, float* scores, void* workspace);
/* dst[r] = src[idx[r]] for rows of row_bytes (multiple of 4) — _gather_beams (:505-537) with
 * flat row indices. If enable != NULL and enable[0] == 0 the rows are copied unpermuted
 * (pass the beam status so that a finished search stops permuting its caches). */
int os2s_gather_rows(os2s_stream_t stream, const void* src, const int32_t* idx, long long rows,
                     long long row_bytes, const int32_t* enable, void* dst);
/* In place on bf16 y [rows, C]: y = residual + dropout(act(y + bias)) — the epilogue of a Dense
 * layer whose matmul ran as a bare GEMM (same semantics and dropout stream as the fused
 * epilogue of os2s_conv1d_fwd_ex with K = 1; act 1 = ReLU). bias / residual may be NULL. */
int os2s_dense_epilogue(os2s_stream_t stream, uint16_t* y, const float* bias, long long rows, int C,
                        int act, float keep_prob, unsigned long long seed, const uint16_t* residual);
/* Y[M,N] = act(X[M,K] . W[N,K]^T + bias) (+ residual) for SMALL M (decoding steps: M =
 * batch*beam rows): one 32x32 output tile per workgroup, K split over its 8 waves with all
 * loads issued up front — the latency-bound regime where the training GEMM
 * (os2s_conv1d_fwd, K=1) leaves most CUs idle. bf16 in/out, fp32 accumulate/bias; relu != 0
 * applies ReLU before the residual add. N % 4 == 0, K % 8 == 0. */
int os2s_gemm_skinny(os2s_stream_t stream, const uint16_t* x, long long ldx, const uint16_t* w,
                     long long ldw, const float* bias, const uint16_t* residual, long long ldr,
                     int M, int N, int K, int relu, uint16_t* y, long long ldy);
/* Decoder self-attention for ONE new position per beam row (SelfAttention with cache,
 * parts/transformer/attention_layer.py:133-139): appends knew/vnew [N, H*dh] (row stride
 * ldnew) to the append-only caches [N, Tmax, H*dh] at slot `step`, sets
 * ancestry[n, step] = n, and attends over positions 0..step of beam n's history, position j
 * living in cache row ancestry[n, j] ([N, Tmax] int32, permuted by the beam search instead
 * of the caches). If status_dev != NULL the step is read from status_dev[1] (device-side
 * loop index). dh == 64. */
int os2s_decode_self_attention(os2s_stream_t stream, const uint16_t* q, long long ldq,
                               const uint16_t* knew, const uint16_t* vnew, long long ldnew,
                               uint16_t* kcache, uint16_t* vcache, int32_t* ancestry, int N,
                               int H, int dh, int Tmax, int step, const int32_t* status_dev,
                               float scale, uint16_t* o, long long ldo);
/* Encoder-decoder attention for one query per beam row over the PACKED encoder keys/values
 * [N_src, ldkv] projected once per sentence; beam row n attends sentence n / beam
 * (tokens cu_k[b] .. cu_k[b+1]). */
int os2s_decode_cross_attention(os2s_stream_t stream, const uint16_t* q, long long ldq,
                                const uint16_t* k, const uint16_t* v, long long ldkv,
                                const int32_t* cu_k, int beam, int N, int H, int dh,
                                int max_len, float scale, uint16_t* o, long long ldo);

/* ------------------------------------------------------------------------
 * Recurrent layers (one direction of one layer per call; the time loop is inside).
 *   cell 0: cuDNN GRU  (tf.contrib.cudnn_rnn.CudnnGRU, encoders/ds2_encoder.py:294-328)
 *   cell 1: cuDNN LSTM (CudnnLSTM, encoders/tacotron2_encoder.py:254-263), gates i,f,g,o
 *   cell 2: tf.nn.rnn_cell.LSTMCell (encoders/rnn_encoders.py:292-300,
 *           parts/rnns/utils.py:17-89), gates i,j,f,o with forget_bias
 * gx [B,T,G*H] bf16 = input projections of all steps (x Wx^T + input bias; computed by
 * the caller with os2s_conv1d_fwd); wh [G*H,H] bf16; bh [G*H] fp32 recurrent bias or NULL;
 * lens [B] or NULL: dynamic_rnn semantics (state passes through and outputs are zero
 * past a sequence end; reverse=1 processes each sequence from ITS last frame);
 * y: bf16 outputs, row (b,t) at y + (b*T+t)*ldy (ldy >= H: both directions of a layer can
 * write the two halves of one [B,T,2H] tensor); gates [B,T,4H] bf16 and c_seq [B,T,H] fp32 are saved for the
 * backward (may be NULL for inference). Initial states are zero.
 * Backward: dy [B,T,H] -> dgx (and dgr, the recurrent-side gate gradients, which differ
 * from dgx only for the GRU candidate gate; pass NULL for LSTMs). The caller derives
 * dX = dgx Wx, dWx = dgx^T X, dWh = dgr^T H_prev (time-shifted outputs), biases = column
 * sums with the GEMM/wgrad entry points.
 * ---------------------------------------------------------------------- */
size_t os2s_rnn_fwd_workspace_bytes(int B, int H);
/* cuDNN-form GRU layers with B <= 32, H <= 1024 run their whole forward recurrence in ONE persistent
 * launch (csrc/rnn_xcd.hip: a direction per XCD, weights stationary in registers, the hidden state
 * exchanged through that XCD's L2) instead of one launch per time step; os2s_rnn_fwd_workspace_bytes
 * includes its exchange buffers. os2s_gru_xcd_set_mode: 0 = per-step launches, 1 = persistent kernel,
 * -1 = environment OS2S_GRU_XCD (default on), 2 (test hook) = on, and the next forward launch starts with
 * its abort flag set. A launch that gives up (unexpected workgroup placement, a wait that times out) leaves
 * its outputs partially written: its abort code is latched, behind every persistent forward AND backward
 * launch, into a sticky host-visible word no launch clears; os2s_gru_xcd_status() returns it (1 = timeout,
 * 2 = placement, 3 = both; 0 = fine so far) — definite after a stream synchronisation; clear != 0 resets the
 * word. Launches do not fail on a set word: in a data-parallel job every rank must enqueue the same collectives,
 * so the step runs to its end and the host layer (Model.train_step) reads the word after every step that ran
 * persistent launches (os2s_gru_xcd_launch_count), agrees on the answer over the ranks and redoes an aborted
 * step on the launch-per-step kernels (os2s_gru_xcd_set_mode(0)). */
long long os2s_gru_xcd_launch_count(void);
void os2s_gru_xcd_set_mode(int mode);
int os2s_gru_xcd_status(int clear);
size_t os2s_gru_xcd_workspace_bytes(int B, int H);
size_t os2s_gru_xcd_bwd_workspace_bytes(int B, int H);   /* backward-through-time twin (B <= 16) */
int os2s_rnn_layer_fwd(os2s_stream_t stream, int cell, const uint16_t* gx, const uint16_t* wh,
                       const float* bh, const int32_t* lens, int B, int T, int H, int reverse,
                       float forget_bias, uint16_t* y, long long ldy, uint16_t* gates,
                       float* c_seq, void* workspace, size_t workspace_bytes);
size_t os2s_rnn_bwd_workspace_bytes(int B, int H);
int os2s_rnn_layer_bwd(os2s_stream_t stream, int cell, const uint16_t* whT, const int32_t* lens,
                       const uint16_t* dy, long long lddy, const uint16_t* y, long long ldy,
                       const uint16_t* gates, const float* c_seq, int B, int T, int H, int reverse,
                       float forget_bias, uint16_t* dgx, uint16_t* dgr, void* workspace,
                       size_t workspace_bytes);
/* Both directions of a bidirectional layer (tf.nn.bidirectional_dynamic_rnn /
 * cuDNN direction="bidirectional", encoders/ds2_encoder.py:304-321) in ONE launch per time
 * step: ndir = 1 or 2 descriptors with the same meaning as the arguments above; the
 * workspace is ndir x os2s_rnn_{fwd,bwd}_workspace_bytes. */
typedef struct os2s_rnn_dir_fwd {
  const uint16_t* gx; const uint16_t* wh; const float* bh;
  uint16_t* y; long long ldy; uint16_t* gates; float* c_seq; int reverse;
} os2s_rnn_dir_fwd_t;
typedef struct os2s_rnn_dir_bwd {
  const uint16_t* whT; const uint16_t* dy; long long lddy; const uint16_t* y; long long ldy;
  const uint16_t* gates; const float* c_seq; uint16_t* dgx; uint16_t* dgr; int reverse;
} os2s_rnn_dir_bwd_t;
int os2s_rnn_layer_fwd_multi(os2s_stream_t stream, int cell, int ndir,
                             const os2s_rnn_dir_fwd_t* dirs, const int32_t* lens, int B, int T,
                             int H, float forget_bias, void* workspace, size_t workspace_bytes);
int os2s_rnn_layer_bwd_multi(os2s_stream_t stream, int cell, int ndir,
                             const os2s_rnn_dir_bwd_t* dirs, const int32_t* lens, int B, int T,
                             int H, float forget_bias, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * Attention-RNN decoder loop: the AttentionWrapper cell that tf.contrib.seq2seq.dynamic_decode
 * steps in RNNDecoderWithAttention (gnmt / gnmt_v2: decoders/rnn_decoders.py:147-321,
 * parts/rnns/gnmt.py:32-79) and Tacotron2Decoder (decoders/tacotron2_decoder.py:257-420):
 *   per step t:  cat0[t] = [dropout(attention_{t-1}) (M), h0_{t-1} (H)]
 *                h0 = LSTMCell(gx0[t] + cat0[t] Wcat0^T)            (gates i,j,f,o; forget_bias)
 *                [L == 2: cat1[t] = [dropout_out(h0_t), h1_{t-1}];  h1 = LSTMCell(cat1[t] Wcat1^T + bias1)]
 *                query = dropout_out(h_top) Wq^T
 *                score_s = nv . tanh(keys_s + query [+ location_s] [+ b]),  masked softmax over s < src_len
 *                   score_mode 0: nv = v (Bahdanau, parts/rnns/attention_wrapper.py:536-539)
 *                              1: nv = g v/|v|, bias b (normalised Bahdanau, :520-535)
 *                              2: location-sensitive (:641-715, 749-878): location_s = dense_w^T
 *                                 (conv1d_SAME(cumulative alignments; conv_w [K,F], conv_b) at s),
 *                                 cumulative state += alignments after the step; bias b iff use_bias
 *                              3: Luong (multiplicative, tf.contrib.seq2seq.LuongAttention as built by
 *                                 decoders/rnn_decoders.py:100-111): score_s = keys_s . query with the
 *                                 query = cell output (pass wq = identity, U == H; v is ignored)
 *                attention_t = context = sum_s align_s values_s   (attention_layer_size = None)
 * Everything outside the recurrence is the caller's (GEMM entry points): gx0 = inputs W_in^T
 * + bias for all steps, keys = values W_mem^T, layers above the attention cell, projections.
 * All state lives in caller-owned sequence buffers (zero them before t = 0):
 *   cat[l]   bf16 [B, T+1, Kc_l]  (Kc_0 = M+H, Kc_1 = 2H): row t = input of step t
 *   c_seq[l] fp32 [B, T, H], gates[l] bf16 [B, T, 4H] (saved for backward; gates may be NULL
 *   for inference), cum_seq fp32 [B, T+1, S] (mode 2; row t = state before step t),
 *   align_seq fp32 [B, T, S], q_seq fp32 [B, T, U]
 *   y_top: dropped top-cell outputs, row (b,t) at y_top + b*y_top_bs + t*y_top_ts (bf16)
 *   ctx:   raw contexts, row (b,t) at ctx + b*ctx_bs + t*ctx_ts (bf16)
 * so os2s_attn_decoder_fwd may be called for any step range [t_begin, t_end): the whole
 * teacher-forced sequence at once, or one step at a time for greedy / free-running decoding.
 * tgt_len (or NULL): steps t >= tgt_len[b] leave sample b untouched (impute_finished).
 * Dropout masks come from the library's counter hash: attention-input dropout indexes the
 * logical [B, T+1, M] tensor (row t+1 = attention_t), output dropout [B, T, H] per layer.
 * Limits: H % 8 == 0, M % 8 == 0, U % 128 == 0 and <= 512; location-sensitive mode:
 * U == 128, loc_k <= 32.
 * ---------------------------------------------------------------------- */
typedef struct os2s_attn_decoder {
  int B, T, S, L, H, M, U;
  int score_mode, use_bias, loc_k, loc_f;
  int t_begin, t_end;
  float forget_bias;
  float attn_in_keep; unsigned long long attn_in_seed;
  float out_keep; unsigned long long out_seed[2];
  /* parameters */
  const uint16_t* wcat[2];      /* bf16 [4H, Kc_l] */
  const float* bias[2];         /* fp32 [4H] or NULL */
  const uint16_t* wq;           /* bf16 [U, H] */
  const float* v; const float* g; const float* b;                 /* [U], [1], [U] */
  const float* conv_w; const float* conv_b; const float* dense_w; /* [K,F], [F], [F,U] */
  float* loc_ws;                /* mode 2: fp32 scratch, os2s_attn_decoder_loc_ws_floats(B, S, U, loc_k) floats:
                                 * the folded location filter [(K+1)*U] + partial scores [B, 4, S] */
  /* inputs */
  const uint16_t* gx0;          /* bf16 [B, T, 4H] */
  const uint16_t* keys;         /* bf16 [B, S, U] */
  const uint16_t* values;       /* bf16 [B, S, M], zero past src_len */
  const int32_t* src_len; const int32_t* tgt_len;
  /* state / saved sequences / outputs */
  uint16_t* cat[2]; float* c_seq[2]; uint16_t* gates[2];
  float* cum_seq; float* align_seq; float* q_seq;
  uint16_t* y_top; long long y_top_bs, y_top_ts;
  uint16_t* ctx; long long ctx_bs, ctx_ts;
  /* optional e4m3 copies of wcat[l] (os2s_quantize_rows_e4m3) with one fp32 scale per row: when
   * non-NULL the FORWARD cell kernels stream these instead of the bf16 weights (half the bytes
   * per time step); activations, accumulation and the backward pass are unchanged */
  const uint8_t* wcat8[2];
  const float* wcat8_scale[2];
} os2s_attn_decoder_t;

/* Backward through all T steps (t_begin = 0, t_end = T). Inputs: dy_top / dctx_ext = gradients
 * of the y_top / ctx rows (either may be NULL), wcatT[l] = bf16 [Kc_l, 4H] and wqT = bf16
 * [H, U] transposed weights. Outputs: dg[l] bf16 [B,T,4H] gate gradients (caller: dWcat_l = dg_l^T cat_l,
 * d gx0 = dg_0), dq_seq bf16 [B,T,U] (dWq = dq^T y_top, db = column sums), dkeys fp32
 * [B,S,U], dmem bf16 [B,S,M] = gradient of `values`; dctx_seq bf16 [B,T,M] and dpre_seq bf16
 * [B,T,S,U] (per-step score gradients, reduced over T into dkeys after the loop) are scratch; the
 * small score parameters are ACCUMULATED into dv [U], dg_scalar [1], dconv_w, dconv_b,
 * ddense_w (fp32). */
typedef struct os2s_attn_decoder_grads {
  const uint16_t* wcatT[2];
  const uint16_t* wqT;          /* bf16 [H, U] transposed query layer */
  const uint16_t* dy_top; long long dy_top_bs, dy_top_ts;
  const uint16_t* dctx_ext; long long dctx_bs, dctx_ts;
  uint16_t* dg[2];
  uint16_t* dq_seq; uint16_t* dctx_seq; uint16_t* dpre_seq; float* dkeys; uint16_t* dmem;
  float* dv; float* dg_scalar; float* dconv_w; float* dconv_b; float* ddense_w;
} os2s_attn_decoder_grads_t;

/* ------------------------------------------------------------------------
 * Free-running Tacotron2 decoding: Tacotron2Decoder._decode in eval / infer mode
 * (decoders/tacotron2_decoder.py:378-428) = dynamic_decode(TacotronDecoder(TacotronHelper),
 * impute_finished = False, maximum_iterations = 10 * max(src_len)); per step
 * (parts/tacotron/tacotron_decoder.py:153-190, tacotron_helper.py:138-226):
 *   x_t = prenet(frame_{t-1})  (2 x Dense + ReLU + dropout(prenet_keep), on in every mode; frame_{-1} = 0)
 *   LSTM stack on [x_t, attention_{t-1}] -> location-sensitive attention (query = top cell output)
 *   frame_t = W_out [h_top, attention_t] + b;  stop_t = W_stop frame_t + b
 *   finished |= stop_t > 0  (round(sigmoid) with mask_decoder_sequence); ends when every sample finished.
 * `loop` is the attention cell exactly as os2s_attn_decoder_fwd takes it (score_mode 2, parameters, keys /
 * values / src_len, zeroed sequence buffers, T = step capacity; gx0 is not read: the layer-0 input
 * projection is the first P columns of w0x; tgt_len must be NULL, attn_in_keep = out_keep = 1).
 * One step is four launches with no host interaction; the stop decision is device-resident:
 *   state int32 [4 + 2B], zeroed by the caller: [1] = number of steps after which every sample had
 *   finished (0 while running; launches enqueued past that point return at once), [2] finished count,
 *   [4, 4+B) finished flags, [4+B, 4+2B) sequence lengths (the step that raised the stop token counts).
 * So the caller may enqueue any number of steps ahead and poll state[1] at its own interval.
 * Outputs: mel bf16 [B, T, n_mel] (decoder frames), stop fp32 [B, T] (logits), loop->align_seq,
 * loop->y_top / ctx rows. os2s_tacotron_infer_steps with t_begin == 0 also prepares step 0.
 * Returns OS2S_ERR_UNSUPPORTED for shapes the step kernels are not built for (B > 32, H / M / P not
 * multiples of 64, H > 1024, P > 256, n_mel > 128, ...): drive os2s_attn_decoder_fwd step by step then.
 * ---------------------------------------------------------------------- */
typedef struct os2s_tacotron_infer {
  const os2s_attn_decoder_t* loop;
  int P, n_mel;                    /* pre-net units, frame size */
  int mask_decoder_sequence;
  float prenet_keep; unsigned long long prenet_seed[2];
  /* layer 0 with the pre-net columns in front: [4H, P + M + H] = [kernel rows of the inputs | attention | state];
   * exactly one of w0x (bf16) / w0x8 (e4m3 + per-row scales; then loop->wcat8[1] serves layer 1) */
  const uint16_t* w0x; const uint8_t* w0x8; const float* w0x8_scale;
  const float* bias0;              /* [4H] */
  const uint16_t* wp1; const float* bp1;   /* pre-net layer 1: bf16 [P, n_mel], fp32 [P] */
  const uint16_t* wp2; const float* bp2;   /* pre-net layer 2: bf16 [P, P], fp32 [P] */
  const uint16_t* wout_h;          /* bf16 [n_mel, H]: output-projection columns of the cell output */
  /* transposed operands of the two attention-weighted sums, positions contiguous, rows padded with zeros to
   * Sp = S rounded up to 32 (caller GEMM + transposes, once per batch): */
  const uint16_t* pv_t;            /* bf16 [B, n_mel rounded up to 16, Sp] = (values W_out[:, H:]^T)^T */
  const uint16_t* values_t;        /* bf16 [B, M, Sp] = values^T */
  const float* bout;               /* [n_mel] */
  const uint16_t* wstop; const float* bstop;   /* bf16 [n_mel], fp32 [1] */
  float* mh;                       /* fp32 [B, n_mel] scratch (the cell-output half of the frame of the step) */
  uint16_t* x_seq;                 /* bf16 [B, T+1, P]: pre-net outputs, row t = input of step t */
  uint16_t* mel;                   /* bf16 [B, T, n_mel] */
  float* stop;                     /* fp32 [B, T] */
  int32_t* state;                  /* int32 [os2s_tacotron_infer_state_ints(B)] */
} os2s_tacotron_infer_t;
size_t os2s_tacotron_infer_state_ints(int B);
int os2s_tacotron_infer_supported(const os2s_tacotron_infer_t* x);
int os2s_tacotron_infer_steps(os2s_stream_t stream, const os2s_tacotron_infer_t* x, int t_begin, int t_end);

/* fp8 weight storage (BASELINE.json configs[4]: Tacotron2 decode with fp8 weights; the reference has
 * no fp8 — models/model.py:88 — so the policy is this library's): q[r,k] = e4m3(w[r,k] / scale[r]),
 * scale[r] = max_k |w[r,k]| / 448, OCP e4m3fn. w bf16 [rows, K], K % 8 == 0. */
int os2s_quantize_rows_e4m3(os2s_stream_t stream, const uint16_t* w, int rows, int K, uint8_t* q,
                            float* scale);
int os2s_attn_decoder_fwd(os2s_stream_t stream, const os2s_attn_decoder_t* d);
/* floats of os2s_attn_decoder_t.loc_ws (location-sensitive mode) */
size_t os2s_attn_decoder_loc_ws_floats(int B, int S, int U, int loc_k);
size_t os2s_attn_decoder_bwd_workspace_bytes(const os2s_attn_decoder_t* d);
int os2s_attn_decoder_bwd(os2s_stream_t stream, const os2s_attn_decoder_t* d,
                          const os2s_attn_decoder_grads_t* grads, void* workspace,
                          size_t workspace_bytes);

/* ------------------------------------------------------------------------
 * TTS spectrogram features: get_speech_features (data/text2speech/speech_utils.py:98-182) =
 * librosa.stft(y, n_fft) [hop = n_fft/4 by default, periodic Hann(n_fft) passed in `window`,
 * centre=True with reflect padding] -> |X|^mag_power -> out_mag[b,t,:n_mag] =
 * log(clip(mag, data_min_mag)) and out_mel[b,t,:] = log(clip(mel_basis . mag, data_min_mel)).
 * signal fp32 [B, sig_stride], n_samples [B]; frames t >= 1 + n_samples/hop are filled with
 * pad_mel / pad_mag. The mel basis is passed in compact form (per filter: first bin, length,
 * weights mel_wt[i * n_mels + m]). Either output may be NULL. fp32 [B, T, *].
 * ---------------------------------------------------------------------- */
int os2s_tts_spectrogram(os2s_stream_t stream, const float* signal, long long sig_stride,
                         const int32_t* n_samples, const float* window, int B, int n_fft, int hop,
                         int T, int mag_power, float data_min_mag, float data_min_mel, int n_mag,
                         int n_mels, const int32_t* mel_start, const int32_t* mel_len,
                         const float* mel_wt, int mel_maxlen, float* out_mel, float* out_mag,
                         float pad_mel, float pad_mag);

/* ------------------------------------------------------------------------
 * Depthwise half of tf.layers.separable_conv1d (layer type "sep_conv1d",
 * parts/cnns/conv_blocks.py:11-16; QuartzNet): y[b,t,c] = sum_k x[b, t*stride + k*dil -
 * padL, c] * w[k,c]; x, y bf16 channels-last, w fp32 [K, C] (TF depthwise_kernel [K, C, 1]),
 * x rows >= in_len[b] read as zero, output tiles past out_len[b] are skipped (may be NULL).
 * flip_taps = 1 applies w[K-1-k] (the data gradient: call with x = dz, padL' = (K-1)*dil -
 * padL, stride 1). os2s_depthwise_conv1d_wgrad ACCUMULATES dw[k,c] += sum_{b,t} dy[b,t,c] *
 * x[b, t*stride + k*dil - padL, c] (fp32 atomics). The pointwise half is os2s_conv1d_fwd, K = 1.
 * ---------------------------------------------------------------------- */
int os2s_depthwise_conv1d_fwd(os2s_stream_t stream, const uint16_t* x, const float* w, uint16_t* y,
                              const int32_t* in_len, const int32_t* out_len, int B, int Tin,
                              int Tout, int C, int K, int stride, int dil, int padL,
                              int flip_taps);
int os2s_depthwise_conv1d_wgrad(os2s_stream_t stream, const uint16_t* x, const uint16_t* dy,
                                float* dw, const int32_t* in_len, int B, int Tin, int Tout, int C,
                                int K, int stride, int dil, int padL);
/* The data gradient of a stride-1 / dilation-1 depthwise convolution (2 <= K <= 96) as the LAST contribution to
 * the gradient of the output of a conv + BatchNorm + ReLU (+ dropout) layer (conv_bn_actv,
 * parts/cnns/conv_blocks.py:170-232, followed by a sep_conv1d layer), with that layer's activation backward and
 * the partial sums of its BatchNorm backward in the kernel's store phase (the depthwise twin of
 * os2s_conv1d_dgrad_bnact_ws):
 *   dx = mask(conv(dz, flipped taps) + addend),  mask = (mask_ref > 0) ? mask_scale : 0   (addend may be NULL or dx)
 *   stats[part, 0, c] = sum over the part's rows of dx,  stats[part, 1, c] = sum of dx * stat_ref
 * dz [B, Tin, C], dx / addend / mask_ref / stat_ref [B, Tout, C] bf16; stats fp32
 * [os2s_depthwise_dgrad_bnact_num_parts(B, Tout, K), 2, C], ZERO on entry (tiles past out_len are not visited);
 * os2s_bn_bwd_finalize_raw turns them into dgamma / dbeta / c1 / c2. padL = (K - 1) - the forward padding. */
int os2s_depthwise_dgrad_bnact_num_parts(int B, int Tout, int K);
int os2s_depthwise_dgrad_bnact(os2s_stream_t stream, const uint16_t* dz, const float* w, uint16_t* dx,
                               const uint16_t* addend, float* stats, const int32_t* out_len, int B, int Tin,
                               int Tout, int C, int K, int padL, const uint16_t* mask_ref, float mask_scale,
                               const uint16_t* stat_ref);

/* A separable layer with ONE tap — the residual branches of a separable block (parts/cnns/conv_blocks.py:66,79-85:
 * tf.layers.separable_conv1d with kernel_size 1) — is a per-channel scale d [Cin] in front of the pointwise
 * convolution W [Cout, Cin]: y = x (W diag d)^T. os2s_pointwise_fold writes the bf16 operands of the 1x1 kernels,
 *   w_eff [Cout, Cin] = W diag(d)   (forward),      wt_eff [Cin, Cout] = its transpose   (data gradient),
 * from the fp32 variables; os2s_pointwise_fold_bwd splits G = dy^T x [Cout, Cin] (os2s_conv1d_wgrad_ws on the
 * layer's INPUT) into the two variables' gradients, accumulated:
 *   dw[co, ci] += G[co, ci] d[ci],      dd[ci] += sum_co W[co, ci] G[co, ci]          (one writer per element) */
int os2s_pointwise_fold(os2s_stream_t stream, const float* w, const float* d, uint16_t* w_eff, uint16_t* wt_eff,
                        int cout, int cin);
int os2s_pointwise_fold_bwd(os2s_stream_t stream, const float* g, const float* w, const float* d, float* dw,
                            float* dd, int cout, int cin);

/* ------------------------------------------------------------------------
 * Text2SpeechLoss terms (losses/text2speech_loss.py:35-209). One call per term:
 *   mode 0: tf.losses.mean_squared_error, 1: absolute_difference (l1_norm), both with
 *           weights = sequence_mask(lens) and SUM_BY_NONZERO_WEIGHTS: sum / (F * sum_b len_b)
 *   mode 2: masked tf.nn.sigmoid_cross_entropy_with_logits / sum(mask)   (stop token, F = 1)
 * pred bf16 rows (b,t) at pred + (b*T+t)*ld_pred (F columns used), target fp32 likewise;
 * loss[0] += weight * term; dpred (same layout as pred, may be NULL) = weight * (*grad_scale_dev)
 * * d term / d pred. partial: fp32 [os2s_tts_loss_num_parts(B,T)] scratch. lens NULL = no mask.
 * ---------------------------------------------------------------------- */
int os2s_tts_loss_num_parts(int B, int T);
int os2s_tts_loss(os2s_stream_t stream, const uint16_t* pred, long long ld_pred,
                  const float* target, long long ld_target, const int32_t* lens, int B, int T,
                  int F, int mode, float weight, const float* grad_scale_dev, float* partial,
                  float* loss, uint16_t* dpred);
/* The same terms when prediction and target have different row counts per sample (eval / infer: the
 * free-running decoder stops on its own stop token; text2speech_loss.py:80-131): both are padded to
 * T = max(T_pred, T_target) — predictions with zeros, targets with target_pad (0 for spectrogram
 * rows, 1 for the stop token, :100-101) — and the mask is sequence_mask(lens, T), so a live target row
 * the decoder never produced is compared with zeros. pred rows (b,t) at pred + (b*T_pred+t)*ld_pred,
 * target rows at target + (b*T_target+t)*ld_target; dpred (layout of pred) gets no rows past T_pred;
 * partial: [os2s_tts_loss_num_parts(B, max(T_pred, T_target))]. */
int os2s_tts_loss_padded(os2s_stream_t stream, const uint16_t* pred, long long ld_pred, int T_pred,
                         const float* target, long long ld_target, int T_target, float target_pad,
                         const int32_t* lens, int B, int F, int mode, float weight,
                         const float* grad_scale_dev, float* partial, float* loss, uint16_t* dpred);
/* tf.exp on the magnitude branch (decoders/tacotron2_decoder.py:541-542) and its gradient
 * helper y = a * b; out[b,c] (+)= sum_t x[b,t,c] (gradient of a vector tiled over time:
 * the style embedding, encoders/tacotron2_encoder.py:168-172). n % 8 == 0. */
int os2s_exp_fwd(os2s_stream_t stream, const uint16_t* x, long long n, uint16_t* y);
/* activation=tf.nn.tanh of a tf.layers.dense (reference_activation, tacotron2_encoder.py:448-456):
 * y = tanh(x); dx = dy * (1 - y^2). */
int os2s_tanh_fwd(os2s_stream_t stream, const uint16_t* x, long long n, uint16_t* y);
int os2s_tanh_bwd(os2s_stream_t stream, const uint16_t* dy, const uint16_t* y, long long n,
                  uint16_t* dx);
int os2s_mul_bf16(os2s_stream_t stream, const uint16_t* a, const uint16_t* b, long long n,
                  uint16_t* y);
int os2s_sum_time(os2s_stream_t stream, const uint16_t* x, long long ld, int B, int T, int C,
                  float* out, int accumulate);

/* ------------------------------------------------------------------------
 * Global style tokens (Tacotron2Encoder._embed_style, encoders/tacotron2_encoder.py:341-505).
 * os2s_gru_tf_*: tf.nn.rnn_cell.GRUCell under dynamic_rnn(sequence_length) (:397-417):
 *   [r,u] = sigmoid(x Wg_x + h Wg_h + bg), c = tanh(x Wc_x + (r*h) Wc_h + bc), h' = u h + (1-u) c.
 *   gxg [B,T,2H] / gxc [B,T,H] bf16 hold the input parts (+ biases; caller GEMMs); wgh [H,2H],
 *   wch [H,H] fp32 are the state rows of the TF kernels ([in,out]); h_final [B,H] fp32 is the
 *   state at each sample's last valid step. Saved for backward: h_seq [B,T+1,H], r/u/c_seq
 *   [B,T,H] fp32, hprev16 / rh16 [B,T,H] bf16 (h_{t-1} and r*h_{t-1}: operands of the
 *   recurrent-weight gradient GEMMs dWg_h = hprev^T dgxg, dWc_h = rh^T dgxc).
 *   Backward: dh_final -> dgxg, dgxc (bf16); wghT [2H,H], wchT [H,H] transposed kernels.
 * os2s_gst_attention_*: multi-head attention in "bahdanau" mode over N <= 64 style tokens
 *   (parts/transformer/attention_layer.py:171-186): per head (depth 64)
 *   w = softmax_n sum_d tanh(att_v[d] tanh(k[n,d] + q[b,d])), out = sum_n w_n v[n].
 *   q/out/dq [B, heads*64] bf16, k/v [N, heads*64] bf16, w [B,heads,N] fp32; the backward
 *   ACCUMULATES dk, dv [N, heads*64] and datt_v [64] (fp32, atomics).
 * ---------------------------------------------------------------------- */
int os2s_gru_tf_fwd(os2s_stream_t stream, const uint16_t* gxg, const uint16_t* gxc,
                    const float* wgh, const float* wch, const int32_t* lens, int B, int T, int H,
                    float* h_seq, float* r_seq, float* u_seq, float* c_seq, uint16_t* hprev16,
                    uint16_t* rh16, float* h_final);
int os2s_gru_tf_bwd(os2s_stream_t stream, const float* dh_final, const float* wghT,
                    const float* wchT, const int32_t* lens, int B, int T, int H,
                    const float* h_seq, const float* r_seq, const float* u_seq, const float* c_seq,
                    uint16_t* dgxg, uint16_t* dgxc);
int os2s_gst_attention_fwd(os2s_stream_t stream, const uint16_t* q, const uint16_t* k,
                           const uint16_t* v, const float* att_v, int B, int heads, int N,
                           uint16_t* out, float* w);
int os2s_gst_attention_bwd(os2s_stream_t stream, const uint16_t* dout, const uint16_t* q,
                           const uint16_t* k, const uint16_t* v, const float* att_v, const float* w,
                           int B, int heads, int N, uint16_t* dq, float* dk, float* dv,
                           float* datt_v);

/* ------------------------------------------------------------------------
 * conv2d (time x frequency) of DeepSpeech2 (tf.layers.conv2d in conv_bn_actv,
 * encoders/ds2_encoder.py:252-266) on the 1-D implicit-GEMM kernel: activations are
 * flattened to [B, T, F*C]; the frequency convolution becomes the banded channel mixing
 *   wexp[kt][(fo,co)][(fi,ci)] = w[kt][fi - fo*sF + padF][ci][co]
 * (w: fp32 master kernel [KT,KF,Cin,Cout], TF layout). expand builds wexp (bf16,
 * [KT][Fo*Cout][Fi*Cin] = the os2s_conv1d_fwd weight layout); reduce folds the gradient
 * of wexp (fp32, from os2s_conv1d_wgrad) back: dw += ... .
 * ---------------------------------------------------------------------- */
int os2s_conv2d_toeplitz_expand(os2s_stream_t stream, const float* w, int KT, int KF, int Cin,
                                int Cout, int Fi, int Fo, int sF, int padF, uint16_t* wexp);
int os2s_conv2d_toeplitz_reduce(os2s_stream_t stream, const float* dwexp, int KT, int KF, int Cin,
                                int Cout, int Fi, int Fo, int sF, int padF, float* dw);

/* ------------------------------------------------------------------------
 * CTC prefix beam search with a word n-gram language model — HOST entry
 * points: every pointer is host memory, no stream. Replaces the reference's only native code on
 * this path, the CPU TensorFlow op CTCBeamSearchDecoderWithLM
 * (ctc_decoder_with_lm/beam_search.cc:452-804; Step :245-380, TopPaths :398-429, end-of-sequence
 * rescoring :730-739) with WordLMBeamScorer (ctc_decoder_with_lm/beam_search.h:32-217), as bound
 * by FullyConnectedCTCDecoder.decode_with_lm (open_seq2seq/decoders/fc_decoders.py:206-235).
 *
 * os2s_ctc_scorer_create: lm_path = an ARPA text file (any order <= 8), or a KenLM binary
 *   (format version 5) in one of the two layouts the reference ships a sample of:
 *     - model type 5, the type the reference's op loads (QuantArrayTrieModel, beam_search.h:22),
 *       order 2 (ctc-test-lm.binary);
 *     - model type 0 (probing hash tables, kenlm's default; order <= 8): after the common header
 *       and n-gram counts, the vocabulary {u32 version, u32 bound} + B(count[0]) packed 12-byte
 *       buckets {u64 MurmurHash64A(word), u32 id}, B(n) = max(n + 1, (u64)(multiplier * n));
 *       (count[0] + 1) unigrams {f32 prob, f32 backoff}; per middle order B(count) buckets
 *       {u64 key, f32 prob, f32 backoff}; highest order B(count) packed {u64 key, f32 prob};
 *       key(w1..wn) = combine(..combine(combine(wn, wn-1), wn-2).., w1), combine(c, w) =
 *       c * 8978948897894561157 ^ (1 + w) * 17894857484156487943; empty bucket = key 0; the sign
 *       bit of a stored probability is a flag; word strings follow in id order
 *       (toy_speech_data/toy_data-lm.binary);
 *   other binaries return OS2S_ERR_UNSUPPORTED (use the ARPA file they were built from). trie_path = the text
 *   letter trie written by generate_trie (trie_node.h:46-82); alphabet_path = one label per line
 *   (alphabet.h:24-40), C - 1 labels, blank = C - 1 is implied. alpha weighs log10 P(word |
 *   history), beta is the per-word bonus, trie_weight weighs the letter-prefix score.
 * os2s_ctc_scorer_ngram_score: log10 P(words[n-1] | words[..n-2]) exactly as ScoreNGram
 *   (beam_search.h:172-200) pads / truncates the history; -100 for an out-of-vocabulary word.
 * os2s_ctc_beam_search: logits fp32, element (t, b, c) at logits[t*ld_t + b*ld_b + c] (raw
 *   logits; each frame is log-softmax normalised inside, beam_search.cc:262-271); seq_len [B];
 *   scorer may be NULL (plain tf.nn.ctc_beam_search_decoder semantics). n_threads <= 0: one per
 *   host core, capped at B. Outputs: out_ids [B, top_paths, T] int32 padded with -1,
 *   out_len [B, top_paths], out_log_prob [B, top_paths] (natural-log beam scores, best first).
 * ---------------------------------------------------------------------- */
int os2s_ctc_scorer_create(const char* lm_path, const char* trie_path, const char* alphabet_path,
                           float alpha, float beta, float trie_weight, void** scorer);
void os2s_ctc_scorer_destroy(void* scorer);
/* SetAlpha / SetBeta / SetTrieWeight of the scorer (beam_search.h:150-160): re-weight without
 * re-reading the model (grid search over alpha, beta: scripts/decode.py). Not thread-safe
 * against a running os2s_ctc_beam_search on the same scorer. */
int os2s_ctc_scorer_set_weights(void* scorer, float alpha, float beta, float trie_weight);
int os2s_ctc_scorer_ngram_score(const void* scorer, const char* const* words, int n_words,
                                float* log10_prob);
/* The reference's generate_trie tool (ctc_decoder_with_lm/generate_trie.cpp:32-64): one
 * TrieNode::Insert per whitespace-separated word of vocab_path with its unigram log10
 * probability from the null context; writes the text trie the scorer reads. */
int os2s_ctc_generate_trie(const char* alphabet_path, const char* lm_path, const char* vocab_path,
                           const char* trie_path);
int os2s_ctc_beam_search(const float* logits, long long ld_t, long long ld_b,
                         const int32_t* seq_len, int T, int B, int C, int beam_width,
                         int top_paths, int merge_repeated, const void* scorer, int n_threads,
                         int32_t* out_ids, int32_t* out_len, float* out_log_prob);

/* ------------------------------------------------------------------------
 * The reference's second CTC decoder — the `ctc_decoders` module of decoders/ (swig wrapper used
 * offline by scripts/decode.py): prefix beam search over softmax PROBABILITIES with an external
 * scorer and a dictionary constraint. HOST entry points (host pointers, no stream).
 * Replaces Scorer(alpha, beta, model_path, vocabulary) (decoders/scorer.cpp:16-52; reset_params,
 * is_character_based, get_max_order, get_dict_size) and ctc_beam_search_decoder(_batch)
 * (decoders/ctc_beam_search_decoder.cpp:18-178, 420-459): vocabulary = the n_vocab labels without
 * the blank (blank = C - 1); the dictionary is every language-model word spellable with the
 * vocabulary; lm_path as for os2s_ctc_scorer_create. probs element (t, b, c) at
 * probs[t*ld_t + b*ld_b + c]; cutoff_prob / cutoff_top_n prune the labels of a frame
 * (decoder_utils.cpp:7-37; the reference's defaults are 1.0 and 40). scorer may be NULL.
 * Outputs as os2s_ctc_beam_search: out_ids [B, top_paths, T] padded with -1, out_len, out_score
 * (natural-log acoustic score + alpha * log10 LM + beta per word, as the reference mixes them).
 * ---------------------------------------------------------------------- */
int os2s_ctc_dict_scorer_create(const char* lm_path, const char* const* vocabulary, int n_vocab,
                                double alpha, double beta, void** scorer);
void os2s_ctc_dict_scorer_destroy(void* scorer);
int os2s_ctc_dict_scorer_set_weights(void* scorer, double alpha, double beta);
int os2s_ctc_dict_scorer_info(const void* scorer, int* is_character_based, int* max_order,
                              int* dict_size);
int os2s_ctc_dict_beam_search(const float* probs, long long ld_t, long long ld_b,
                              const int32_t* seq_len, int T, int B, int C, int beam_size,
                              double cutoff_prob, int cutoff_top_n, int top_paths,
                              const void* scorer, int n_threads, int32_t* out_ids,
                              int32_t* out_len, float* out_score);

/* ------------------------------------------------------------------------
 * Per-sample normalisations of the TDNN encoder (csrc/sample_norm.hip): tf.contrib.layers.layer_norm and
 * tf.contrib.layers.instance_norm as conv_ln_actv / conv_in_actv apply them to a channels-last convolution
 * output (open_seq2seq/parts/cnns/conv_blocks.py:234-309; selected by TDNNEncoder's `normalization`
 * 'layer_norm' / 'instance_norm', encoders/tdnn_encoder.py:144-156).
 *   x, z, dz, dx  [B, T, C] bf16 (C even); gamma, beta, dgamma, dbeta [C] fp32; mean, rstd [B, C] fp32
 *   mode 0 = instance norm: statistics per (sample, channel) over the T frames of the padded tensor
 *            (the reference normalises the padded tensor), epsilon 1e-6 in the reference
 *   mode 1 = layer norm with the defaults the reference leaves in place (begin_norm_axis = 1,
 *            begin_params_axis = -1): statistics per sample over all T x C values, epsilon 1e-12
 * fwd: z = gamma * (x - mean) * rstd + beta, mean / rstd saved for bwd. bwd: dx, and dgamma / dbeta
 * ACCUMULATED (+=). partial: os2s_sample_norm_partial_floats(B, C) floats; scratch (bwd): 4 * B * C floats.
 * Deterministic (fixed-order reductions). Activation, dropout and the sequence mask are the caller's next
 * pass (os2s_bn_act_fwd with scale 1 / shift 0).
 * ---------------------------------------------------------------------- */
size_t os2s_sample_norm_partial_floats(int B, int C);
int os2s_sample_norm_fwd(os2s_stream_t stream, const uint16_t* x, const float* gamma, const float* beta,
                         int B, int T, int C, int mode, float eps, uint16_t* z, float* mean, float* rstd,
                         float* partial);
int os2s_sample_norm_bwd(os2s_stream_t stream, const uint16_t* dz, const uint16_t* x, const float* gamma,
                         const float* mean, const float* rstd, int B, int T, int C, int mode, uint16_t* dx,
                         float* dgamma, float* dbeta, float* partial, float* scratch);

/* y[b, t * stride, :] = x[b, t, :] with every other row of y [B, Tup, C] zero (Tup >= (T - 1) * stride + 1, C a
 * multiple of 8): the zero-upsampled output gradient of a STRIDED tf.layers.conv1d past the first layer
 * (parts/cnns/conv_blocks.py:195-206 with strides > 1) — its data gradient is the stride-1 data gradient of
 * the upsampled tensor (os2s_conv1d_fwd on the tap-flipped weights). */
int os2s_upsample_rows_bf16(os2s_stream_t stream, const uint16_t* x, int B, int T, int C, int stride, int Tup,
                            uint16_t* y);

/* ------------------------------------------------------------------------
 * Dense-residual block ends without branch tensors (csrc/dense_residual.hip).
 * conv_bn_res_bn_actv (parts/cnns/conv_blocks.py:61-168) adds, at the end of block k, one
 * tf.layers.conv1d(kernel_size=1) + tf.layers.batch_normalization per dense-residual input (the inputs of
 * blocks 0 .. k: encoders/tdnn_encoder.py:188-192) to the main branch. Each branch output is a linear image of
 * its input r_i, so its batch statistics follow from the input's column sums s_i and Gram matrix
 * G_i = r_i^T r_i, the sum of all branches is ONE GEMM of the channel-concatenated inputs with the BN-scaled,
 * stacked kernels, and every gradient follows from P_k = [r_0 .. r_k]^T dz_k (formulas: the header of
 * csrc/dense_residual.hip). These entry points are the small kernels between the GEMMs; the GEMMs themselves
 * are os2s_conv1x1_cat_fwd, os2s_conv1x1_wgrad_grouped_ws, os2s_conv1x1_fwd_grouped_ex and os2s_gemm_nt_ws.
 * Results equal the branch-by-branch evaluation up to rounding (no bf16 branch tensor is rounded on the way).
 * ---------------------------------------------------------------------- */
/* dst[b, t, 0:C] = t < lens[b] ? src[b, t, 0:C] : 0 for a [B, T, C] bf16 tensor into a channel slice of a wider
 * one (row strides in elements, multiples of 8; lens NULL = every row); colsum_partial (or NULL):
 * [os2s_dres_copy_num_parts(B, T), C] fp32 column sums of the copied rows per 128-row block. */
int os2s_dres_copy_num_parts(int B, int T);
int os2s_dres_copy_cols(os2s_stream_t stream, const uint16_t* src, long long src_row_stride, uint16_t* dst,
                        long long dst_row_stride, const int32_t* lens, int B, int T, int C, float* colsum_partial);
/* s[c] = sum of the partials, m = s / count, and the covariance C = gram / count - m m^T as a bf16 pair:
 * chl [2C, C]: rows 0 .. C-1 = bf16(C), rows C .. 2C-1 = bf16(C - hi). gram [C, C] fp32 = r^T r. */
int os2s_dres_cov(os2s_stream_t stream, const float* colsum_partial, int nparts, const float* gram, int C,
                  long long count, float* s, float* m, uint16_t* chl);
/* One residual branch of a block end. Arrays of these live in DEVICE memory (built once per model: every pointer
 * is a parameter, a gradient or a persistent weight-sized buffer). */
typedef struct {
  const uint16_t* w;       /* [Cout, c] bf16: the branch's 1x1 kernel (device layout [1, Cout, c]) */
  const float* tt;         /* [Cout, 2c] fp32: w . [C_hi | C_lo]^T of the branch's input (training) */
  const float* m;          /* [c] channel means of the input (training) */
  const float* s;          /* [c] channel sums of the input (backward) */
  const float* gamma;      /* [Cout] */
  const float* beta;       /* [Cout] */
  float* moving_mean;      /* [Cout] updated in training, read otherwise */
  float* moving_var;       /* [Cout] */
  float* mean;             /* [Cout] batch statistics: written by os2s_dres_bn_fwd, read by os2s_dres_bn_bwd */
  float* rstd;             /* [Cout] */
  float* dgamma;           /* [Cout] accumulated into */
  float* dbeta;            /* [Cout] accumulated into */
  float* dw;               /* [Cout, c] fp32 kernel gradient, accumulated into */
  uint16_t* wd1;           /* the input's stacked transposed matrices, rows = the input's channels (+ 8 for wd2), */
  uint16_t* wd2;           /*   row stride ld, already offset to this block end's columns: wd1[a][co] = w d1,     */
  uint16_t* wt;            /*   wd2[a][co] = -w d2 (row c: the constant-row coefficients), wt[a][co] = w          */
  long long ld;
  int c, koff;             /* input channels (multiple of 64); first channel of the input in the concatenation */
} os2s_dres_seg_t;
/* Forward of block end k over its nseg <= 16 branches (ascending koff): batch (training = 1: from m and tt; the
 * moving statistics are updated with TF's conventions, see os2s_bn_finalize) or moving statistics -> mean / rstd,
 * wp [Cout, Kk] = the kernels scaled by gamma rstd, stacked along the concatenated input channels, and
 * shift [Cout] = sum over the branches of beta - mean gamma rstd. count = B * T. */
int os2s_dres_bn_fwd(os2s_stream_t stream, const os2s_dres_seg_t* segs_dev, int nseg, int Cout, int Kk,
                     uint16_t* wp, float* shift, long long count, float eps, float momentum, int training);
/* Backward of block end k given P [Cout, Kk] fp32 = dz^T [r_0 .. r_k] and mean_dz [Cout] = sum_rows dz / count:
 * dgamma, dbeta, dw of every branch (accumulated) and the branch's columns of wd1 / wd2 / wt. coef: scratch of
 * nseg * 4 * Cout floats. Kk and every c multiples of 64, Cout a multiple of 8. Deterministic. */
int os2s_dres_bn_bwd(os2s_stream_t stream, const os2s_dres_seg_t* segs_dev, int nseg, int Cout, int Kk,
                     const float* P, const float* mean_dz, long long count, float* coef);
/* 1x1 convolution between channel slices of wider tensors over a ragged batch:
 *   y[b, t, 0:Cout] (+)= x[b, t, 0:Cin] . w^T (+ bias),  x rows x_row_stride apart, y rows y_row_stride apart
 * (elements, multiples of 8; batch strides = T rows), w [Cout, Cin] bf16 contiguous. in_len / out_len as in
 * os2s_conv1d_fwd_ws. The sum of a block end's residual branches (x = the concatenated block inputs, w = wp of
 * os2s_dres_bn_fwd) and the data gradients of a dense-residual input (x = the concatenated dz of the block ends
 * that read it, w = wd1). Runs on the 256 x 256 ping-pong tile over the live windows when Cin % 64 == 0 and
 * B <= 64, else on the lockstep tile. Workspace: os2s_conv1d_workspace_bytes() (contract of os2s_conv1d_fwd_ws). */
int os2s_conv1x1_cat_fwd(os2s_stream_t stream, const uint16_t* x, long long x_row_stride, const uint16_t* w,
                         uint16_t* y, long long y_row_stride, const int32_t* in_len, const int32_t* out_len,
                         const float* bias, int B, int T, int Cin, int Cout, int accumulate, void* workspace,
                         size_t workspace_bytes);
/* os2s_conv1x1_fwd_grouped with fp32 outputs when out_f32 = 1 (y_i [B, T, Cout_i] fp32; no statistics): the
 * products w . [C_hi | C_lo]^T of a block end's branches in one launch (x_i = the kernel [1, Cout, c_i] read as
 * Cout rows, w_i = chl_i [2 c_i, c_i]). */
int os2s_conv1x1_fwd_grouped_ex(os2s_stream_t stream, const os2s_conv_group_t* groups, int ngroups,
                                const int32_t* in_len, const int32_t* out_len, int B, int T, int out_f32);

#ifdef __cplusplus
}
#endif
#endif /* OS2S_H_ */
