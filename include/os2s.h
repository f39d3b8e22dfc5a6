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

#ifdef __cplusplus
}
#endif
#endif /* OS2S_H_ */
