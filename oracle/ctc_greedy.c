/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path
 * (openseq2seq_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this.
 *
 * Plain-C restatement of CTC greedy (best-path) decoding as the reference uses it:
 *   - tf.nn.ctc_greedy_decoder(logits[T,B,V], seq_len, merge_repeated=True)
 *     called from open_seq2seq/decoders/fc_decoders.py:244-251 (time-major fp32
 *     logits, blank = V-1 per open_seq2seq/models/speech2text.py:123-125);
 *   - decoders/ctc_greedy_decoder.cpp:19-43 (per-frame argmax with strict '<'
 *     comparison => first maximum wins; emit when the frame's id differs from
 *     the previous frame's id; drop blanks afterwards).
 * The TF op additionally returns neg_sum_logits = -sum_t max_v logits[t,b,v]
 * (pinned by ctc_decoder_with_lm/ctc-test.py:66: -7079.117 +- 1e-3).
 *
 * Parity pin: checked against the golden vector ctc_decoder_with_lm/ctc-test.pickle
 * ('then seconds', tests/golden/ctc_test_logits.npy) and against the compiled
 * reference C++ (oracle/_ref/libref_ctc_greedy.so) in tests/test_ctc_greedy_oracle.py.
 */
#include <stdint.h>
#include <stddef.h>

int oracle_ctc_greedy_decode(const float* logits, const int32_t* seq_len, int T,
                             int B, int V, int blank, int merge_repeated,
                             int32_t* out_ids /* [B,T], -1 padded */,
                             int32_t* out_len /* [B] */,
                             float* neg_sum_logits /* [B] or NULL */) {
  for (int b = 0; b < B; ++b) {
    int len = seq_len[b];
    if (len < 0) len = 0;
    if (len > T) len = T;
    int n = 0;
    int prev = -1;
    float score = 0.0f;
    for (int t = 0; t < len; ++t) {
      const float* row = logits + ((size_t)t * B + b) * V;
      int best = 0;
      float bv = row[0];
      for (int v = 1; v < V; ++v) {
        if (bv < row[v]) { /* strict: first maximum wins */
          bv = row[v];
          best = v;
        }
      }
      score += bv;
      if (best != blank && !(merge_repeated && best == prev)) {
        out_ids[(size_t)b * T + n++] = best;
      }
      prev = best;
    }
    for (int t = n; t < T; ++t) out_ids[(size_t)b * T + t] = -1;
    out_len[b] = n;
    if (neg_sum_logits) neg_sum_logits[b] = -score;
  }
  return 0;
}
