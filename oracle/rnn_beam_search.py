"""ORACLE (test infrastructure only): NumPy restatement of tf.contrib.seq2seq.BeamSearchDecoder as
driven by BeamSearchRNNDecoderWithAttention (open_seq2seq/decoders/rnn_decoders.py:324-532:
BeamSearchDecoder(cell, embedding, start_tokens=GO, end_token=END, beam_width,
output_layer, length_penalty_weight) under dynamic_decode(maximum_iterations = 2 * max source
length), outputs = final_outputs.predicted_ids[:, :, 0]).

BeamSearchDecoder lives in TensorFlow (tensorflow/contrib/seq2seq/python/ops/beam_search_decoder.py,
TF 1.x), which is neither vendored in /root/reference nor installable here: its published algorithm
(_beam_search_step, _mask_probs, _get_scores / _length_penalty, gather_tree in finalize) is
restated and cross-checked against exhaustive search (tests/test_oracle_rnn_beam_search.py).
PARITY STATUS (round 5): the reference carries its OWN copy of that class
(parts/rnns/rnn_beam_search_decoder.py — beams 1.. start "finished" with log-probability -inf instead of
the time-0 special case below; the same search) and THAT file is executed, under the reference's
BeamSearchRNNDecoderWithAttention, on the TF-primitive stand-in oracle/ref_shim/tf1: this restatement
returns the same winner ids, and every beam's length, finished flag and log-probability (1e-5), on three
cases (tests/test_ref_exec_nmt_beam.py). tile_batch, gather_tree, top_k and dynamic_decode are
TensorFlow library code, restated in the stand-in.

One step, state = (log_probs [B,W], finished [B,W], lengths [B,W]):
  step_log_probs = log_softmax(logits); finished beams put all mass on END (_mask_probs: END -> 0,
  every other token -> float32 min)
  total = log_probs[..., None] + step_log_probs
  candidate length = lengths + (not finished) * (token != END)
  scores = total / ((5 + length) / 6) ** length_penalty_weight
  time 0: only beam 0 competes (all beams start identical); top W of the W*V flat scores
  (tf.nn.top_k: descending, lower index first among equals)
  next log_probs = total at the winners; next finished = parent finished | token == END;
  next lengths = parent lengths + (not parent finished)
"""
import numpy as np


def log_softmax(x):
  x = x.astype(np.float32)
  m = x.max(-1, keepdims=True)
  return (x - m) - np.log(np.exp(x - m).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def length_penalty(lengths, weight):
  if weight == 0.0:
    return np.ones_like(lengths, dtype=np.float32)
  return np.power((np.float32(5.0) + lengths.astype(np.float32)) / np.float32(6.0), np.float32(weight)).astype(np.float32)


def top_k(values, k):
  idx = np.argsort(-values, axis=-1, kind="stable")[..., :k]
  return np.take_along_axis(values, idx, axis=-1), idx


def beam_step(logits, log_probs, finished, lengths, time, end_token, length_penalty_weight):
  """logits [B, W, V] -> (scores, word_ids, parent_beam, new_log_probs, new_finished, new_lengths)."""
  B, W, V = logits.shape
  step = log_softmax(logits)
  fin = finished.astype(bool)
  masked = np.full((V,), np.finfo(np.float32).min, np.float32)
  masked[end_token] = 0.0
  step = np.where(fin[..., None], masked[None, None, :], step)
  total = log_probs[..., None].astype(np.float32) + step
  not_end = np.ones(V, np.int64)
  not_end[end_token] = 0
  cand_len = lengths[..., None] + (~fin)[..., None].astype(np.int64) * not_end[None, None, :]
  scores = total / length_penalty(cand_len, length_penalty_weight)
  flat = scores.reshape(B, W * V) if time > 0 else scores[:, 0]
  best, idx = top_k(flat, W)
  word = (idx % V).astype(np.int32)
  parent = (idx // V).astype(np.int32)
  rows = np.arange(B)[:, None]
  new_lp = total.reshape(B, W * V)[rows, idx] if time > 0 else total[:, 0][rows, idx]
  par_fin = fin[rows, parent]
  new_fin = par_fin | (word == end_token)
  new_len = lengths[rows, parent] + (~par_fin).astype(lengths.dtype)
  return best.astype(np.float32), word, parent, new_lp.astype(np.float32), new_fin, new_len


def gather_tree(step_ids, parent_ids, max_lengths, end_token):
  """tf.contrib.seq2seq.gather_tree: [T, B, W] ids / parents -> full sequences per final beam; after
  the first END of a sequence every later position is END."""
  T, B, W = step_ids.shape
  out = np.full((T, B, W), end_token, np.int32)
  for b in range(B):
    for w in range(W):
      L = min(int(max_lengths[b]), T)
      if L <= 0:
        continue
      parent = w
      for t in range(L - 1, -1, -1):
        out[t, b, w] = step_ids[t, b, parent]
        parent = parent_ids[t, b, parent]
      seen = False
      for t in range(L):
        if seen:
          out[t, b, w] = end_token
        elif out[t, b, w] == end_token:
          seen = True
  return out


def beam_search(logits_fn, B, beam_width, vocab_size, start_token, end_token, length_penalty_weight,
                maximum_iterations, return_log_probs=False):
  """logits_fn(ids [B*W] int, time, parent_rows [B*W] | None) -> logits [B*W, V]. Returns
  (predicted_ids [B, T, W], lengths [B, W], scores [B, W])."""
  W = beam_width
  log_probs = np.tile(np.array([[0.0] + [-np.inf] * (W - 1)], np.float32), [B, 1])
  finished = np.zeros((B, W), bool)
  lengths = np.zeros((B, W), np.int64)
  ids = np.full((B * W,), start_token, np.int32)
  parents_flat = None
  hist_ids, hist_par = [], []
  scores = np.zeros((B, W), np.float32)
  for time in range(maximum_iterations):
    logits = np.asarray(logits_fn(ids, time, parents_flat), np.float32).reshape(B, W, vocab_size)
    scores, word, parent, log_probs, finished, lengths = beam_step(
        logits, log_probs, finished, lengths, time, end_token, length_penalty_weight)
    hist_ids.append(word)
    hist_par.append(parent)
    ids = word.reshape(-1)
    parents_flat = (np.arange(B)[:, None] * W + parent).reshape(-1)
    if finished.all():
      break
  T = len(hist_ids)
  step_ids = np.stack(hist_ids, 0)
  par = np.stack(hist_par, 0)
  max_len = lengths.max(1)
  pred = gather_tree(step_ids, par, max_len, end_token)
  if return_log_probs:        # + final_state.log_probs / .finished of every beam
    return np.transpose(pred, (1, 0, 2)), lengths, scores, log_probs, finished
  return np.transpose(pred, (1, 0, 2)), lengths, scores
