"""ORACLE (test infrastructure only): NumPy restatement of the Transformer beam search,
open_seq2seq/parts/transformer/beam_search.py (line numbers below refer to that file), with
the same gather-based formulation the reference builds as a tf.while_loop:

  sequence_beam_search :386-418, SequenceBeamSearch.search :71-94,
  _create_initial_state :96-161, _continue_search :163-203, _search_step :205-236,
  _grow_alive_seq :238-296, _get_new_alive_state :298-327, _get_new_finished_state :329-383,
  helpers :421-541 (_log_prob_from_logits, _length_normalization, _expand_to_beam_size,
  _flatten/_unflatten_beam_dim, _gather_beams, _gather_topk_beams).

PARITY STATUS (round 5): the whole search is pinned to the reference's OWN CODE — sequence_beam_search executed
from its file on the TF-primitive stand-in oracle/ref_shim/tf1 (tf.while_loop restated as a traced loop): ids exact,
scores 1e-5 in three regimes (tests/test_ref_exec_beam_search.py); driving oracle/transformer.py it returns the ids of
the reference's TransformerDecoder.predict exactly, at d 32 and at d 512 (tests/test_ref_exec_transformer.py). Also
pinned to the reference's own known answers (beam_search_test.py: expand / flatten /
unflatten shapes, _gather_beams and _gather_topk_beams values) in
tests/test_oracle_beam_search.py. tf.nn.top_k semantics: descending values, the LOWER index
wins between equal values (a stable argsort of the negated values restates that). All score
arithmetic is float32 like the reference's graph; the length-normalisation factor
pow((5 + len) / 6, alpha) is evaluated in float32.
"""
import numpy as np

INF = np.float32(32768.0)   # beam_search.py:26


def log_prob_from_logits(logits):
  """:421-422 (float32, max-shifted logsumexp like tf.reduce_logsumexp)."""
  logits = logits.astype(np.float32)
  m = np.max(logits, axis=2, keepdims=True)
  m = np.where(np.isfinite(m), m, np.float32(0))
  lse = np.log(np.sum(np.exp(logits - m), axis=2, keepdims=True, dtype=np.float32)).astype(np.float32) + m
  return logits - lse


def length_normalization(alpha, length):
  """:425-427"""
  return np.power(np.float32((5.0 + np.float32(length)) / 6.0), np.float32(alpha)).astype(np.float32)


def expand_to_beam_size(tensor, beam_size):
  """:430-444"""
  tensor = np.expand_dims(tensor, axis=1)
  tile_dims = [1] * tensor.ndim
  tile_dims[1] = beam_size
  return np.tile(tensor, tile_dims)


def flatten_beam_dim(tensor):
  """:474-486"""
  shape = list(tensor.shape)
  shape[0] *= shape[1]
  shape.pop(1)
  return tensor.reshape(shape)


def unflatten_beam_dim(tensor, batch_size, beam_size):
  """:489-502"""
  return tensor.reshape([batch_size, beam_size] + list(tensor.shape[1:]))


def _map(fn, nested):
  if isinstance(nested, dict):
    return {k: _map(fn, v) for k, v in nested.items()}
  if isinstance(nested, (list, tuple)):
    return type(nested)(_map(fn, v) for v in nested)
  return fn(nested)


def top_k(values, k):
  """tf.nn.top_k over the last axis: (values, indices), lower index first among equals."""
  values = np.asarray(values)
  idx = np.argsort(-values, axis=-1, kind="stable")[..., :k]
  return np.take_along_axis(values, idx, axis=-1), idx.astype(np.int32)


def gather_beams(nested, beam_indices, batch_size, new_beam_size):
  """:505-537"""
  beam_indices = np.asarray(beam_indices)
  batch_pos = (np.arange(batch_size * new_beam_size) // new_beam_size).reshape(batch_size, new_beam_size)
  return _map(lambda state: np.asarray(state)[batch_pos, beam_indices], nested)


def gather_topk_beams(nested, score_or_log_prob, batch_size, beam_size):
  """:540-541"""
  _, topk_indexes = top_k(np.asarray(score_or_log_prob), beam_size)
  return gather_beams(nested, topk_indexes, batch_size, beam_size)


def sequence_beam_search(symbols_to_logits_fn, initial_ids, initial_cache, vocab_size, beam_size,
                         alpha, max_decode_length, eos_id, return_steps=False):
  """symbols_to_logits_fn(ids [B*beam, i+1], i, cache) -> (logits [B*beam, V] float, cache).
  Returns (finished_seq [B, beam, steps+1] int32, finished_scores [B, beam] float32)."""
  initial_ids = np.asarray(initial_ids, np.int32)
  B = initial_ids.shape[0]
  # ---- _create_initial_state ------------------------------------------------------------
  i = 0
  alive_seq = expand_to_beam_size(initial_ids, beam_size)[:, :, None]
  alive_log_probs = np.tile(np.array([[0.0] + [-np.inf] * (beam_size - 1)], np.float32), [B, 1])
  alive_cache = _map(lambda t: expand_to_beam_size(np.asarray(t), beam_size), initial_cache)
  finished_seq = np.zeros(alive_seq.shape, np.int32)
  finished_scores = np.ones([B, beam_size], np.float32) * -INF
  finished_flags = np.zeros([B, beam_size], bool)

  def continue_search():
    if not i < max_decode_length:
      return False
    max_length_norm = length_normalization(alpha, max_decode_length)
    best_alive_scores = alive_log_probs[:, 0] / max_length_norm
    fs = finished_scores * finished_flags.astype(np.float32)
    lowest = np.min(fs, axis=1)
    lowest = lowest + (np.float32(1.) - np.any(finished_flags, 1).astype(np.float32)) * -INF
    return not bool(np.all(lowest > best_alive_scores))

  while continue_search():
    # ---- _grow_alive_seq ----------------------------------------------------------------
    beams_to_keep = 2 * beam_size
    flat_ids = flatten_beam_dim(alive_seq)
    flat_cache = _map(flatten_beam_dim, alive_cache)
    flat_logits, flat_cache = symbols_to_logits_fn(flat_ids, i, flat_cache)
    logits = unflatten_beam_dim(np.asarray(flat_logits, np.float32), B, beam_size)
    new_cache = _map(lambda t: unflatten_beam_dim(np.asarray(t), B, beam_size), flat_cache)
    candidate_log_probs = log_prob_from_logits(logits)
    log_probs = candidate_log_probs + alive_log_probs[:, :, None]
    flat_log_probs = log_probs.reshape(-1, beam_size * vocab_size)
    topk_log_probs, topk_indices = top_k(flat_log_probs, beams_to_keep)
    topk_beam_indices = topk_indices // vocab_size
    topk_seq, new_cache = gather_beams([alive_seq, new_cache], topk_beam_indices, B, beams_to_keep)
    topk_ids = (topk_indices % vocab_size)[:, :, None]
    new_seq = np.concatenate([topk_seq, topk_ids.astype(np.int32)], axis=2)
    new_log_probs = topk_log_probs.astype(np.float32)
    # ---- _get_new_alive_state -----------------------------------------------------------
    new_finished_flags = new_seq[:, :, -1] == eos_id
    masked = new_log_probs + new_finished_flags.astype(np.float32) * -INF
    top_alive_seq, top_alive_log_probs, top_alive_cache = gather_topk_beams(
        [new_seq, masked, new_cache], masked, B, beam_size)
    # ---- _get_new_finished_state --------------------------------------------------------
    fseq = np.concatenate([finished_seq, np.zeros([B, beam_size, 1], np.int32)], axis=2)
    length_norm = length_normalization(alpha, i + 1)
    new_scores = new_log_probs / length_norm
    new_scores = new_scores + (np.float32(1.) - new_finished_flags.astype(np.float32)) * -INF
    fseq = np.concatenate([fseq, new_seq], axis=1)
    fscores = np.concatenate([finished_scores, new_scores.astype(np.float32)], axis=1)
    fflags = np.concatenate([finished_flags, new_finished_flags], axis=1)
    finished_seq, finished_scores, finished_flags = gather_topk_beams(
        [fseq, fscores, fflags], fscores, B, beam_size)
    alive_seq, alive_log_probs, alive_cache = top_alive_seq, top_alive_log_probs, top_alive_cache
    i += 1

  # ---- search() epilogue :85-94 -----------------------------------------------------------
  any_fin = np.any(finished_flags, 1)
  out_seq = np.where(any_fin[:, None, None], finished_seq, alive_seq)
  out_scores = np.where(any_fin[:, None], finished_scores, alive_log_probs)
  if return_steps:
    return out_seq.astype(np.int32), out_scores.astype(np.float32), i
  return out_seq.astype(np.int32), out_scores.astype(np.float32)
