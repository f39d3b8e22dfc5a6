"""The beam-search oracle (oracle/beam_search.py) against the reference's own known answers
(open_seq2seq/parts/transformer/beam_search_test.py:28-100) and against brute-force
enumeration of every sequence on a tiny vocabulary (with a beam wide enough to hold all
alive prefixes, beam search is exhaustive search)."""
import itertools

import numpy as np

from oracle import beam_search as obs


def test_expand_flatten_unflatten_shapes():
  assert obs.expand_to_beam_size(np.ones([7, 4, 2, 5]), 3).shape == (7, 3, 4, 2, 5)
  assert obs.flatten_beam_dim(np.ones([7, 4, 2, 5])).shape == (28, 2, 5)
  assert obs.unflatten_beam_dim(np.ones([28, 2, 5]), 7, 4).shape == (7, 4, 2, 5)


def test_gather_beams_kat():
  x = np.arange(24).reshape(2, 3, 4)
  y = obs.gather_beams(x, [[1, 2], [0, 2]], 2, 2)
  assert y.tolist() == [[[4, 5, 6, 7], [8, 9, 10, 11]], [[12, 13, 14, 15], [20, 21, 22, 23]]]


def test_gather_topk_beams_kat():
  x = np.arange(24).reshape(2, 3, 4)
  y = obs.gather_topk_beams(x, [[0, 1, 1], [1, 0, 1]], 2, 2)      # ties: lower index first
  assert y.tolist() == [[[4, 5, 6, 7], [8, 9, 10, 11]], [[12, 13, 14, 15], [20, 21, 22, 23]]]


def test_top_k_ties_prefer_lower_index():
  v, i = obs.top_k(np.array([[1., 3., 3., 2., 3.]], np.float32), 3)
  assert i.tolist() == [[1, 2, 4]] and v.tolist() == [[3., 3., 3.]]


def _markov_fn(table):
  """logits depend on (previous token, step)."""
  def fn(ids, i, cache):
    return table[i][ids[:, -1]], cache
  return fn


def test_exhaustive_equivalence():
  rng = np.random.RandomState(3)
  V, T, eos = 4, 4, 1
  alpha = 0.6
  table = rng.randn(T, V, V).astype(np.float32) * 2.0
  beam = V ** T       # never prunes
  seqs, scores = obs.sequence_beam_search(_markov_fn(table), np.zeros(1, np.int32), {}, V, beam,
                                          alpha, T, eos)
  # brute force: every sequence that ends with its first EOS within T steps
  logp = table - np.log(np.exp(table).sum(-1, keepdims=True))
  best = (-np.inf, None)
  for L in range(1, T + 1):
    for body in itertools.product([t for t in range(V) if t != eos], repeat=L - 1):
      s = list(body) + [eos]
      lp, prev = 0.0, 0
      for i, t in enumerate(s):
        lp += logp[i][prev][t]
        prev = t
      sc = lp / ((5.0 + L) / 6.0) ** alpha
      if sc > best[0]:
        best = (sc, s)
  top = seqs[0, 0, 1:].tolist()
  assert top[:len(best[1])] == best[1] and all(t == 0 for t in top[len(best[1]):])
  assert abs(scores[0, 0] - best[0]) < 1e-5


def test_no_finished_returns_alive():
  """:85-94: a batch item without any finished sequence returns its alive beams."""
  V, T = 6, 3
  table = np.zeros((T, V, V), np.float32)
  table[:, :, 1] = -1e4          # EOS practically impossible
  table[:, :, 3] = 2.0
  seqs, scores = obs.sequence_beam_search(_markov_fn(table), np.zeros(2, np.int32), {}, V, 2, 0.6, T, 1)
  assert seqs.shape == (2, 2, T + 1) and seqs[0, 0].tolist() == [0, 3, 3, 3]
  assert np.all(scores[:, 0] > -10)
