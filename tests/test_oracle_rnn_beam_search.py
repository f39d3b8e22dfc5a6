"""The RNN beam-search oracle (oracle/rnn_beam_search.py, restating TF's BeamSearchDecoder)
against exhaustive enumeration, greedy equivalence at beam width 1, and gather_tree behaviour."""
import itertools

import numpy as np

from oracle import rnn_beam_search as orb


def _table_fn(table, V):
  def fn(ids, time, parents):
    return table[min(time, table.shape[0] - 1)][ids % table.shape[1]]
  return fn


def test_beam1_is_greedy():
  rng = np.random.RandomState(0)
  V, T = 7, 6
  table = rng.randn(T, V, V).astype(np.float32) * 2
  pred, lengths, _ = orb.beam_search(_table_fn(table, V), 3, 1, V, 2, 1, 0.0, T)
  for b in range(3):
    cur, seq = 2, []
    for t in range(T):
      cur = int(np.argmax(table[t][cur]))
      seq.append(cur)
      if cur == 1:
        break
    got = pred[b, :len(seq), 0].tolist()
    assert got == seq, (got, seq)


def test_wide_beam_is_exhaustive():
  """beam width = vocabulary size and two steps: every two-token path survives the first step,
  so the best final hypothesis is the exhaustive optimum (no length penalty)."""
  rng = np.random.RandomState(1)
  V, T, end = 8, 2, 1
  table = rng.randn(T, V, V).astype(np.float32) * 1.5
  pred, lengths, scores = orb.beam_search(_table_fn(table, V), 1, V, V, 0, end, 0.0, T)
  logp = table - np.log(np.exp(table).sum(-1, keepdims=True))
  best = (-np.inf, None)
  for a in range(V):
    if a == end:                     # finished after one token: continues with END at no cost
      cand = (logp[0][0][a], [a, end])
    else:
      b2 = int(np.argmax(logp[1][a]))
      cand = (logp[0][0][a] + logp[1][a][b2], [a, b2])
    if cand[0] > best[0]:
      best = cand
  assert abs(scores[0, 0] - best[0]) < 1e-4
  assert pred[0, :, 0].tolist() == best[1]
  assert np.all(np.diff(scores[0]) <= 1e-6)           # beams come out sorted by score


def test_length_penalty_and_finished_beams():
  V, end = 6, 1
  rng = np.random.RandomState(2)
  table = rng.randn(5, V, V).astype(np.float32)
  table[:, :, end] += 1.0
  for lpw in (0.0, 0.6, 1.0):
    pred, lengths, scores = orb.beam_search(_table_fn(table, V), 2, 3, V, 0, end, lpw, 5)
    assert np.all(np.diff(scores, axis=1) <= 1e-6)
    for b in range(2):
      for w in range(3):
        seq = pred[b, :, w].tolist()
        if end in seq:                 # after the first END only END; stored length counts it
          k = seq.index(end)
          assert all(t == end for t in seq[k:]) and lengths[b, w] == k + 1
        else:
          assert lengths[b, w] == len(seq)
  # a finished hypothesis keeps its log-probability while it waits in the beam
  lp = np.array([[-1.0, -2.0]], np.float32)
  fin = np.array([[True, False]])
  ln = np.array([[2, 2]], np.int64)
  logits = np.zeros((1, 2, V), np.float32)
  sc, word, parent, nlp, nfin, nlen = orb.beam_step(logits, lp, fin, ln, 3, end, 0.0)
  assert word[0, 0] == end and parent[0, 0] == 0 and nlp[0, 0] == -1.0 and nlen[0, 0] == 2 and nfin[0, 0]
  assert parent[0, 1] == 1 and abs(nlp[0, 1] - (-2.0 - np.log(V))) < 1e-5 and nlen[0, 1] == 3


def test_gather_tree():
  ids = np.array([[[2, 3]], [[4, 1]], [[5, 6]]], np.int32)        # [T=3, B=1, W=2]
  par = np.array([[[0, 0]], [[1, 0]], [[0, 1]]], np.int32)
  out = orb.gather_tree(ids, par, np.array([3]), 1)
  # beam 0 at t=2: id 5, parent 0 -> t=1 id 4 (parent 1) -> t=0 id 3
  assert out[:, 0, 0].tolist() == [3, 4, 5]
  # beam 1 at t=2: id 6, parent 1 -> t=1 id 1 (END) -> after END everything is END
  assert out[:, 0, 1].tolist() == [2, 1, 1]
