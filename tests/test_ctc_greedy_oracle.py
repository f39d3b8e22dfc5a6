"""CPU tests: pin the CTC greedy oracle against the reference's golden vector
(ctc_decoder_with_lm/ctc-test.py:60-67) and against the reference's own C++
(decoders/ctc_greedy_decoder.cpp compiled into oracle/_ref/)."""
import json
import os

import numpy as np
import pytest

from oracle import ctc_greedy as og

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _golden():
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))
  meta = json.load(open(os.path.join(GOLD, "ctc_test_meta.json")))
  return seq, meta


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_golden_then_seconds(impl):
  seq, meta = _golden()
  f = og.greedy_numpy if impl == "numpy" else og.greedy_c
  ids, lens, neg = f(seq, np.array([seq.shape[0]], np.int32))
  text = "".join(meta["vocab"][c] for c in ids[0, :lens[0]])
  assert text == meta["greedy_text"] == "then seconds"
  assert abs(float(neg[0]) - meta["greedy_neg_sum_logits"]) < meta["tol"]


def _softmax(x):
  e = np.exp(x - x.max(-1, keepdims=True))
  return e / e.sum(-1, keepdims=True)


@pytest.mark.skipif(not og.reference_cpp_available(),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_restatement_matches_reference_cpp():
  seq, meta = _golden()
  probs = _softmax(seq[:, 0, :].astype(np.float64))
  ref_ids = og.greedy_reference_cpp(probs)
  ids, lens, _ = og.greedy_c(seq, np.array([seq.shape[0]], np.int32))
  assert np.array_equal(ref_ids, ids[0, :lens[0]])
  rng = np.random.RandomState(1)
  for trial in range(20):
    T, V = int(rng.randint(1, 200)), int(rng.randint(2, 40))
    # peaky logits so repeats / blanks are frequent
    lg = (rng.randn(T, 1, V) * 3).astype(np.float32)
    lg[:, 0, V - 1] += 2.0
    lg = np.repeat(lg, rng.randint(1, 4), axis=0)[:T]
    r = og.greedy_reference_cpp(_softmax(lg[:, 0, :].astype(np.float64)))
    for f in (og.greedy_numpy, og.greedy_c):
      ids, lens, _ = f(lg, np.array([T], np.int32))
      assert np.array_equal(r, ids[0, :lens[0]]), (trial, f.__name__)


def test_numpy_vs_c_edge_cases():
  rng = np.random.RandomState(2)
  T, B, V = 64, 7, 29
  lg = rng.randn(T, B, V).astype(np.float32)
  lg[5:15, 2, :] = 0.0  # exact ties: first maximum (id 0) must win
  lg[:, 3, :] = lg[0:1, 3, :]  # one symbol repeated over all frames
  lens = np.array([64, 0, 64, 64, 1, 33, 70], np.int32)  # empty / ragged / over-long
  for merge in (True, False):
    a = og.greedy_numpy(lg, lens, merge_repeated=merge)
    b = og.greedy_c(lg, lens, merge_repeated=merge)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.allclose(a[2], b[2], rtol=1e-6, atol=1e-4)
  ids, n, _ = og.greedy_numpy(lg, lens)
  assert n[1] == 0 and (ids[1] == -1).all()
  assert n[3] <= 1
