"""GPU parity: os2s_ctc_greedy_decode (HIP, through the C ABI) vs the oracle.
Integer outputs => bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(logits, lens, **kw):
  import torch
  from openseq2seq_amd import capi
  dev = torch.device("cuda:0")
  ids, n, neg = capi.ctc_greedy_decode(torch.from_numpy(logits).to(dev),
                                       torch.from_numpy(lens).to(dev), **kw)
  torch.cuda.synchronize()
  return ids.cpu().numpy(), n.cpu().numpy(), neg.cpu().numpy()


def test_golden_then_seconds(cuda):
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))
  meta = json.load(open(os.path.join(GOLD, "ctc_test_meta.json")))
  ids, n, neg = _run(seq, np.array([seq.shape[0]], np.int32))
  text = "".join(meta["vocab"][c] for c in ids[0, :n[0]])
  assert text == "then seconds"
  assert abs(float(neg[0]) - meta["greedy_neg_sum_logits"]) < meta["tol"]


@pytest.mark.parametrize("T,B,V", [(1, 1, 2), (64, 7, 29), (257, 3, 29), (840, 32, 29),
                                    (300, 5, 32), (100, 4, 1000), (17, 2, 40000)])
@pytest.mark.parametrize("merge", [True, False])
def test_vs_oracle(cuda, T, B, V, merge):
  from oracle import ctc_greedy as og
  rng = np.random.RandomState(T * 131 + B * 7 + V)
  lg = (rng.randn(T, B, V) * 2).astype(np.float32)
  lg[:, :, V - 1] += 1.5
  if T > 20:
    lg[5:15, 0, :] = 0.0  # ties -> lowest index
    lg[:, B - 1, :] = lg[0:1, B - 1, :]  # constant symbol
  lens = rng.randint(0, T + 1, size=B).astype(np.int32)
  lens[0] = T
  ids, n, neg = _run(lg, lens, merge_repeated=merge)
  rid, rn, rneg = og.greedy_c(lg, lens, merge_repeated=merge)
  assert np.array_equal(n, rn)
  assert np.array_equal(ids, rid)
  assert np.allclose(neg, rneg, rtol=1e-5, atol=1e-2)


def test_full_size_properties(cuda):
  """BASELINE size (T'=840, B=32, V=29): idempotence-style properties that do
  not need the oracle: outputs contain no blank, no adjacent repeats arise only
  from blank separation, lengths <= seq_len."""
  rng = np.random.RandomState(0)
  T, B, V = 840, 32, 29
  lg = rng.randn(T, B, V).astype(np.float32)
  lens = rng.randint(1, T + 1, size=B).astype(np.int32)
  ids, n, _ = _run(lg, lens)
  am = lg.argmax(-1)  # [T,B]
  for b in range(B):
    out = ids[b, :n[b]]
    assert (out != V - 1).all() and (out >= 0).all()
    assert n[b] <= lens[b]
    assert (ids[b, n[b]:] == -1).all()
    # re-collapse the framewise argmax: must reproduce
    a = am[:lens[b], b]
    keep = a != V - 1
    keep[1:] &= a[1:] != a[:-1]
    assert np.array_equal(out, a[keep])
