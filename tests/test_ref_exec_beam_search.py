"""The Transformer beam-search oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_beam_search.npz = open_seq2seq/parts/transformer/beam_search.py:sequence_beam_search — the
tf.while_loop over _continue_search / _search_step (alive and finished sets, 2 x beam candidates, length
normalisation ((5 + len) / 6)^alpha, INF = 32768, the "no finished sequence" fallback) — executed from the reference's
file by tests/golden/make_ref_exec.py on a table-driven symbols_to_logits_fn (logits = table[step][last id] + a cache
entry that rides through the search's expand / flatten / gather plumbing). tf.while_loop is TensorFlow library code:
the stand-in calls cond and body once and replays the recorded graphs, reporting loop-variable dimensions that the
shape invariants leave open as unknown (oracle/ref_shim/tf1). oracle/beam_search.py — what the device's beam-search
kernels are tested against bit-exactly — must return the same ids (exact) and scores (1e-5) in three regimes:
beams finishing at different lengths, no beam ever finishing, early termination."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402
from oracle import beam_search as obs  # noqa: E402


@pytest.mark.parametrize("case", sorted(rx.gen.BEAM_CASES))
def test_oracle_reproduces_the_reference_beam_search(case):
  d = np.load(os.path.join(HERE, "golden", "ref_exec_beam_search.npz"))
  cfg = rx.gen.BEAM_CASES[case]
  table, bias = rx.gen.beam_tables(cfg)

  def fn(ids, i, cache):
    return table[i][ids[:, -1]] + cache["bias"], cache
  ids, scores = obs.sequence_beam_search(fn, np.zeros(cfg["B"], np.int32), {"bias": bias}, cfg["V"], cfg["beam"],
                                         cfg["alpha"], cfg["L"], cfg["eos"])
  ref_ids, ref_scores = d[case + "/ids"], d[case + "/scores"]
  assert ids.shape == ref_ids.shape, (ids.shape, ref_ids.shape)
  assert np.array_equal(ids, ref_ids), (ids, ref_ids)
  assert np.allclose(scores, ref_scores, rtol=1e-5, atol=1e-5), (scores, ref_scores)
  L1 = ref_ids.shape[2]
  if case == "finishes":
    ends = sorted({int(np.argmax(r == cfg["eos"])) for r in ref_ids.reshape(-1, L1) if (r == cfg["eos"]).any()})
    assert len(ends) >= 2, "beams finish at different lengths"
  if case == "never_finishes":
    assert not (ref_ids == cfg["eos"]).any() and L1 == cfg["L"] + 1, "alive sequences are returned"
  if case == "early_stop":
    assert L1 < cfg["L"] + 1, "the loop stops before max_decode_length"


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "beam_search"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and "reproduced" in r.stdout, r.stdout + r.stderr
