"""Learning-rate policies and loss scalers against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_optim.npz = open_seq2seq/optimizers/lr_policies.py (all seven policies, eleven parameter sets)
evaluated at 39 global steps, and automatic_loss_scaler.py's BackoffScaler / LogMaxScaler driven through 260
(has_nan, amax) events with their update_op built once and run per event — executed from the reference's files by
tests/golden/make_ref_exec.py. Held against: the oracle restatements (oracle/optim.py) and the host-side policy
functions of the product (openseq2seq_amd/optimizers/lr_policies.py — what Model.compile evaluates when a policy has
no device id, and what the device ids are tested against on the GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402
from oracle import optim as oo  # noqa: E402


def _cases():
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_optim.npz")))
  return d, rx.gen.LR_CASES, [int(s) for s in d["lr_steps"]]


def test_lr_policies_match_the_reference_code():
  from openseq2seq_amd.optimizers import lr_policies as product
  d, cases, steps = _cases()
  assert len(cases) == 11 and {c[0] for c in cases} == {"fixed_lr", "piecewise_constant", "exp_decay", "poly_decay",
                                                        "cosine_decay", "transformer_policy", "inv_poly_decay"}
  for i, (name, params) in enumerate(cases):
    ref = d["lr/%d/%s" % (i, name)]
    for fn, who in ((getattr(oo, name), "oracle"), (getattr(product, name), "product")):
      got = np.array([float(fn(st, **params)) for st in steps])
      # the reference computes in float32 (tf.cast(global_step, tf.float32), tf.pow): 2e-6 relative
      assert np.allclose(got, ref, rtol=2e-6, atol=1e-12), (who, name, params, np.abs(got / ref - 1).max())


def test_loss_scalers_match_the_reference_code():
  d, _, _ = _cases()
  nan, amax = d["ev_nan"], d["ev_amax"]
  assert nan.sum() >= 7 and np.isinf(amax).sum() >= 4
  made = {
      0: oo.BackoffScaler(step_window=16),
      1: oo.BackoffScaler(scale_min=8.0, scale_max=4096.0, step_factor=4.0, step_window=5),
      2: oo.LogMaxScaler(),
      3: oo.LogMaxScaler(scale_max=2.0 ** 10, beta1=0.9, beta2=0.95, overflow_std_dev=2.0),
  }
  for i, sc in made.items():
    key = [k for k in d if k.startswith("scale/%d/" % i)][0]
    ref = d[key]
    trace = [float(sc.scale)]
    for h, a in zip(nan, amax):
      sc.update(bool(h), np.float32(a))
      trace.append(float(sc.scale))
    trace = np.array(trace, np.float32)
    if "backoff" in key:
      assert np.array_equal(trace, ref), (key, np.nonzero(trace != ref)[0][:5])      # powers of the step factor: exact
      assert len(set(ref.tolist())) >= 4, "the trace shrinks AND grows"
    else:
      assert np.allclose(trace, ref, rtol=2e-4), (key, np.abs(trace / ref - 1).max())
      assert ref.min() < ref.max()


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "optim"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and "reproduced" in r.stdout, r.stdout + r.stderr
