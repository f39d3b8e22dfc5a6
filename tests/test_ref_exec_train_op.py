"""The train op against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_train_op.npz = optimizers.optimize_loss(dtype="mixed") of the reference — the
MixedPrecisionOptimizerWrapper, post_process_gradients (LARC / global-norm clipping), the automatic loss scalers and
NovoGrad (optimizers/novograd.py) / Adam / Momentum — built once and run 28 steps on a toy model (two fp16 matrices, an
fp16 bias, an fp32 vector) whose half-precision gradients overflow at the initial loss scale, executed from the
reference's files by tests/golden/make_ref_exec.py. Three cases: NovoGrad + LARC + Backoff (the Jasper configuration),
Adam + Backoff + l2 (the Transformer's optimizer family), Momentum + global-norm clipping + LogMax.

oracle/optim.py:RefOptimizer — the restatement every device optimizer test is held against — is fed the recorded
scaled gradients and must reproduce, step by step: the loss scale going into the step, which steps were skipped
(global_step), the FP32 master copies and the fp16 / fp32 variables after the step. That includes NovoGrad AS WRITTEN
in the reference (the second-moment EMA variable is never assigned: novograd.py:107-113 — executed here, not argued)."""
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


def _scaler(cfg):
  p = cfg["loss_scaling_params"]
  return oo.BackoffScaler(**p) if cfg["loss_scaling"] == "Backoff" else oo.LogMaxScaler(**p)


@pytest.mark.parametrize("case", sorted(rx.gen.TRAIN_OP_CASES))
def test_ref_optimizer_reproduces_the_reference_train_op(case):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_train_op.npz")))
  cfg = rx.gen.TRAIN_OP_CASES[case]
  consts = rx.gen.train_op_constants()
  names = [v[0] for v in rx.gen.TRAIN_OP_VARS]
  half = {v[0]: v[2] == "float16" for v in rx.gen.TRAIN_OP_VARS}
  lr_name, lr_params = cfg["lr"]
  opt = oo.RefOptimizer([consts[n][0].astype(np.float32) for n in names], optimizer=cfg["optimizer"],
                        opt_params=cfg["optimizer_params"], lr_fn=lambda s: getattr(oo, lr_name)(s, **lr_params),
                        larc_params=cfg.get("larc_params"), clip_gradients=cfg.get("clip_gradients"),
                        scaler=_scaler(cfg),
                        l2=[cfg["l2"] if (half[n] and consts[n][0].ndim == 2) else 0.0 for n in names])
  steps = int(d["steps"])
  skipped = 0
  for st in range(steps):
    assert float(opt.loss_scale) == float(d[case + "/scale_in"][st]), (st, opt.loss_scale, d[case + "/scale_in"][st])
    assert abs(float(opt.lr_fn(opt.global_step)) / float(d[case + "/lr"][st]) - 1) < 2e-6, st
    skip = opt.step([d["%s/g/%s" % (case, n)][st] for n in names])
    skipped += bool(skip)
    assert opt.global_step == int(d[case + "/global_step_after"][st]), (st, opt.global_step)
    for i, n in enumerate(names):
      ref_m = d["%s/m/%s" % (case, n)][st]
      assert np.allclose(opt.w[i], ref_m, rtol=1e-4, atol=2e-5), (st, n, np.abs(opt.w[i] - ref_m).max())
      if half[n]:       # the fp16 variable = saturate_cast(master copy) (mp_wrapper.py:104-109)
        as16 = np.clip(opt.w[i], -65504, 65504).astype(np.float16).astype(np.float32)
        assert np.abs(as16 - d["%s/w/%s" % (case, n)][st]).max() <= 2e-3 * np.abs(as16).max(), (st, n)
  assert float(opt.loss_scale) == float(d[case + "/scale_final"])
  # the trace exercises what it is meant to: skipped steps at the start (fp16 overflow), applied steps after
  assert int(d[case + "/global_step_after"][-1]) == steps - skipped
  if cfg["loss_scaling"] == "Backoff":
    assert skipped >= 4 and steps - skipped >= 18, skipped
    sc = d[case + "/scale_in"]
    assert (np.diff(sc) > 0).any() and (np.diff(sc) < 0).any(), "the scale shrinks on overflow and grows after a window"
  print("%s: %d of %d steps skipped, final loss scale %g" % (case, skipped, steps, float(opt.loss_scale)))


def test_only_half_precision_variables_get_master_copies():
  """mp_wrapper.py:66: 'if var.dtype.base_dtype == tf.float16' — the fp32 vector has no FP32-master-copy twin."""
  d = np.load(os.path.join(HERE, "golden", "ref_exec_train_op.npz"))
  for case in rx.gen.TRAIN_OP_CASES:
    assert [str(n) for n in d[case + "/master_names"]] == [
        "Loss_Optimization/FP32-master-copy/ForwardPass/" + n for n in ("bias", "w1", "w2")]


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "train_op"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and "reproduced" in r.stdout, r.stdout + r.stderr
