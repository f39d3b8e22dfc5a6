"""The DeepSpeech2 oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_ds2.npz = open_seq2seq's DeepSpeech2Encoder._encode (encoders/ds2_encoder.py:158-401) executed
from its file by tests/golden/make_ref_exec.py: two conv2d + BatchNorm + ReLU blocks (strides [2, 2] / [1, 2], "SAME"),
the [B, T, F, C] -> [T, B, F * C] hand-over, the cuDNN GRU over the whole padded length (no sequence lengths), the row
convolution, dense + ReLU — a bidirectional two-layer case (ds2_large_8gpus.py's shape) and a unidirectional one with
row_conv. oracle/ds2.py must reproduce outputs (1e-5), output lengths (exact) and all variable gradients (1e-4) under
the surrogate loss sum(outputs * R). The cuDNN RNN is a TensorFlow library object: stand-in and oracle both use
torch.nn.GRU for the cell, so this pins the WIRING of the encoder (flattening order of frequency x channels, the
time-major transposes, the missing lengths, the row convolution's padding and BatchNorm), not the GRU cell equations
(those: tests/test_oracle_rnn.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402
from oracle import ds2 as ods  # noqa: E402

SC = "ForwardPass/ds2_encoder/"


@pytest.mark.parametrize("case", sorted(rx.gen.DS2_CASES))
def test_oracle_reproduces_the_reference_ds2_encoder(case):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_ds2.npz")))
  cfg = rx.gen.DS2_CASES[case]
  B, T, F, H, NH = [int(v) for v in d["dims"]]
  names = [str(n) for n in d[case + "/var_names"]]
  leaves = {n: torch.from_numpy(d["%s/var/%s" % (case, n)].copy()).requires_grad_(True) for n in names}
  W = {n[len(SC):]: t for n, t in leaves.items()}
  x = torch.from_numpy(rx.gen.tdnn_input(int(d[case + "/seed"]), d[case + "/src_len"], T, F))
  gru = torch.nn.GRU(8 * 5, H, num_layers=cfg["layers"], bidirectional=not cfg["unidir"], batch_first=True)
  pnames = [n for n, _ in gru.named_parameters()]
  assert sorted(pnames) == sorted(n[len("cudnn_gru/"):] for n in W if n.startswith("cudnn_gru/"))

  class G(object):          # the oracle calls gru(r): route the call through the autograd leaves
    def __call__(self, r):
      return torch.func.functional_call(gru, {n: W["cudnn_gru/" + n] for n in pnames}, (r,))
  if cfg["row_conv"]:
    # ds2_encode applies GRU -> FC; the row convolution sits between them (ds2_encoder.py:356-371)
    h = ods.ds2_encode(x, rx.gen.DS2_CONV, W, None, torch.eye(40), torch.zeros(40))     # conv stack only (relu is idempotent)
    r, _ = G()(h)
    r = ods.row_conv(r, W["row_conv/w"][:, 0, :, 0], W["row_conv/bn/gamma"], W["row_conv/bn/beta"])
    out = torch.relu(r @ W["fully_connected/kernel"] + W["fully_connected/bias"])
  else:
    out = ods.ds2_encode(x, rx.gen.DS2_CONV, W, G(), W["fully_connected/kernel"], W["fully_connected/bias"])
  assert rx.rel(out.detach().numpy(), d[case + "/out"]) < 1e-5
  # lengths: ceil(len / stride_T) per "SAME" layer (ds2_encoder.py:228-233)
  assert (-(-d[case + "/src_len"] // 2)).tolist() == d[case + "/out_len"].tolist()
  (out * torch.from_numpy(d[case + "/R"])).sum().backward()
  worst = 0.0
  for n in names:
    r = rx.rel(leaves[n].grad.numpy(), d["%s/grad/%s" % (case, n)])
    worst = max(worst, r)
    assert r < 1e-4, (n, r)
  print("%s: worst gradient rel-L2 vs the reference's code %.2e" % (case, worst))


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "ds2"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and "reproduced" in r.stdout, r.stdout + r.stderr
