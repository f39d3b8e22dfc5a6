"""The Jasper / TDNN oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_tdnn*.npz = open_seq2seq's TDNNEncoder._encode over conv_bn_actv / conv_bn_res_bn_actv and the
FullyConnectedCTCDecoder (dense, time-major logits, tf.nn.ctc_greedy_decoder), executed from the reference's files by
tests/golden/make_ref_exec.py on a Jasper-shaped stack (stride-2 first layer, three dense-residual blocks of two
repeats, a dilation-2 layer, a 1x1 layer; ragged batch, conv masks, BatchNorm on batch statistics). oracle/tdnn.py
must reproduce the encoder output, the shrunken lengths, the logits, the greedy ids (bit-exact), the updated BatchNorm
moving statistics and the gradient of every variable under the surrogate loss sum(logits * R) (tf.nn.ctc_loss is
TensorFlow-internal and not part of the reference's source). fp32 both sides: forward 1e-5, gradients 1e-4."""
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
from oracle import cnn, tdnn  # noqa: E402
from oracle import ctc_greedy  # noqa: E402

ENC = "ForwardPass/w2l_encoder/"
FC = "ForwardPass/fully_connected_ctc_decoder/fully_connected/"


def tdnn_input(d):
  return rx.gen.tdnn_input(int(d["seed"]), d["src_len"], int(d["T"]), int(d["F"]))


def oracle_forward(d, names, emulate_bf16=False):
  arrays = rx.variables(d, names)
  leaves = {n: torch.from_numpy(np.array(a, np.float32)).requires_grad_(True) for n, a in arrays.items()}
  weights = {n[len(ENC):]: t for n, t in leaves.items() if n.startswith(ENC)}
  layers = rx.gen.jasper_layers([int(v) for v in d["chans"]], [int(v) for v in d["kern"]])
  x = torch.from_numpy(tdnn_input(d))
  if emulate_bf16:
    x = x.to(torch.bfloat16).float()            # the device is handed bf16 features
  feats, out_len = tdnn.tdnn_encode(x, torch.from_numpy(d["src_len"]), layers, weights, activation="relu",
                                    use_conv_mask=True, bn_eps=1e-3, emulate_bf16=emulate_bf16)
  logits = (feats @ leaves[FC + "kernel"] + leaves[FC + "bias"]).permute(1, 0, 2)
  return leaves, weights, layers, feats, out_len, logits


@pytest.mark.parametrize("fixture", ["tdnn", "tdnn_wide"])
def test_oracle_reproduces_the_reference_tdnn(fixture):
  d, names = rx.load(fixture)
  leaves, weights, layers, feats, out_len, logits = oracle_forward(d, names)
  assert np.array_equal(out_len.numpy().astype(np.int32), d["out_len"])
  if "enc_out" in d:
    assert rx.rel(feats.detach().numpy(), d["enc_out"]) < 1e-5
  assert rx.rel(logits.detach().numpy(), d["logits"]) < 1e-5
  loss = (logits * torch.from_numpy(d["R"])).sum()
  assert abs(float(loss.detach()) - float(d["loss"])) < 1e-4 * max(1.0, abs(float(d["loss"])))
  loss.backward()
  worst = 0.0
  for n in names:
    worst = max(worst, rx.check_gradient(d, n, leaves[n].grad.numpy(), 1e-4))
  # greedy decode (fc_decoders.py:244-250: merge_repeated=True, blank = V - 1): ids bit-exact
  ids, lens, _ = ctc_greedy.greedy_numpy(logits.detach().numpy(), out_len.numpy(), blank=int(d["V"]) - 1)
  for b in range(len(d["src_len"])):
    ref = [int(v) for v in d["greedy_ids"][b] if v >= 0]
    assert [int(v) for v in ids[b, :lens[b]]] == ref, (b, ids[b], ref)
  print("%s: worst gradient error vs the reference's code %.2e" % (fixture, worst))


@pytest.mark.parametrize("fixture", ["tdnn", "tdnn_wide"])
def test_oracle_ctc_loss_follows_the_reference_wrapper(fixture):
  """CTCLoss._compute_loss (losses/ctc_loss.py:44-88) executed on the reference's logits: dense_to_sparse of the
  label matrix, ignore_longer_outputs_than_inputs (the last sample's transcript does not fit: zero loss, zero
  gradient), mask_nans, mean over the whole batch. tf.nn.ctc_loss itself is TensorFlow-internal — the stand-in and the
  oracle both sit on torch's ctc_loss, so this pins the WRAPPER (what is averaged, what is zeroed), the values of the
  recursion are pinned elsewhere (tests/test_oracle_ctc.py)."""
  d, names = rx.load(fixture)
  logits = torch.from_numpy(d["logits"]).requires_grad_(True)
  out_len, labels, label_len = d["out_len"], d["labels"], d["label_len"]
  assert label_len[-1] > out_len[-1]
  # fc_ctc applies the FC layer itself: hand it the logits through an identity "layer"
  lg, loss = tdnn.fc_ctc(logits.permute(1, 0, 2), out_len, torch.eye(logits.shape[2]), torch.zeros(logits.shape[2]),
                         labels, label_len)
  assert abs(float(loss.detach()) - float(d["ctc_loss"])) < 1e-5 * abs(float(d["ctc_loss"]))
  loss.backward()
  g = logits.grad.numpy()
  assert rx.rel(g, d["ctc_dlogits"]) < 1e-5
  assert np.abs(d["ctc_dlogits"][:, -1, :]).max() == 0.0, "the sample that does not fit contributes nothing"


@pytest.mark.parametrize("fixture", ["tdnn", "tdnn_wide"])
def test_oracle_moving_statistics_follow_the_reference(fixture):
  """UPDATE_OPS of tf.layers.batch_normalization on the 4-D (fused) path the reference forces with expand_dims
  (conv_blocks.py:139-156): moving = moving * momentum + batch * (1 - momentum), momentum 0.90, the batch variance
  with Bessel's correction. One step from the initial 0 / 1."""
  d, names = rx.load(fixture)
  leaves, weights, layers, *_ = oracle_forward(d, names)
  # recompute every layer's pre-BatchNorm tensor with the oracle's own pieces, layer by layer, through the
  # recorded names: run the encoder again with a hook on batch_norm_train
  seen = {}
  orig = cnn.batch_norm_train

  def spy(y, gamma, beta, eps, *a, **k):
    out = orig(y, gamma, beta, eps, momentum=0.90, moving_mean=torch.zeros_like(gamma), moving_var=torch.ones_like(gamma))
    for key, t in weights.items():
      if t is gamma:
        seen[key[:-len("/gamma")]] = (out[3].detach().numpy(), out[4].detach().numpy())
    return out
  cnn.batch_norm_train = spy
  try:
    tdnn.tdnn_encode(torch.from_numpy(tdnn_input(d)), torch.from_numpy(d["src_len"]), layers, weights, activation="relu",
                     use_conv_mask=True, bn_eps=1e-3)
  finally:
    cnn.batch_norm_train = orig
  moving = [str(n) for n in d["moving_names"]]
  assert len(seen) * 2 == len(moving)
  for n in moving:
    scope, leaf = n[len(ENC):].rsplit("/", 1)
    got = seen[scope][0 if leaf == "moving_mean" else 1]
    assert rx.rel(got, d["moving/" + n]) < 1e-5, (n, rx.rel(got, d["moving/" + n]))


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixtures():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "tdnn",
                      "tdnn_wide"], capture_output=True, text=True, timeout=900)
  assert r.returncode == 0 and r.stdout.count("reproduced") == 2, r.stdout + r.stderr
