"""The HIP Jasper / TDNN path against the REFERENCE'S OWN CODE (no oracle in between).

tests/golden/ref_exec_tdnn_wide.npz = open_seq2seq's TDNNEncoder + FullyConnectedCTCDecoder executed from their files
(tests/golden/make_ref_exec.py) on a Jasper-shaped stack at widths the device convolution kernels run at (64 features;
128 / 192 / 256 / 320 / 384 channels; K 11 / 13 / 17 / 29, dilation 2, the 1x1 layer; stride-2 first layer; three
dense-residual blocks; ragged batch of 3 with conv masks; BatchNorm on batch statistics). The device encoder and
decoder are built from the same layer list, their variables are loaded BY THE REFERENCE'S NAMES through the checkpoint
importer, and one training-mode forward + backward pass must give the reference's encoder output, output lengths
(exact), logits, updated BatchNorm moving statistics and — with the fixture's surrogate d(loss)/d(logits) = R injected
where CTCLoss would deposit its gradient — the gradient of every variable (the encoder output itself is compared through oracle/tdnn.py, which the same test
holds to the fixture's logits at 1e-5). bf16 tolerances of
tests/test_jasper_e2e_gpu.py (b): outputs 3e-2 rel-L2, gradients: norm within 20 %, projection 4 x 0.2 x norm."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture
def deterministic_kernels():
  """No fp32 atomics in the parameter-gradient kernels: the same numbers every run (the gradient bounds below are
  measured values with one rounding step of margin, not envelopes of run-to-run noise)."""
  from openseq2seq_amd import capi
  was = capi.deterministic()
  capi.set_deterministic(True)
  yield
  capi.set_deterministic(was)


@pytest.mark.parametrize("residual_path", ["algebra", "branches"])
def test_device_tdnn_reproduces_the_reference_code(cuda, deterministic_kernels, monkeypatch, residual_path):
  """residual_path: the dense-residual block ends as GEMMs over the concatenated block inputs (the default,
  parts/cnns/dense_residual.py) or branch by branch. Bounds that are measured values of the branch path with one
  rounding step of margin — the largest elementwise logit error and the distance to the oracle that emulates the
  BRANCH path's bf16 storage points — carry the algebra path's own measured values."""
  from openseq2seq_amd.parts.cnns import dense_residual
  monkeypatch.setattr(dense_residual, "ENABLED", residual_path == "algebra")
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.decoders.fc_decoders import FullyConnectedCTCDecoder, decode_outputs
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.utils import checkpoint
  d, names = rx.load("tdnn_wide")
  layers = rx.gen.jasper_layers([int(v) for v in d["chans"]], [int(v) for v in d["kern"]])
  V, F = int(d["V"]), int(d["F"])
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                     "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, F)
  assert (enc._dres_plan is not None) == (residual_path == "algebra")
  dec = FullyConnectedCTCDecoder({"tgt_vocab_size": V, "dtype": "mixed"}, None, mode="train").build(store, enc.output_dim)
  store.finalize()
  tf_arrays = rx.variables(d, names)
  used = set()
  for p in store.params:
    lo = getattr(p, "logical_out", None)
    a = checkpoint.import_param(p.name, p.shape, p.kind, tf_arrays, lo)
    assert a is not None and tuple(a.shape) == tuple(p.shape), (p.name, None if a is None else a.shape, p.shape)
    p.master.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda).view_as(p.master))
    used |= {n for n, _ in checkpoint.export_param(p.name, p.shape, p.kind, a, lo)}
  assert used == set(names), (sorted(used ^ set(names)))
  store.refresh_compute_copies()
  # ---- forward (train mode) + backward with the surrogate gradient -------------------------------------------
  from test_ref_exec_tdnn import oracle_forward, tdnn_input
  x = torch.from_numpy(tdnn_input(d)).to(torch.bfloat16).to(cuda)
  lens = torch.from_numpy(d["src_len"]).to(cuda)
  tape = Tape()
  e = enc.encode({"source_tensors": [x, lens], "tape": tape, "seed": 3})
  dd = dec.decode({"encoder_output": e, "tape": tape})
  # CTCLoss on the device's logits vs the reference's CTCLoss on its own (one sample's transcript does not fit its
  # output length: zero loss, as ignore_longer_outputs_than_inputs makes it) — then the surrogate gradient replaces
  # the one the loss deposited
  from openseq2seq_amd.losses.ctc_loss import CTCLoss
  ctc = CTCLoss({"dtype": "mixed"}, None).compute_loss(
      {"decoder_output": dd, "target_tensors": [torch.from_numpy(d["labels"]).to(cuda),
                                                torch.from_numpy(d["label_len"]).to(cuda)]})
  assert abs(float(ctc.cpu()[0]) - float(d["ctc_loss"])) < 2e-2 * abs(float(d["ctc_loss"])), (float(ctc.cpu()[0]),
                                                                                             float(d["ctc_loss"]))
  Tq, B = d["logits"].shape[0], d["logits"].shape[1]
  dl = torch.zeros((B, Tq, dec.Vpad), dtype=torch.float32)
  dl[:, :, :V] = torch.from_numpy(d["R"]).permute(1, 0, 2)
  dd["_dlogits_sink"]["dlogits_bf16"] = dl.to(torch.bfloat16).to(cuda)
  store.zero_grads()
  tape.backward()
  torch.cuda.synchronize()
  # ---- against the reference's numbers ----------------------------------------------------------------------------
  assert e["src_length"].cpu().numpy().astype(np.int32).tolist() == d["out_len"].tolist()
  leaves, _, _, o_feats, _, o_logits = oracle_forward(d, names)
  assert rx.rel(o_logits.detach().numpy(), d["logits"]) < 1e-5          # the oracle IS the reference here (fp32)
  r_enc = rx.rel(e["outputs"].float().cpu().numpy(), o_feats.detach().numpy())
  lg = dd["logits"].float().cpu().numpy()
  assert lg.shape == d["logits"].shape, (lg.shape, d["logits"].shape)
  r_log = rx.rel(lg, d["logits"])
  assert r_enc < 3e-2 and r_log < 3e-2, (r_enc, r_log)
  # greedy decode on the device (fc_decoders.py:244-250) vs the reference's tf.nn.ctc_greedy_decoder ids: bf16
  # storage may flip an argmax between two near-equal logits, so the frame-wise argmax must agree on >= 97 % of the
  # live frames and the decoded strings are reported
  live = np.arange(Tq)[:, None] < d["out_len"][None, :]
  agree = float((lg.argmax(-1) == d["logits"].argmax(-1))[live].mean())
  assert agree >= 0.97, agree
  # ... element by element: every live logit within EPS x the largest magnitude of its frame (10 bf16 ulps — measured
  # worst 0.0293 with the branch-by-branch residual path, 0.0327 with the dense-residual algebra: the logits sit
  # behind nine BatchNorm layers of bf16 activations). A frame whose runner-up reference logits are more
  # than 2 EPS below the top MUST then decode to the reference's symbol; a close call may go to any symbol within
  # 2 EPS. The decoded STRINGS (fc_decoders.py:244-251: argmax, merge repeats, drop the blank) are therefore held
  # against the reference's own tf.nn.ctc_greedy_decoder output exactly: equal on every sample without a close call,
  # and otherwise a member of the set of strings the close calls allow (enumerated: <= 2^20 per sample; on this
  # fixture 4 - 12 close frames of 20 - 48 per sample, 24 - 31 104 admissible strings).
  import itertools
  EPS = (1.25 if residual_path == "algebra" else 1.0) * 2.0 ** -5
  ref_lg = d["logits"]
  fmax = np.abs(ref_lg).max(-1)
  err = np.abs(lg - ref_lg).max(-1) / fmax
  assert float(err[live].max()) <= EPS, float(err[live].max())
  ids = decode_outputs(dec, dd)[0]            # (dense ids [B, T] padded with -1, lengths [B])
  dev_ids, dev_len = ids[0].cpu().numpy(), ids[1].cpu().numpy()
  blank = V - 1

  def collapse(path):
    out, prev = [], -1
    for sym in path:
      if sym != prev and sym != blank:
        out.append(int(sym))
      prev = sym
    return tuple(out)

  exact = close_calls = 0
  for b in range(B):
    n = int(d["out_len"][b])
    mine = tuple(int(v) for v in dev_ids[b, :int(dev_len[b])])
    ref_ids = tuple(int(v) for v in np.asarray(d["greedy_ids"][b]).ravel() if int(v) >= 0)
    assert collapse(ref_lg[:n, b].argmax(-1)) == ref_ids           # what the fixture's ids mean
    cands = [np.nonzero(ref_lg[t, b] >= ref_lg[t, b].max() - 2 * EPS * fmax[t, b])[0].tolist() for t in range(n)]
    ways = int(np.prod([len(c) for c in cands], dtype=np.float64))
    close_calls += sum(len(c) > 1 for c in cands)
    if ways == 1:
      assert mine == ref_ids, (b, mine, ref_ids)
      exact += 1
    else:
      assert ways <= 1 << 20, (b, ways)
      allowed = {collapse(path) for path in itertools.product(*cands)}
      assert mine in allowed, (b, mine, ref_ids, ways)
  decode_report = "%d of %d strings equal by necessity, %d close frames of %d" % (exact, B, close_calls, int(live.sum()))
  # moving statistics after ONE training-mode forward from 0 / 1 (momentum 0.90, Bessel-corrected batch variance)
  worst_mv = 0.0
  for n in [str(v) for v in d["moving_names"]]:
    got = store.state[n].float().cpu().numpy()
    worst_mv = max(worst_mv, rx.rel(got, d["moving/" + n]))
  assert worst_mv < 2e-2, worst_mv
  # gradients. The fixture stores (norm, seeded projection) per variable; oracle/tdnn.py reproduces those to 1e-4
  # (asserted again here), so its full gradient TENSORS are the reference's: the device is held against them
  # tensor by tensor (cosine / rel-L2), and against the stored projections directly.
  (o_logits * torch.from_numpy(d["R"])).sum().backward()
  # ... and the same oracle with the device's bf16 STORAGE points emulated (weights rounded to bf16 as the device's
  # compute copies are): separates storage noise through nine BatchNorm layers from logic errors (tight bound)
  d16 = dict(d)
  arrays16 = {n: (torch.from_numpy(np.array(a, np.float32)).to(torch.bfloat16).float().numpy() if a.ndim >= 2 else a)
              for n, a in rx.variables(d, names).items()}
  for n, a in arrays16.items():
    d16["var/" + n] = a
  leaves16, _, _, _, _, l16 = oracle_forward(d16, names, emulate_bf16=True)
  (l16 * torch.from_numpy(d["R"]).to(torch.bfloat16).float()).sum().backward()
  worst, worst_cos, worst_cos16 = 0.0, (1.0, ""), (1.0, "")
  for p in store.params:
    lo = getattr(p, "logical_out", None)
    g = p.grad.detach().float().cpu().numpy()
    for tf_name, tf_g in checkpoint.export_param(p.name, p.shape, p.kind, g, lo):
      ref = leaves[tf_name].grad.numpy()
      rx.check_gradient(d, tf_name, ref, 1e-4)
      worst = max(worst, rx.check_gradient(d, tf_name, tf_g, 0.3))

      def cosine(a, b):
        return float((a.astype(np.float64) * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
      cos, cos16 = cosine(tf_g, ref), cosine(tf_g, leaves16[tf_name].grad.numpy())
      worst_cos, worst_cos16 = min(worst_cos, (cos, tf_name)), min(worst_cos16, (cos16, tf_name))
      # vs the reference's fp32 numbers: measured worst 0.9797 / 0.204 on the first layer's gamma (nine BatchNorm
      # layers of 288 rows above it: a ReLU mask flipped by one bf16 ulp carries a full-size gradient error); vs the
      # bf16-storage emulation, whose masks flip with the device's: measured worst 0.9941 / 0.109 (first kernel)
      # (three runs: 0.9766 - 0.9797 and 0.9929 - 0.9941; the weight-gradient atomics differ in the last bit run to
      # run and the masks amplify it — bounds with margin)
      assert cos > 0.96 and rx.rel(tf_g, ref) < 0.3, (tf_name, cos, rx.rel(tf_g, ref))
      # (round 6: the test runs in deterministic mode — no atomics, the same numbers every run — and the bound
      # against the storage-emulating oracle is the measured 0.9941 / 0.109 with one rounding step of margin)
      c16, r16 = (0.99, 0.12) if residual_path == "branches" else (0.975, 0.25)
      assert cos16 > c16 and rx.rel(tf_g, leaves16[tf_name].grad.numpy()) < r16, \
          (tf_name, cos16, rx.rel(tf_g, leaves16[tf_name].grad.numpy()))
  print(residual_path, "device vs the reference's code: CTC loss %.4f vs %.4f, encoder output %.2e, logits %.2e, argmax agreement %.3f, moving statistics "
        "%.2e, worst gradient cosine %.4f (%s) [%.4f with bf16 storage emulated (%s)], worst projection error %.2e; "
        "decoded %s" % (float(ctc.cpu()[0]), float(d["ctc_loss"]), r_enc, r_log, agree, worst_mv, worst_cos[0], worst_cos[1], worst_cos16[0], worst_cos16[1],
                        worst, decode_report))
