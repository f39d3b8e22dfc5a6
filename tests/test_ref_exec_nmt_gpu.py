"""The HIP RNN NMT path (BASELINE configs[0]'s architecture) against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_nmt_full.npz = open_seq2seq's BidirectionalRNNEncoderWithEmbedding -> RNNDecoderWithAttention
(gnmt_v2: AttentionWrapper + normalised Bahdanau attention + GNMTAttentionMultiCell) -> BasicSequenceLoss executed from
their files (tests/golden/make_ref_exec.py) at E = H = 64, attention depth 128, two layers, V 30, a ragged batch of 4.
The device model is built from the same configuration and its parameters are filled from the reference's variables —
each LSTM cell's one kernel [in + H, 4H] split by rows into the device's input / attention / state matrices (gate order
i, j, f, o on both sides) — then one forward + backward pass must give the reference's encoder output, logits, loss and
variable gradients. bf16 tolerances of tests/test_nmt_e2e_gpu.py: outputs 3e-2 / 5e-2, loss 2e-2; gradients: tensor by
tensor against the oracle (which this test first holds to the fixture's gradient projections at 1e-4: its tensors ARE
the reference's) cosine 0.99 / rel-L2 0.12, and against the stored (norm, projection) pairs directly."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402

pytestmark = pytest.mark.gpu
ENC = "ForwardPass/bidir_rnn_encoder_with_emb/"
DEC = "ForwardPass/rnn_decoder_with_attention/"
ATT = DEC + "decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/"


def test_device_nmt_reproduces_the_reference_code(cuda):
  from test_nmt_e2e_gpu import _build, _oracle_params, _cmp
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  from oracle import nmt as onmt
  d, names = rx.load("nmt_full")
  C = rx.gen.NMT_FULL
  B, S, T, V, E, H, U = [C[k] for k in ("B", "S", "T", "V", "E", "H", "U")]
  M = 2 * H
  ref = {n: torch.from_numpy(np.array(a, np.float32)) for n, a in rx.variables(d, names).items()}
  store, enc, dec, lossf = _build(cuda, "gnmt_v2", V=V, E=E, H=H, layers=C["layers"])

  # reference variable -> (device parameter, how to cut it): the inverse of test_nmt_e2e_gpu._oracle_params
  slices = {}          # device param name -> list of (reference name, fn: reference-layout tensor -> this param's part)

  def put(p, value, refname, back):
    v = value.contiguous()
    assert v.numel() == p.master.numel(), (p.name, tuple(v.shape), tuple(p.master.shape))
    p.master.copy_(v.reshape(p.master.shape).to(cuda))
    slices[p.name] = (refname, back)
  put(enc.embedding.table, ref[ENC + "EncoderEmbeddingMatrix"], ENC + "EncoderEmbeddingMatrix", lambda g: g)
  for key, stack in zip(("fw", "bw"), enc.stacks):
    for i, l in enumerate(stack):
      kn = ENC + "bidirectional_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell/kernel" % (key, i)
      cin = E if i == 0 else H
      kT = ref[kn].t()                                   # [4H, in + H]
      put(l.wx[0], kT[:, :cin], kn, lambda g, cin=cin: g.t()[:, :cin])
      put(l.wh, kT[:, cin:], kn, lambda g, cin=cin: g.t()[:, cin:])
      bn = kn[:-len("kernel")] + "bias"
      put(l.bx, ref[bn], bn, lambda g: g)
  c = dec.cell
  k0 = ref[ATT + "lstm_cell/kernel"].t()                 # [4H, E + M + H]
  put(c.w_in, k0[:, :E], ATT + "lstm_cell/kernel", lambda g: g.t()[:, :E])
  put(c.wcat[0], k0[:, E:], ATT + "lstm_cell/kernel", lambda g: g.t()[:, E:])
  put(c.bias[0], ref[ATT + "lstm_cell/bias"], ATT + "lstm_cell/bias", lambda g: g)
  put(c.w_q, ref[ATT + "bahdanau_attention/query_layer/kernel"].t(), ATT + "bahdanau_attention/query_layer/kernel",
      lambda g: g.t())
  put(c.w_mem, ref[DEC + "AttentionMechanism/memory_layer/kernel"].t(), DEC + "AttentionMechanism/memory_layer/kernel",
      lambda g: g.t())
  for attr, leafname in (("v", "attention_v"), ("g", "attention_g"), ("b", "attention_b")):
    put(getattr(c, attr), ref[ATT + "bahdanau_attention/" + leafname], ATT + "bahdanau_attention/" + leafname,
        lambda g: g)
  put(dec.embedding.table, ref[DEC + "DecoderEmbeddingMatrix"], DEC + "DecoderEmbeddingMatrix", lambda g: g)
  for i, l in enumerate(dec.upper, start=1):
    kn = DEC + "decoder/multi_rnn_cell/cell_%d/lstm_cell/kernel" % i
    kT = ref[kn].t()                                     # [4H, H + M + H]: layer below | attention | h
    put(l.wx[0], kT[:, :H], kn, lambda g: g.t()[:, :H])
    put(l.wx[1], kT[:, H:H + M], kn, lambda g: g.t()[:, H:H + M])
    put(l.wh, kT[:, H + M:], kn, lambda g: g.t()[:, H + M:])
    put(l.bx, ref[kn[:-len("kernel")] + "bias"], kn[:-len("kernel")] + "bias", lambda g: g)
  pk = ref[DEC + "decoder/dense/kernel"].t()             # [V, H]
  proj = torch.zeros((dec.Vpad, H))
  proj[:V] = pk
  put(dec.proj, proj, DEC + "decoder/dense/kernel", lambda g: g.t()[:V])
  assert {p.name for p in store.params} == set(slices), sorted({p.name for p in store.params} ^ set(slices))
  assert {v[0] for v in slices.values()} == set(names), "every reference variable went into the device model"
  store.refresh_compute_copies()
  # what utils/checkpoint.py would write for this model IS the reference's variable list, value for value: loading a
  # reference checkpoint by name puts every array where the hand placement above put it
  from openseq2seq_amd.utils import checkpoint

  class _M(object):
    params = {"dtype": "float32"}
  _M.store = store
  written = checkpoint.model_variables(_M())
  opaque = lambda n: False           # cuDNN layers: one opaque buffer in TensorFlow
  for n in names:
    if not opaque(n):
      assert written[n].shape == tuple(ref[n].shape) and np.array_equal(written[n], ref[n].numpy()), n
  # ---- one step ------------------------------------------------------------------------------------------------
  src, src_len = torch.from_numpy(d["src"]), torch.from_numpy(d["src_len"])
  tgt, tgt_len = torch.from_numpy(d["tgt"]), torch.from_numpy(d["tgt_len"])
  tape = Tape()
  store.zero_grads()
  e = enc.encode({"source_tensors": [src.to(cuda), src_len.to(cuda)], "tape": tape, "seeds": SeedSeq(3)})
  dd = dec.decode({"encoder_output": e, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)], "tape": tape})
  L = lossf.compute_loss({"decoder_output": dd, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)]})
  tape.backward()
  torch.cuda.synchronize()
  # ---- against the reference's numbers -----------------------------------------------------------------------------
  live_s = (np.arange(S)[None, :] < d["src_len"][:, None])
  r_enc = rx.rel(e["outputs"].float().cpu().numpy()[live_s], d["enc_out"].astype(np.float32)[live_s])
  live_t = (np.arange(T)[None, :] < d["tgt_len"][:, None])
  r_log = rx.rel(dd["logits"].float().cpu().numpy()[..., :V][live_t], d["logits"][live_t])
  assert r_enc < 3e-2 and r_log < 5e-2, (r_enc, r_log)
  assert abs(float(L.item()) - float(d["loss"])) <= 2e-2 * abs(float(d["loss"])), (float(L.item()), float(d["loss"]))
  # the oracle on the same (fp32) variables: its gradients reproduce the fixture's projections, so they are the
  # reference's tensors; then the device against them, device layout
  P, D_, leaves = _oracle_params(store, enc, dec, V)        # leaves in device layout, from the device's bf16 copies
  for p in store.params:                                    # ... replaced by the exact fp32 reference values
    leaves[p.name].data.copy_(p.master.cpu().reshape(leaves[p.name].shape))
  enc_out = onmt.encoder(P, src, src_len)
  logits = onmt.decoder_logits(D_, enc_out, src_len, tgt, tgt_len, "gnmt_v2")[..., :V]
  onmt.basic_sequence_loss(logits, tgt, tgt_len, B).backward()
  assert rx.rel(logits.detach().numpy()[live_t], d["logits"][live_t]) < 1e-5
  # gradients per reference variable = the sum of its device parts mapped back (every part owns disjoint rows)
  worst = 0.0
  for n in names:
    parts_dev, parts_orc = [], []
    for p in store.params:
      if slices[p.name][0] != n:
        continue
      parts_dev.append((p, p.grad.detach().float().cpu()))
      parts_orc.append((p, leaves[p.name].grad if leaves[p.name].grad is not None else torch.zeros_like(leaves[p.name])))

    def assemble(parts):
      full = torch.zeros(tuple(int(v) for v in d["shape/" + n]))
      for p, g in parts:
        g = g.reshape(leaves[p.name].shape)
        if p is dec.proj:
          full.t()[:V] = g[:V]
        elif p in (c.w_in,):
          full.t()[:, :E] = g
        elif p is c.wcat[0]:
          full.t()[:, E:] = g
        elif any(p is l.wx[0] for st in enc.stacks for l in st):
          full.t()[:, :g.shape[1]] = g
        elif any(p is l.wh for st in enc.stacks for l in st):
          full.t()[:, full.shape[0] - H:] = g
        elif any(p is l.wx[0] for l in dec.upper):
          full.t()[:, :H] = g
        elif any(p is l.wx[1] for l in dec.upper):
          full.t()[:, H:H + M] = g
        elif any(p is l.wh for l in dec.upper):
          full.t()[:, H + M:] = g
        elif p in (c.w_q, c.w_mem):
          full.copy_(g.t())
        else:
          full.copy_(g.reshape(full.shape))
      return full.numpy()
    g_orc, g_dev = assemble(parts_orc), assemble(parts_dev)
    rx.check_gradient(d, n, g_orc, 1e-4)
    worst = max(worst, rx.check_gradient(d, n, g_dev, 0.2))
    _cmp(torch.from_numpy(g_dev).reshape(-1), torch.from_numpy(g_orc).reshape(-1), n)
  print("device vs the reference's code: encoder output %.2e, logits %.2e, loss %.4f vs %.4f, worst gradient projection "
        "error %.2e" % (r_enc, r_log, float(L.item()), float(d["loss"]), worst))


def test_device_gnmt_like_encoder_reproduces_the_reference_code(cuda, tmp_path):
  """The HIP GNMT-like encoder (one bidirectional LSTM layer, two unidirectional ones, a residual connection into the
  third; encoders/rnn_encoders.py:320-470) against the reference's OWN GNMTLikeEncoderWithEmbedding executed at widths
  the device's recurrent kernels take (tests/golden/ref_exec_nmt_encoder_dev.npz: embedding 64, 64 units, a ragged
  4 x 12 batch). The device encoder is restored BY THE REFERENCE'S VARIABLE NAMES (utils/checkpoint.load: each
  lstm_cell/kernel [in + H, 4H] split by rows into the device's input / state matrices), runs forward and — with the
  fixture's surrogate gradient d(loss)/d(outputs) = R — backward; outputs (live positions) 3e-2, every variable's
  gradient, mapped back to the reference's layout by the checkpoint writer's own translation, cosine 0.999 / rel-L2
  0.03 against the reference's tensors (measured 1.0000 / 0.006)."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.rnn_encoders import GNMTLikeEncoderWithEmbedding
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  from openseq2seq_amd.utils import checkpoint
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_encoder_dev.npz")))
  case = "gnmt_like"
  names = [str(n) for n in d[case + "/var_names"]]
  B, S, V, E, H = [int(v) for v in d["dims"]]
  store = FlatParams(cuda)
  enc = GNMTLikeEncoderWithEmbedding(
      {"src_vocab_size": V, "src_emb_size": E, "core_cell": "LSTMCell", "core_cell_params": {"num_units": H, "forget_bias": 1.0},
       "encoder_layers": 3, "encoder_use_skip_connections": False, "encoder_dp_input_keep_prob": 1.0,
       "encoder_dp_output_keep_prob": 1.0, "dtype": "mixed"}, None, mode="train").build(store)
  store.finalize()
  np.savez(str(tmp_path / "model.ckpt-0.npz"), **{n: d["%s/var/%s" % (case, n)] for n in names})

  class _Model(object):
    params = {"dtype": "float32"}
  _Model.store = store
  assert checkpoint.load(_Model(), str(tmp_path / "model.ckpt-0"), restore_optimizer=False) == []
  written = checkpoint.model_variables(_Model())
  assert set(written) == set(names)
  for n in names:
    assert np.array_equal(written[n], d["%s/var/%s" % (case, n)]), n
  # ---- forward + backward on the device ---------------------------------------------------------------------------
  src, src_len = torch.from_numpy(d[case + "/src"]).to(cuda), torch.from_numpy(d[case + "/src_len"]).to(cuda)
  tape = Tape()
  store.zero_grads()
  e = enc.encode({"source_tensors": [src, src_len], "tape": tape, "seeds": SeedSeq(3)})
  R = torch.from_numpy(d[case + "/R"])
  live = (torch.arange(S)[None, :] < torch.from_numpy(d[case + "/src_len"])[:, None])
  # (the reference's dynamic_rnn emits zeros past a sample's length: the surrogate gradient there meets a constant)
  e["outputs_act"].grad = (R * live[:, :, None]).to(torch.bfloat16).to(cuda)
  tape.backward()
  torch.cuda.synchronize()
  got = e["outputs"].float().cpu().numpy()
  r_out = rx.rel(got[live.numpy()], d[case + "/out"][live.numpy()])
  assert r_out < 3e-2, r_out
  assert float(np.abs(d[case + "/out"][~live.numpy()]).max()) == 0.0
  # gradients in the reference's layout: the checkpoint writer's translation applied to the gradient buffers
  masters = [p.master.clone() for p in store.params]
  try:
    for p in store.params:
      p.master.copy_(p.grad)
    g_written = checkpoint.model_variables(_Model())
  finally:
    for p, m in zip(store.params, masters):
      p.master.copy_(m)
  worst_cos, worst_rel = (1.0, ""), (0.0, "")
  for n in names:
    g, ref = g_written[n].astype(np.float64), d["%s/grad/%s" % (case, n)].astype(np.float64)
    cos = float((g * ref).sum() / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
    rl = rx.rel(g, ref)
    worst_cos, worst_rel = min(worst_cos, (cos, n)), max(worst_rel, (rl, n))
    assert cos > 0.999 and rl < 0.03, (n, cos, rl)     # measured 1.0000 / 0.006
  print("device GNMT-like encoder vs the reference's code: outputs %.2e, worst gradient cosine %.4f (%s), worst "
        "rel-L2 %.3f (%s)" % (r_out, worst_cos[0], worst_cos[1].split("/", 2)[-1], worst_rel[0], worst_rel[1].split("/", 2)[-1]))
