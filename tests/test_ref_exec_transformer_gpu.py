"""The HIP Transformer path against the REFERENCE'S OWN CODE (no oracle in between).

tests/golden/ref_exec_transformer_d512.npz = open_seq2seq's TransformerEncoder / TransformerDecoder /
PaddedCrossEntropyLossWithSmoothing executed from their files (tests/golden/make_ref_exec.py) at the narrowest widths
the device kernels take: d_model 512, 8 heads of 64, filter 1024, 2 + 2 layers, vocabulary 90 padded to 96, a ragged
batch of 3. The device model is built from the same config, its variables are loaded BY THE REFERENCE'S NAMES through
the checkpoint importer (utils/checkpoint.py: Dense kernels transposed, q / k / v fused), and one forward + backward
pass on packed tokens must give the reference's logits, loss and variable gradients. Tolerances are the bf16 ones of
tests/test_transformer_e2e_gpu.py (the device stores activations and weights in bf16, the fixture is fp32): logits
rel-L2 3e-2, loss 2e-2, gradients: norm within 20 %, seeded projection within 4 x 0.2 x norm."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402

pytestmark = pytest.mark.gpu



def test_device_transformer_reproduces_the_reference_code(cuda):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.transformer_encoder import TransformerEncoder
  from openseq2seq_amd.decoders.transformer_decoder import TransformerDecoder
  from openseq2seq_amd.losses.sequence_loss import PaddedCrossEntropyLossWithSmoothing
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  from openseq2seq_amd.parts.transformer import packing
  from openseq2seq_amd.utils import checkpoint
  d, names = rx.load("transformer_d512")
  B, S, T, V, D, H, F, NL = [int(v) for v in d["config"]]
  store = FlatParams(cuda)
  enc = TransformerEncoder({"encoder_layers": NL, "hidden_size": D, "num_heads": H, "attention_dropout": 0.0,
                            "filter_size": F, "src_vocab_size": V, "relu_dropout": 0.0,
                            "layer_postprocess_dropout": 0.0, "remove_padding": True,
                            "pad_embeddings_2_eight": True, "dtype": "mixed"}, None, mode="train").build(store)
  dec = TransformerDecoder({"EOS_ID": 1, "layer_postprocess_dropout": 0.0, "num_hidden_layers": NL,
                            "hidden_size": D, "num_heads": H, "attention_dropout": 0.0, "relu_dropout": 0.0,
                            "filter_size": F, "batch_size": B, "tgt_vocab_size": V, "beam_size": 4, "alpha": 0.6,
                            "extra_decode_length": 5, "dtype": "mixed"}, None, mode="train").build(store)
  lossf = PaddedCrossEntropyLossWithSmoothing({"label_smoothing": float(d["label_smoothing"]), "tgt_vocab_size": V,
                                               "batch_size": B, "pad_embeddings_2_eight": True,
                                               "dtype": "mixed"}, None)
  store.finalize(need_m2=False)
  # ---- the reference's variables, by the reference's names -------------------------------------------------
  tf_arrays = rx.variables(d, names)
  used = set()
  for p in store.params:
    a = checkpoint.import_param(p.name, p.shape, p.kind, tf_arrays, getattr(p, "logical_out", None))
    assert a is not None and tuple(a.shape) == tuple(p.shape), (p.name, None if a is None else a.shape, p.shape)
    p.master.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda).view_as(p.master))
    for tf_name, _ in checkpoint.export_param(p.name, p.shape, p.kind, a, getattr(p, "logical_out", None)):
      used.add(tf_name)
  assert used == set(names), "the device model holds exactly the reference's variables, under the reference's names"
  store.refresh_compute_copies()
  # ---- one forward + backward pass on the packed batch --------------------------------------------------------
  src, sl, tgt, tl = d["src"], d["src_len"], d["tgt"], d["tgt_len"]
  batch = {'source_tensors': [torch.from_numpy(src).to(cuda), torch.from_numpy(sl).to(cuda)],
           'target_tensors': [torch.from_numpy(tgt).to(cuda), torch.from_numpy(tl).to(cuda)],
           'packed_source': packing.to_device(packing.pack_ids(src, sl), cuda),
           'packed_target': packing.to_device(packing.pack_ids(tgt, tl, shift_right=True), cuda)}
  tape = Tape()
  store.zero_grads()
  e = enc.encode({'source_tensors': batch['source_tensors'], 'tape': tape, 'seeds': SeedSeq(1),
                  'packed_source': batch['packed_source']})
  dd = dec.decode({'encoder_output': e, 'target_tensors': batch['target_tensors'], 'tape': tape,
                   'packed_target': batch['packed_target']})
  L = lossf.compute_loss({'decoder_output': dd, 'target_tensors': batch['target_tensors']})
  tape.backward()
  torch.cuda.synchronize()
  # ---- against the reference's numbers ----------------------------------------------------------------------------
  ref_loss = float(d["loss"])
  assert abs(float(L.cpu()[0]) - ref_loss) <= 2e-2 * abs(ref_loss), (float(L.cpu()[0]), ref_loss)
  lg = dd["logits"].float().cpu().numpy()
  ref_rows = np.concatenate([d["logits"][b, :tl[b]] for b in range(B)], 0)
  assert lg.shape == ref_rows.shape, (lg.shape, ref_rows.shape)
  r = rx.rel(lg, ref_rows)
  assert r < 3e-2, r
  # gradients. The fixture stores (norm, seeded projection) per variable; oracle/transformer.py reproduces those to
  # 1e-4 (asserted again here), so its full gradient TENSORS are the reference's: the device is held against them
  # tensor by tensor (cosine / rel-L2 of tests/test_transformer_e2e_gpu.py), and against the stored projections.
  from test_ref_exec_transformer import oracle_params
  from oracle import transformer as ot
  PE, PD, leaves = oracle_params(d, NL, names)
  s_ids, t_ids = torch.from_numpy(src).long(), torch.from_numpy(tgt).long()
  o_enc, o_bias = ot.encoder(s_ids, PE, H)
  ot.padded_xent_smoothing(ot.decoder_pass(t_ids, o_enc, o_bias, PD, H), t_ids, float(d["label_smoothing"])).backward()
  worst, worst_cos = 0.0, (1.0, "")
  for p in store.params:
    g = p.grad.detach().float().cpu().numpy()
    for tf_name, tf_g in checkpoint.export_param(p.name, p.shape, p.kind, g, getattr(p, "logical_out", None)):
      n = tf_name
      ref = leaves[n].grad.numpy()
      rx.check_gradient(d, n, ref, 1e-4)
      worst = max(worst, rx.check_gradient(d, n, tf_g, 0.2))
      cos = float((tf_g.astype(np.float64) * ref).sum() / (np.linalg.norm(tf_g) * np.linalg.norm(ref) + 1e-30))
      worst_cos = min(worst_cos, (cos, n))
      assert cos > 0.98 and rx.rel(tf_g, ref) < 0.2, (n, cos, rx.rel(tf_g, ref))
  print("device vs the reference's code: loss %.5f vs %.5f, logits rel-L2 %.2e, worst gradient cosine %.4f (%s), "
        "worst projection error %.2e" % (float(L.cpu()[0]), ref_loss, r, worst_cos[0], worst_cos[1], worst))


def test_device_beam_search_reproduces_the_reference_code(cuda, tmp_path):
  """TransformerDecoder.predict — the cached decode step under sequence_beam_search (decoders/transformer_decoder.py:
  232-326, parts/transformer/beam_search.py) — executed from the reference's files at d_model 512, 8 heads, V 96,
  beam 4, alpha 0.6 (tests/golden/ref_exec_transformer_infer_d512.npz) against the HIP beam search (K / V caches,
  fused top-k, hipGraph-replayed loop), restored from a TensorFlow-V2 checkpoint file that holds the reference graph's
  variables under the reference's names. A random model is an ill-conditioned search: the generator re-runs the
  reference six times with every matrix perturbed by 2^-7 relative and marks the rows whose winner never changes
  (`stable`); the device (bf16) must return those rows exactly, zero padding after EOS included, and the fp32 oracle
  (which returns all four rows exactly, tests/test_ref_exec_transformer.py) scores every device winner."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.transformer_encoder import TransformerEncoder
  from openseq2seq_amd.decoders.transformer_decoder import TransformerDecoder
  from openseq2seq_amd.utils import checkpoint, tensor_bundle
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_transformer_infer_d512.npz")))
  C = rx.gen.TRANSFORMER_BEAM
  B, S, V, D, H, F, NL = C["dims"]
  names = [str(n) for n in d["var_names"]]
  arrays = {n: rx.gen.transformer_beam_variable(n, tuple(int(v) for v in d["shape/" + n]), C["seed"]) for n in names}
  prefix = str(tmp_path / "model.ckpt-0")
  tensor_bundle.write_bundle(prefix, dict(arrays, global_step=np.asarray(0, np.int64)))
  store = FlatParams(cuda)
  enc = TransformerEncoder({"encoder_layers": NL, "hidden_size": D, "num_heads": H, "attention_dropout": 0.1,
                            "filter_size": F, "src_vocab_size": V, "relu_dropout": 0.1,
                            "layer_postprocess_dropout": 0.1, "remove_padding": True, "dtype": "mixed"}, None,
                           mode="infer").build(store)
  dec = TransformerDecoder({"EOS_ID": 1, "layer_postprocess_dropout": 0.1, "num_hidden_layers": NL, "hidden_size": D,
                            "num_heads": H, "attention_dropout": 0.1, "relu_dropout": 0.1, "filter_size": F,
                            "batch_size": B, "tgt_vocab_size": V, "beam_size": C["beam"], "alpha": 0.6,
                            "extra_decode_length": C["extra"], "dtype": "mixed"}, None, mode="infer").build(store)
  store.finalize()

  class M(object):
    params = {"dtype": "mixed"}
  M.store = store
  assert checkpoint.load(M(), prefix, restore_optimizer=False, strict=True) == []
  src, sl = torch.from_numpy(d["src"]).to(cuda), torch.from_numpy(d["src_len"]).to(cuda)
  e = enc.encode({"source_tensors": [src, sl]})
  out = dec.decode({"encoder_output": e})
  torch.cuda.synchronize()
  ids = out["outputs"][0].cpu().numpy()
  ref = d["ids"]
  T = max(ids.shape[1], ref.shape[1])
  pad = lambda a: np.concatenate([a, np.zeros((a.shape[0], T - a.shape[1]), a.dtype)], 1)      # noqa: E731
  ids, ref = pad(ids), pad(ref)
  exact = [bool(np.array_equal(ids[b], ref[b])) for b in range(B)]
  print("rows reproduced exactly:", exact, "stable under perturbation:", d["stable"].tolist())
  for b in range(B):
    if not exact[b]:
      print("row", b, "device", ids[b].tolist(), "reference", ref[b].tolist())
    if d["stable"][b]:
      assert exact[b], (b, ids[b].tolist(), ref[b].tolist())
  assert sum(exact) * 2 >= B
