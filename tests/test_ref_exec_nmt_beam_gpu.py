"""The HIP RNN beam search, restored from a checkpoint under the REFERENCE'S variable names, against the reference's
own BeamSearchRNNDecoderWithAttention.

tests/golden/ref_exec_nmt_beam.npz = open_seq2seq's bidirectional encoder + BeamSearchRNNDecoderWithAttention over its own
BeamSearchDecoder (parts/rnns/rnn_beam_search_decoder.py), executed from the reference's files. Here: the variables of
that graph are written, under the names and in the shapes the reference's graph has them (one lstm_cell/kernel per
cell, bidirectional_rnn/fw/..., decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/...), into a TensorFlow-V2
checkpoint file; a Text2Text model in infer mode built from the device's own config restores it with
utils/checkpoint.load (strict) and decodes the fixture's source batch.

A random recurrent model is an ill-conditioned beam-search problem: at most steps the candidates at the beam boundary
are near-ties, and which of them survives decides later winners. The generator therefore re-runs THE REFERENCE with
every matrix perturbed by 2^-7 relative (four times a bf16 weight's rounding error), six times, and records the rows
whose winner never changes (`stable`). The device (bf16 weights and activations) must reproduce those rows exactly —
ids and length, finished rows and rows that run to the iteration cap — and their final scores to 2 % (8 % where the
matrices are scaled by 8: logits of magnitude 20); for the other
rows the fp32 score of the device's winner is printed beside the reference winner's."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402

pytestmark = pytest.mark.gpu


def _score(PD, enc_row, len_row, ids, cfg, att):
  """fp32 score of one hypothesis under the oracle decoder: sum of log-probabilities up to and including the first
  END (or all steps), divided by the GNMT length penalty of that many steps."""
  from oracle import nmt as onmt
  from oracle import rnn_beam_search as orb
  T = len(ids)
  end = [t for t in range(T) if ids[t] == cfg["END"]]
  n = end[0] + 1 if end else T
  prefix = torch.tensor([[2] + [int(v) for v in ids[:n - 1]]])
  with torch.no_grad():
    lg = onmt.decoder_logits(PD, enc_row[None], len_row[None], prefix, torch.tensor([n], dtype=torch.int32),
                             attention_type=att, skip=False)[0]
  lp = torch.log_softmax(lg, -1)
  total = float(sum(lp[t, int(ids[t])] for t in range(n)))
  # a surviving hypothesis is ranked with its final length, which counts the END token: the END candidate itself is
  # scored one shorter, but from the next step on the finished beam carries lengths + 1 (_beam_search_step)
  return total / float(orb.length_penalty(np.array([n]), cfg["lp"])[0])


@pytest.mark.parametrize("case", sorted(rx.gen.NMT_BEAM_CASES))
def test_device_beam_search_from_a_reference_named_checkpoint(cuda, tmp_path, case):
  from openseq2seq_amd.configs.nmt import nmt_small_config
  from openseq2seq_amd.decoders import BeamSearchRNNDecoderWithAttention
  from openseq2seq_amd.utils import checkpoint, tensor_bundle
  from oracle import nmt as onmt
  from test_ref_exec_nmt_beam import oracle_params
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_beam.npz")))
  cfg = rx.gen.NMT_BEAM_CASES[case]
  D = rx.gen.NMT_FULL
  B, V, E, H, U, NL, W = D["B"], D["V"], D["E"], D["H"], D["U"], D["layers"], cfg["beam"]
  seed = int(d["seed"])
  # ---- the reference graph's variables as a TensorFlow-V2 checkpoint ----------------------------------------------
  arrays = {}
  for n in [str(v) for v in d[case + "/var_names"]]:
    arrays[n] = rx.gen.nmt_beam_variable(n, tuple(int(x) for x in d["%s/shape/%s" % (case, n)]), seed, cfg["gain"])
  prefix = str(tmp_path / "model.ckpt-0")
  tensor_bundle.write_bundle(prefix, dict(arrays, global_step=np.asarray(0, np.int64)))
  # ---- the device model from its own config, restored by name -----------------------------------------------------
  cls, params = nmt_small_config(batch_size_per_gpu=B, vocab=V)
  params = copy.deepcopy(params)
  cell = {"num_units": H, "forget_bias": 1.0}
  params["encoder_params"].update(core_cell_params=cell, src_emb_size=E, encoder_layers=NL)
  params["decoder"] = BeamSearchRNNDecoderWithAttention
  params["decoder_params"].update(core_cell_params=dict(cell), tgt_emb_size=E, attention_layer_size=U,
                                  decoder_layers=NL, attention_type=cfg["att"], END_SYMBOL=cfg["END"],
                                  beam_width=W, length_penalty=cfg["lp"])
  model = cls(params, mode="infer", hvd=None, device=cuda)
  model.compile()
  assert checkpoint.load(model, prefix, restore_optimizer=False, strict=True) == []
  got = checkpoint.model_variables(model)
  assert {k for k in got if not k.startswith(checkpoint.MASTER_PREFIX)} == set(arrays)
  src, src_len = torch.from_numpy(d["src"]), torch.from_numpy(d["src_len"])
  enc = model._encoder.encode({"source_tensors": [src.to(cuda), src_len.to(cuda)]})
  out = model._decoder.decode({"encoder_output": enc})
  torch.cuda.synchronize()
  top = out["logits"].cpu().numpy()
  lengths = np.asarray(out["beam_sequence_lengths"])
  ref, ref_len, ref_fin = d[case + "/top_ids"], d[case + "/lengths"], d[case + "/finished"]
  T = ref.shape[1]
  assert top.shape == ref.shape, (top.shape, ref.shape)
  # ---- against the reference --------------------------------------------------------------------------------------
  PE, PD = oracle_params(d, case)
  with torch.no_grad():
    enc_o = onmt.encoder(PE, src, src_len)
  r_enc = rx.rel(enc["outputs"].float().cpu().numpy()[np.arange(src.shape[1])[None, :] < d["src_len"][:, None]],
                 enc_o.numpy()[np.arange(src.shape[1])[None, :] < d["src_len"][:, None]])
  # matrices times 8 put the LSTM gates deep into saturation: the bf16 error of the pre-activations is amplified
  assert r_enc < (3e-2 if cfg["gain"] <= 2.5 else 6e-2), r_enc
  exact = [bool(np.array_equal(top[b], ref[b])) for b in range(B)]
  report = []
  from oracle import rnn_beam_search as orb
  want = d[case + "/log_probs"] / orb.length_penalty(ref_len, cfg["lp"])
  for b in range(B):
    s_ref = _score(PD, enc_o[b], src_len[b], ref[b], cfg, cfg["att"])
    assert abs(s_ref - want[b, 0]) < 1e-4 * abs(want[b, 0]), "the scoring helper reproduces the reference's own score"
    s_dev = s_ref if exact[b] else _score(PD, enc_o[b], src_len[b], top[b], cfg, cfg["att"])
    report.append((b, exact[b], int(lengths[b, 0]), int(ref_len[b, 0]), round(s_dev, 4), round(s_ref, 4)))
    if not exact[b]:
      print("row", b, "device beams", out["predicted_ids"][b].cpu().numpy().T.tolist(), "scores",
            out["scores"][b].float().cpu().numpy().tolist(), "lengths", lengths[b].tolist(), "reference", ref[b].tolist())
    if d[case + "/stable"][b]:
      assert exact[b] and int(lengths[b, 0]) == int(ref_len[b, 0]), report
  print("%s: encoder %.2e; rows (exact, device length, reference length, fp32 score of the device / reference winner): %s"
        % (case, r_enc, report))
  assert d[case + "/stable"].any()
  # the last step's scores of the surviving beams = log_probs / penalty(lengths), for the rows reproduced exactly
  have = out["scores"].float().cpu().numpy()
  for b in range(B):
    if exact[b]:
      # logits of magnitude 20 at gain 8: a bf16 ulp of one logit is 0.06 - 0.12, per step
      tol = 2e-2 if cfg["gain"] <= 2.5 else 8e-2
      assert abs(have[b, 0] - want[b, 0]) <= tol * abs(want[b, 0]) + 2e-2, (b, have[b], want[b])
