"""The RNN NMT decoder oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_nmt_decoder.npz = open_seq2seq's RNNDecoderWithAttention._decode in train mode
(decoders/rnn_decoders.py:147-321: decoder embedding, single_cell, BahdanauAttention(normalize=True) inside an
AttentionWrapper, GNMTAttentionMultiCell with the old ('gnmt') and new ('gnmt_v2') attention wiring, gnmt_residual_fn
skip connections, output projection, TrainingHelper + dynamic_decode(impute_finished=True)) followed by
BasicSequenceLoss, executed from the reference's files by tests/golden/make_ref_exec.py (the LSTM cell class,
dynamic_decode and the helpers are TensorFlow library code, restated in oracle/ref_shim/tf1/rnn.py).
oracle/nmt.py:decoder_logits + basic_sequence_loss must reproduce logits (1e-5; zero rows past each target length),
loss and the gradient of every variable including the encoder outputs (1e-4)."""
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
from oracle import nmt as onmt  # noqa: E402

SC = "ForwardPass/rnn_decoder_with_attention/"
ATT = SC + "decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/"


@pytest.mark.parametrize("case", sorted(rx.gen.NMT_CASES))
def test_oracle_reproduces_the_reference_nmt_decoder(case):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_decoder.npz")))
  cfg = rx.gen.NMT_CASES[case]
  B, S, T, V, E, H, M, U = [int(v) for v in d["dims"]]
  names = [str(n) for n in d[case + "/var_names"]]
  leaf = {n: torch.from_numpy(d["%s/var/%s" % (case, n)].copy()).requires_grad_(True) for n in names}
  # reference layouts -> the oracle's (device) layouts: kernels [in, 4H] -> [4H, in], rows split by source
  k0 = leaf[ATT + "lstm_cell/kernel"].t()                       # [4H, E + M + H]: inputs | attention | h
  cell = {"w_in": k0[:, :E], "wcat": [k0[:, E:]], "b0": leaf[ATT + "lstm_cell/bias"], "bias": [None],
          "wq": leaf[ATT + "bahdanau_attention/query_layer/kernel"].t(),
          "wmem": leaf[SC + "AttentionMechanism/memory_layer/kernel"].t(),
          "v": leaf[ATT + "bahdanau_attention/attention_v"], "g": leaf[ATT + "bahdanau_attention/attention_g"],
          "b": leaf[ATT + "bahdanau_attention/attention_b"]}
  upper = []
  for i in range(1, cfg["layers"]):
    k = leaf[SC + "decoder/multi_rnn_cell/cell_%d/lstm_cell/kernel" % i].t()      # [4H, H + M + H]: below | attention | h
    upper.append({"wx_h": k[:, :H], "wx_a": k[:, H:H + M], "wh": k[:, H + M:],
                  "b": leaf[SC + "decoder/multi_rnn_cell/cell_%d/lstm_cell/bias" % i]})
  P = {"demb": leaf[SC + "DecoderEmbeddingMatrix"], "cell": cell, "upper": upper,
       "proj": leaf[SC + "decoder/dense/kernel"].t()}
  tgt, tgt_len = torch.from_numpy(d[case + "/tgt"]), torch.from_numpy(d[case + "/tgt_len"])
  logits = onmt.decoder_logits(P, leaf["ForwardPass/encoder_outputs"], torch.from_numpy(d[case + "/src_len"]), tgt,
                               tgt_len, attention_type=cfg["attention_type"], skip=cfg["skip"])
  ref = d[case + "/logits"]
  assert d[case + "/final_sequence_lengths"].tolist() == d[case + "/tgt_len"].tolist()
  for b in range(B):            # impute_finished: rows past the target length are zeros in the reference's output
    assert np.abs(ref[b, int(tgt_len[b]):]).max(initial=0.0) == 0.0
  live = (np.arange(T)[None, :] < d[case + "/tgt_len"][:, None])
  assert rx.rel(logits.detach().numpy()[live], ref[live]) < 1e-5
  loss = onmt.basic_sequence_loss(logits, tgt, tgt_len, B)
  assert abs(float(loss.detach()) - float(d[case + "/loss"])) < 1e-5 * abs(float(d[case + "/loss"]))
  loss.backward()
  worst = 0.0
  for n in names:
    r = rx.rel(leaf[n].grad.numpy(), d["%s/grad/%s" % (case, n)])
    worst = max(worst, r)
    assert r < 1e-4, (n, r)
  print("%s: worst gradient rel-L2 vs the reference's code %.2e" % (case, worst))


@pytest.mark.parametrize("case,fixture", [(c, "nmt_encoder") for c in sorted(rx.gen.NMT_ENC_CASES)] +
                         [("gnmt_like", "nmt_encoder_dev")])
def test_oracle_reproduces_the_reference_nmt_encoders(case, fixture):
  """BidirectionalRNNEncoderWithEmbedding / GNMTLikeEncoderWithEmbedding (encoders/rnn_encoders.py:221-305, 380-470)
  executed from the reference's file: embedding lookup, stacks of single_cell LSTM cells, bidirectional_dynamic_rnn with
  sequence lengths (outputs zero past each length, the backward direction reversed within each length), the
  unidirectional upper layers with ResidualWrapper from the second one on. Outputs 1e-5, gradients 1e-4. Fixture
  nmt_encoder_dev: the GNMT-like encoder at the widths the device test runs at (embedding 64, 64 units, 4 x 12)."""
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_%s.npz" % fixture)))
  B, S, V, E, H = [int(v) for v in d["dims"]]
  names = [str(n) for n in d[case + "/var_names"]]
  leaf = {n: torch.from_numpy(d["%s/var/%s" % (case, n)].copy()).requires_grad_(True) for n in names}

  def lyr(prefix, cin):
    k = leaf[prefix + "/kernel"].t()                      # [4H, In + H]
    return {"wx": k[:, :cin], "wh": k[:, cin:], "b": leaf[prefix + "/bias"]}
  ids, lens = torch.from_numpy(d[case + "/src"]), torch.from_numpy(d[case + "/src_len"])
  if case == "bidir":
    sc = "ForwardPass/bidir_rnn_encoder_with_emb/"
    P = {"emb": leaf[sc + "EncoderEmbeddingMatrix"]}
    for key in ("fw", "bw"):
      P[key] = [lyr(sc + "bidirectional_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell" % (key, i), E if i == 0 else H)
                for i in range(2)]
    out = onmt.encoder(P, ids, lens)
  else:
    sc = "ForwardPass/gnmt_encoder_with_emb/"
    P = {"emb": leaf[sc + "EncoderEmbeddingMatrix"],
         "l1fw": lyr(sc + "bidirectional_rnn/fw/lstm_cell", E), "l1bw": lyr(sc + "bidirectional_rnn/bw/lstm_cell", E),
         "uni": [lyr(sc + "rnn/multi_rnn_cell/cell_%d/lstm_cell" % i, 2 * H if i == 0 else H) for i in range(2)]}
    out = onmt.gnmt_like_encoder(P, ids, lens)
  assert rx.rel(out.detach().numpy(), d[case + "/out"]) < 1e-5
  (out * torch.from_numpy(d[case + "/R"])).sum().backward()
  worst = 0.0
  for n in names:
    r = rx.rel(leaf[n].grad.numpy(), d["%s/grad/%s" % (case, n)])
    worst = max(worst, r)
    assert r < 1e-4, (n, r)
  print("%s encoder: worst gradient rel-L2 vs the reference's code %.2e" % (case, worst))


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "nmt_decoder",
                      "nmt_encoder", "nmt_encoder_dev"], capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and r.stdout.count("reproduced") == 3, r.stdout + r.stderr
