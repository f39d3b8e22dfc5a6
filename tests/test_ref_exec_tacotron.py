"""The Tacotron 2 decoder oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_tacotron_decoder.npz = open_seq2seq's Tacotron2Decoder._decode in train mode
(decoders/tacotron2_decoder.py:257-567) executed from the reference's files by tests/golden/make_ref_exec.py:
pre-net (two Dense + ReLU layers with their always-on dropout — the masks the run drew are part of the fixture),
two LSTM cells inside AttentionWrapper(output_attention="both", alignment_history=True) over
LocationSensitiveAttention (parts/rnns/attention_wrapper.py:641-878: query layer, k=1 memory layer, Chorowski location
layer — 32 taps x 32 filters over the CUMULATIVE alignments — score bias on / off, scores masked to -inf past the
source lengths), TacotronDecoder / TacotronTrainingHelper (parts/tacotron/*.py) under dynamic_decode, output and
stop-token projections, the post-net through conv_bn_actv (BatchNorm on batch statistics, tanh / linear), the magnitude
branch (two conv + BN + ReLU layers of 256 / 512 channels, exp, 1x1 projection). The LSTM cell class and dynamic_decode
are TensorFlow library code (oracle/ref_shim/tf1/rnn.py). oracle/tacotron.py:decoder must reproduce the mel frames, the
post-net output, stop logits, magnitude frames, alignments (1e-5) and the gradient of every variable incl. the encoder
outputs (1e-4) under the surrogate loss sum(outputs * R)."""
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
from oracle import tacotron as otaco  # noqa: E402

SC = "ForwardPass/tacotron_2_decoder/"
AW = SC + "decoder/attention_wrapper/"


@pytest.mark.parametrize("case", ["location", "location_bias"])
def test_oracle_reproduces_the_reference_tacotron_decoder(case):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_tacotron_decoder.npz")))
  B, S, T, M, H, U, P_, NMEL, NMAG, K, F = [int(v) for v in d["dims"]]
  names = [str(n) for n in d[case + "/var_names"]]
  seed = int(d[case + "/seed"])
  leaf = {}
  for n in names:
    k = "%s/var/%s" % (case, n)
    a = d[k] if k in d else rx.gen.seeded_array(n, tuple(int(v) for v in d["%s/shape/%s" % (case, n)]), seed)
    leaf[n] = torch.from_numpy(np.array(a, np.float32)).requires_grad_(True)

  def conv(prefix):          # TF [K, Cin, Cout] -> the oracle's [K, Cout, Cin]
    return (leaf[prefix + "/kernel"].permute(0, 2, 1), leaf[prefix + "/bn/gamma"], leaf[prefix + "/bn/beta"])
  k0 = leaf[AW + "multi_rnn_cell/cell_0/lstm_cell/kernel"].t()          # [4H, P + M + H]: pre-net | attention | h
  cell = {"w_in": k0[:, :P_], "b0": leaf[AW + "multi_rnn_cell/cell_0/lstm_cell/bias"],
          "wcat": [k0[:, P_:], leaf[AW + "multi_rnn_cell/cell_1/lstm_cell/kernel"].t()],
          "bias": [None, leaf[AW + "multi_rnn_cell/cell_1/lstm_cell/bias"]],
          "wq": leaf[AW + "location_attention/query_layer/kernel"].t(),
          "wmem": leaf[SC + "AttentionMechanism/memory_layer/kernel"][0].t(),
          "v": leaf[AW + "location_attention/attention_v"],
          "b": leaf.get(AW + "location_attention/attention_bias"),
          "conv_w": leaf[AW + "location_attention/location_conv/kernel"][:, 0, :],
          "conv_b": leaf[AW + "location_attention/location_conv/bias"],
          "dense_w": leaf[AW + "location_attention/location_dense/kernel"][0]}
  assert (cell["b"] is not None) == (case == "location_bias")
  P = {"prenet": [(leaf[SC + "decoder/prenet_%d/kernel" % i].t(), leaf[SC + "decoder/prenet_%d/bias" % i])
                  for i in (1, 2)],
       "cell": cell, "out_w": leaf[SC + "decoder/output_proj/kernel"].t(), "out_b": leaf[SC + "decoder/output_proj/bias"],
       "stop_w": leaf[SC + "decoder/stop_token_proj/kernel"].t(), "stop_b": leaf[SC + "decoder/stop_token_proj/bias"],
       "postnet": [conv(SC + "conv%d" % i) for i in (1, 2, 3)],
       "mag": {"c0": conv(SC + "conv_0"), "c1": conv(SC + "conv_1"), "proj": leaf[SC + "post_net_proj/kernel"][0].t()}}
  spec = torch.from_numpy(d[case + "/spec"])
  masks = [torch.from_numpy(d[case + "/prenet_mask0"]), torch.from_numpy(d[case + "/prenet_mask1"])]
  assert set(np.unique(d[case + "/prenet_mask0"]).tolist()) <= {0.0, 2.0}       # keep 0.5: kept values doubled
  out = otaco.decoder(P, leaf["ForwardPass/encoder_outputs"], torch.from_numpy(d[case + "/src_len"]),
                      spec[:, :, :NMEL], ["tanh", "tanh", None], bn_eps=1e-5, exp_mag=True, prenet_masks=masks)
  # mask_decoder_sequence: a sample is finished once time + 1 >= its spectrogram length; impute_finished=False, so
  # every sample keeps running until the longest one ends
  assert d[case + "/lens"].tolist() == d[case + "/spec_len"].tolist()
  for key, ref in (("mel", "mel"), ("post", "post"), ("stop", "stop"), ("mag", "mag"), ("align", "align")):
    r = rx.rel(out[key].detach().numpy(), d["%s/%s" % (case, ref)])
    assert r < 1e-5, (key, r)
  # scores are masked to -inf past the source lengths: those alignments are exact zeros in the reference's output
  for b in range(B):
    assert np.abs(d[case + "/align"][b, :, int(d[case + "/src_len"][b]):]).max(initial=0.0) == 0.0
  loss = sum((out[k] * torch.from_numpy(d["%s/R%d" % (case, i)])).sum()
             for i, k in enumerate(("mel", "post", "stop", "mag")))
  assert abs(float(loss.detach()) - float(d[case + "/loss"])) < 1e-4 * max(1.0, abs(float(d[case + "/loss"])))
  loss.backward()
  worst = 0.0
  dd = {k[len(case) + 1:]: v for k, v in d.items() if k.startswith(case + "/")}
  dd["seed"] = d[case + "/seed"]
  for n in names:
    worst = max(worst, rx.check_gradient(dd, n, leaf[n].grad.numpy(), 1e-4))
  print("%s: worst gradient error vs the reference's code %.2e" % (case, worst))


def test_oracle_reproduces_the_reference_tacotron_encoder_with_style_tokens():
  """Tacotron2Encoder._encode + _embed_style (encoders/tacotron2_encoder.py:104-505) executed from the reference's
  file: embedding, three conv + BatchNorm + ReLU layers, the cuDNN bidirectional LSTM over the whole padded length
  (torch.nn.LSTM on both sides: its wiring is what is pinned), and the global-style-token branch — two conv2d blocks
  over the style spectrogram, the [B, T, F, C] flattening by tf.unstack / concat, tf.nn.rnn_cell.GRUCell under
  dynamic_rnn with the shrunken lengths (final state), Dense(128, tanh), the reference's multi-head Attention in
  "bahdanau" mode over tanh(token_embeddings) — tiled over time and concatenated to the encoder output.
  oracle/tacotron.py:encoder + oracle/gst.py:style_encoder: outputs 1e-5, gradients of all 36 variables 1e-4."""
  from oracle import gst as ogst
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_tacotron_encoder.npz")))
  D = {str(k): int(v) for k, v in zip(d["dim_names"], d["dims"])}
  names = [str(n) for n in d["var_names"]]
  leaf = {n: torch.from_numpy(d["var/" + n].copy()).requires_grad_(True) for n in names}
  E_ = "ForwardPass/tacotron2_encoder/"
  ST = E_ + "style_encoder/"
  PS = {"convs": [(leaf[ST + "conv%d/kernel" % i], leaf[ST + "conv%d/bn/gamma" % i], leaf[ST + "conv%d/bn/beta" % i])
                  for i in (1, 2)],
        "wg": leaf[ST + "rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel"],
        "bg": leaf[ST + "rnn/multi_rnn_cell/cell_0/gru_cell/gates/bias"],
        "wc": leaf[ST + "rnn/multi_rnn_cell/cell_0/gru_cell/candidate/kernel"],
        "bc": leaf[ST + "rnn/multi_rnn_cell/cell_0/gru_cell/candidate/bias"],
        "ref_w": leaf[ST + "reference_activation/kernel"], "ref_b": leaf[ST + "reference_activation/bias"],
        "tokens": leaf[ST + "token_embeddings"], "wq": leaf[ST + "attention/q/kernel"],
        "wk": leaf[ST + "attention/k/kernel"], "wv": leaf[ST + "attention/v/kernel"],
        "wo": leaf[ST + "attention/output_transform/kernel"], "att_v": leaf[ST + "attention/attention_v"]}
  style = ogst.style_encoder(PS, torch.from_numpy(d["style"]), torch.from_numpy(d["style_len"]),
                             rx.gen.TACO_STYLE_CONV, D["HEADS"])
  lstm = torch.nn.LSTM(D["C"], D["H"], num_layers=1, bidirectional=True, batch_first=True)
  pn = [n for n, _ in lstm.named_parameters()]

  class L(object):
    def __call__(self, x):
      return torch.func.functional_call(lstm, {n: leaf[E_ + n] for n in pn}, (x,))
  P = {"emb": leaf[E_ + "EncoderEmbeddingMatrix"],
       "convs": [(leaf[E_ + "conv%d/kernel" % i].permute(0, 2, 1), leaf[E_ + "conv%d/bn/gamma" % i],
                  leaf[E_ + "conv%d/bn/beta" % i]) for i in (1, 2, 3)]}
  out = otaco.encoder(P, torch.from_numpy(d["text"]), L(), bn_eps=1e-5, style=style)
  assert d["out_len"].tolist() == d["text_len"].tolist()
  assert rx.rel(out.detach().numpy(), d["out"]) < 1e-5
  (out * torch.from_numpy(d["R"])).sum().backward()
  worst = 0.0
  for n in names:
    r = rx.rel(leaf[n].grad.numpy(), d["grad/" + n])
    worst = max(worst, r)
    assert r < 1e-4, (n, r)
  print("tacotron encoder + style tokens: worst gradient rel-L2 vs the reference's code %.2e" % worst)


def test_oracle_reproduces_the_reference_free_running_decode():
  """Tacotron2Decoder._decode in eval mode (decoders/tacotron2_decoder.py:378-428; TacotronHelper,
  parts/tacotron/tacotron_helper.py:138-226) executed from the reference's files: every projected frame goes back
  through the pre-net (dropout on, masks recorded), finished = round(sigmoid(stop logit)) accumulates, the loop runs
  until every sample has finished or 10 x max(src_len) steps; one sample finishes at step 28 of 30 and the others run
  into the limit. oracle/tacotron.py:decoder_infer (the restatement the device's fused decode kernels are tested
  against) must give the same frames, stop logits, alignments (1e-4 over 30 free-running steps), the same step count
  and the same per-sample lengths."""
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_tacotron_infer.npz")))
  out = _oracle_free_running(d, [torch.from_numpy(d["prenet_mask0"]), torch.from_numpy(d["prenet_mask1"])])
  assert int(out["steps"]) == int(d["steps"]) == 30
  assert [int(v) for v in out["lengths"]] == d["lens"].tolist() == [28, 30, 30]
  assert rx.rel(out["mel"].numpy(), d["mel"]) < 1e-4
  assert rx.rel(out["stop"].numpy(), d["stop"][:, :, 0]) < 1e-4
  assert rx.rel(out["align"].numpy(), d["align"]) < 1e-4


def test_oracle_reproduces_the_reference_free_running_decode_at_device_widths():
  """The same at the widths the device's fused decode kernels take (make_ref_exec.py: tacotron_infer_dev — cell and
  memory 64, attention layer 128, pre-net 2 x 64, 16 mel bins, pre-net dropout off, 120 steps, one sample stopping at
  step 41): the oracle gives the same frames / stop logits / alignments (1e-4) and the same lengths, so the device
  test on this fixture (test_ref_exec_tacotron_gpu.py) and the device-vs-oracle tests at full width
  (test_tacotron_infer_gpu.py) are pinned to the reference's own code through the same restatement."""
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_tacotron_infer_dev.npz")))
  out = _oracle_free_running(d, None)
  assert int(out["steps"]) == int(d["steps"]) == 120
  assert [int(v) for v in out["lengths"]] == d["lens"].tolist() == [120, 41, 120, 120]
  assert rx.rel(out["mel"].numpy(), d["mel"]) < 1e-4
  assert rx.rel(out["stop"].numpy(), d["stop"][:, :, 0]) < 1e-4
  assert rx.rel(out["align"].numpy(), d["align"]) < 1e-4


def _oracle_free_running(d, masks):
  B, S, M, H, U, P_, NMEL = [int(v) for v in d["dims"]]
  leaf = {str(n): torch.from_numpy(d["var/" + str(n)].copy()) for n in d["var_names"]}
  k0 = leaf[AW + "multi_rnn_cell/cell_0/lstm_cell/kernel"].t()
  cell = {"w_in": k0[:, :P_], "b0": leaf[AW + "multi_rnn_cell/cell_0/lstm_cell/bias"],
          "wcat": [k0[:, P_:], leaf[AW + "multi_rnn_cell/cell_1/lstm_cell/kernel"].t()],
          "bias": [None, leaf[AW + "multi_rnn_cell/cell_1/lstm_cell/bias"]],
          "wq": leaf[AW + "location_attention/query_layer/kernel"].t(),
          "wmem": leaf[SC + "AttentionMechanism/memory_layer/kernel"][0].t(),
          "v": leaf[AW + "location_attention/attention_v"], "b": leaf[AW + "location_attention/attention_bias"],
          "conv_w": leaf[AW + "location_attention/location_conv/kernel"][:, 0, :],
          "conv_b": leaf[AW + "location_attention/location_conv/bias"],
          "dense_w": leaf[AW + "location_attention/location_dense/kernel"][0]}
  P = {"prenet": [(leaf[SC + "decoder/prenet_%d/kernel" % i].t(), leaf[SC + "decoder/prenet_%d/bias" % i])
                  for i in (1, 2)],
       "cell": cell, "out_w": leaf[SC + "decoder/output_proj/kernel"].t(), "out_b": leaf[SC + "decoder/output_proj/bias"],
       "stop_w": leaf[SC + "decoder/stop_token_proj/kernel"].t(), "stop_b": leaf[SC + "decoder/stop_token_proj/bias"]}
  return otaco.decoder_infer(P, torch.from_numpy(d["enc"]), torch.from_numpy(d["src_len"]), prenet_masks=masks,
                             round_frames_bf16=False)


@pytest.mark.parametrize("case", sorted(rx.gen.T2S_CASES))
def test_oracle_reproduces_the_reference_text2speech_loss(case):
  """Text2SpeechLoss._compute_loss (losses/text2speech_loss.py:35-209) executed from the reference's file on synthetic
  predictions in "both" mode: predictions longer / shorter than the targets (zeros appended to predictions and
  spectrogram, ONES to the stop-token target), masked MSE / L1 with tf.losses' SUM_BY_NONZERO_WEIGHTS reduction,
  masked stop-token cross entropy, the unmasked means, weights and scale. Loss 1e-6, gradients w.r.t. all four
  prediction tensors 1e-5."""
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_t2s_loss.npz")))
  cfg = rx.gen.T2S_CASES[case]
  B, NMEL, NMAG = [int(v) for v in d["dims"]]
  out = {k: torch.from_numpy(d["%s/%s" % (case, k)].copy()).requires_grad_(True) for k in ("mel", "post", "stop", "mag")}
  loss = otaco.text2speech_loss(out, torch.from_numpy(d[case + "/spec"]), torch.from_numpy(d[case + "/stop_target"]),
                                torch.from_numpy(d[case + "/spec_len"]), NMEL, NMAG, l1=cfg.get("l1_norm", False),
                                use_mask=cfg["use_mask"], mel_weight=cfg.get("mel_weight", 1.0),
                                mag_weight=cfg.get("mag_weight", 1.0),
                                stop_token_weight=cfg.get("stop_token_weight", 1.0), scale=cfg.get("scale"))
  assert abs(float(loss.detach()) - float(d[case + "/loss"])) < 1e-6 * max(1.0, abs(float(d[case + "/loss"])))
  loss.backward()
  for k in out:
    assert rx.rel(out[k].grad.numpy(), d["%s/grad/%s" % (case, k)]) < 1e-5, k


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check",
                      "tacotron_decoder", "t2s_loss", "tacotron_infer", "tacotron_infer_dev", "tacotron_encoder"],
                     capture_output=True, text=True, timeout=900)
  assert r.returncode == 0 and r.stdout.count("reproduced") == 5, r.stdout + r.stderr
