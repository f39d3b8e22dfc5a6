"""The Transformer oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_transformer.npz holds what open_seq2seq's TransformerEncoder._encode ->
TransformerDecoder.decode_pass -> PaddedCrossEntropyLossWithSmoothing (the files under /root/reference, run where
they lie by tests/golden/make_ref_exec.py on the TF-primitive stand-in oracle/ref_shim/tf1) computed on a seeded
ragged batch: encoder output, attention bias, logits, loss and the gradient of every trainable variable. The
restatement oracle/transformer.py must reproduce all of it from the same inputs and variables — forward 1e-5,
gradients 1e-4 relative (fp32 on both sides, different summation orders). This pins rows a9-a13 of SURVEY §8a to the
reference's source instead of to a reading of it ("parity pinned modulo the TF primitives restated in tf1").
When the reference checkout is present the generator is re-run and must reproduce the committed fixture.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import transformer as ot  # noqa: E402

sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402


def load_fixture(name="transformer"):
  return rx.load(name)


def oracle_params(d, NL, names=None, arrays=None):
  """reference variable names -> the oracle's parameter dicts (autograd leaves, fp32)."""
  leaves = {}
  if arrays is None:
    arrays = rx.variables(d, names if names is not None else [str(n) for n in d["var_names"]])

  def v(name):
    t = torch.from_numpy(np.array(arrays[name], np.float32)).requires_grad_(True)
    leaves[name] = t
    return t

  def ln(p):
    return {"scale": v(p + "/layer_norm_scale"), "bias": v(p + "/layer_norm_bias")}

  def att(p):
    return {"q": v(p + "/q/kernel"), "k": v(p + "/k/kernel"), "v": v(p + "/v/kernel"),
            "o": v(p + "/output_transform/kernel")}

  def ffn(p):
    return {"w1": v(p + "/filter_layer/kernel"), "b1": v(p + "/filter_layer/bias"),
            "w2": v(p + "/output_layer/kernel"), "b2": v(p + "/output_layer/bias")}
  e, dc = "ForwardPass/transformer_encoder", "ForwardPass/transformer_decoder"
  emb = v(e + "/embedding_shared_weights/embedding_and_softmax/weights")
  PE = {"emb": emb, "layers": [], "ln_out": ln(e + "/layer_normalization")}
  PD = {"emb": emb, "layers": [], "ln_out": ln(dc + "/layer_normalization")}
  for n in range(NL):
    ls = "%s/layer_%d" % (e, n)
    PE["layers"].append({"ln1": ln(ls + "/self_attention/layer_normalization"),
                         "att": att(ls + "/self_attention/self_attention"),
                         "ln2": ln(ls + "/ffn/layer_normalization"), "ffn": ffn(ls + "/ffn/feed_foward_network")})
    ls = "%s/layer_%d" % (dc, n)
    PD["layers"].append({"ln1": ln(ls + "/self_attention/layer_normalization"),
                         "self": att(ls + "/self_attention/self_attention"),
                         "ln2": ln(ls + "/encdec_attention/layer_normalization"),
                         "cross": att(ls + "/encdec_attention/attention"),
                         "ln3": ln(ls + "/ffn/layer_normalization"), "ffn": ffn(ls + "/ffn/feed_foward_network")})
  return PE, PD, leaves


def rel(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("fixture", ["transformer", "transformer_d512"])
def test_oracle_reproduces_the_reference_transformer(fixture):
  d, names = load_fixture(fixture)
  B, S, T, V, D, H, F, NL = [int(v) for v in d["config"]]
  PE, PD, leaves = oracle_params(d, NL)
  assert sorted(leaves) == sorted(names), "every reference variable is consumed by the oracle, and nothing else"
  src, tgt = torch.from_numpy(d["src"]).long(), torch.from_numpy(d["tgt"]).long()
  enc_out, bias = ot.encoder(src, PE, H)
  logits = ot.decoder_pass(tgt, enc_out, bias, PD, H)
  loss = ot.padded_xent_smoothing(logits, tgt, float(d["label_smoothing"]))
  loss.backward()
  assert np.array_equal(bias.numpy(), d["enc_bias"])
  # rows at padded source positions differ BY DESIGN: the reference's FFN gathers the non-pad rows and scatters
  # zeros back for the others (ffn_layer.py:56-85, remove_padding), the oracle (and the packed device layout)
  # never materialises them; no later op reads them (the padding bias masks them as keys)
  live = (d["src"] != 0) & (d["src"] < V)
  assert rel(enc_out.detach().numpy()[live], d["enc_out"][live]) < 1e-5
  assert rel(logits.detach().numpy(), d["logits"]) < 1e-5
  assert abs(float(loss.detach()) - float(d["loss"])) < 1e-5 * abs(float(d["loss"]))
  worst = 0.0
  for n in names:
    worst = max(worst, rx.check_gradient(d, n, leaves[n].grad.numpy(), 1e-4))
  print("worst gradient rel-L2 vs the reference's code: %.2e" % worst)


def test_oracle_reproduces_the_reference_cached_beam_decode():
  """TransformerDecoder.predict in infer mode (decoders/transformer_decoder.py:232-326) executed from the reference's
  files: the cached decode step (embedding of the last id + timing-signal row i, the self-attention bias slice, K / V
  caches concatenated inside the reference's Attention layers and gathered per surviving beam) under
  sequence_beam_search's while loop. The oracle composition the device's beam-search test uses —
  oracle/beam_search.py driving oracle/transformer.py's decoder_pass on the growing prefix (no cache: the cache is an
  optimisation, the logits are the same function of the prefix) — must return the same top-beam ids, exactly."""
  from oracle import beam_search as obs
  d, names = rx.load("transformer_infer")
  B, S, V, D, H, F, NL, beam, extra = [int(v) for v in d["config"]]
  PE, PD, leaves = oracle_params(d, NL, names)
  with torch.no_grad():
    src = torch.from_numpy(d["src"]).long()
    enc_out, bias = ot.encoder(src, PE, H)

    def fn(ids, i, cache):
      tgt = torch.from_numpy(np.concatenate([ids[:, 1:], np.zeros((ids.shape[0], 1), ids.dtype)], 1)).long()
      logits = ot.decoder_pass(tgt, torch.from_numpy(cache["enc"]), torch.from_numpy(cache["bias"]), PD, H)
      return logits[:, i, :].numpy(), cache
    ids, scores = obs.sequence_beam_search(fn, np.zeros(B, np.int32), {"enc": enc_out.numpy(), "bias": bias.numpy()},
                                           V, beam, 0.6, S + extra, 1)
  top = ids[:, 0, 1:]
  assert top.shape == d["ids"].shape, (top.shape, d["ids"].shape)
  assert np.array_equal(top, d["ids"]), (top, d["ids"])
  assert len({tuple(r) for r in d["ids"].tolist()}) == B, "three different hypotheses"
  # predict() ends with decode_pass(top ids) (transformer_decoder.py:318-320): the logits it returns
  with torch.no_grad():
    lg = ot.decoder_pass(torch.from_numpy(d["ids"]).long(), enc_out, bias, PD, H)
  assert rel(lg.numpy(), d["logits"]) < 1e-5


def beam_arrays(d):
  C = rx.gen.TRANSFORMER_BEAM
  return {str(n): rx.gen.transformer_beam_variable(str(n), tuple(int(v) for v in d["shape/" + str(n)]), C["seed"])
          for n in d["var_names"]}


def test_oracle_reproduces_the_reference_beam_decode_at_device_widths():
  """The same at d_model 512, 8 heads, V 96, beam 4 (tests/golden/ref_exec_transformer_infer_d512.npz: the fixture
  the HIP beam search is held to, tests/test_ref_exec_transformer_gpu.py): a row that ends with EOS and is zero-padded,
  rows that run to the length cap; `stable` marks the rows whose winner the REFERENCE keeps under six perturbations
  of every matrix by 2^-7 relative."""
  from oracle import beam_search as obs
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_transformer_infer_d512.npz")))
  C = rx.gen.TRANSFORMER_BEAM
  B, S, V, D, H, F, NL = C["dims"]
  names = [str(n) for n in d["var_names"]]
  PE, PD, leaves = oracle_params(d, NL, names, arrays=beam_arrays(d))
  assert sorted(leaves) == sorted(names)
  with torch.no_grad():
    src = torch.from_numpy(d["src"]).long()
    enc_out, bias = ot.encoder(src, PE, H)

    def fn(ids, i, cache):
      tgt = torch.from_numpy(np.concatenate([ids[:, 1:], np.zeros((ids.shape[0], 1), ids.dtype)], 1)).long()
      logits = ot.decoder_pass(tgt, torch.from_numpy(cache["enc"]), torch.from_numpy(cache["bias"]), PD, H)
      return logits[:, i, :].numpy(), cache
    ids, scores = obs.sequence_beam_search(fn, np.zeros(B, np.int32), {"enc": enc_out.numpy(), "bias": bias.numpy()},
                                           V, C["beam"], 0.6, S + C["extra"], 1)
  assert np.array_equal(ids[:, 0, 1:], d["ids"]), (ids[:, 0, 1:], d["ids"])
  assert d["stable"].any() and not d["stable"].all()
  ended = [(r == 1).any() for r in d["ids"]]
  assert any(ended) and not all(ended), "a row that finishes (zero-padded after EOS) and rows that run to the cap"


def test_fixture_has_the_cases_that_matter():
  d, _ = load_fixture()
  B, S, T, V, D, H, F, NL = [int(v) for v in d["config"]]
  assert (d["src"] >= V).any(), "an id past the vocabulary (mapped to the pad symbol, embedding_layer.py:71-73)"
  assert (d["src"] == 0).any() and (d["tgt"] == 0).any(), "ragged batch: pad positions on both sides"
  assert d["var/ForwardPass/transformer_encoder/embedding_shared_weights/embedding_and_softmax/weights"].shape[0] \
      % 8 == 0, "pad_embeddings_2_eight"
  sc = d["var/ForwardPass/transformer_encoder/layer_normalization/layer_norm_scale"]
  assert np.abs(sc - 1).max() > 1e-2, "LayerNorm parameters away from their 1 / 0 initial values"


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "transformer",
                      "transformer_d512", "transformer_infer", "transformer_infer_d512"], capture_output=True,
                     text=True, timeout=900)
  assert r.returncode == 0 and r.stdout.count("reproduced") == 4, r.stdout + r.stderr
