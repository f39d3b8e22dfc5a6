"""Fixture generator: runs the REFERENCE'S OWN PYTHON SOURCE (the files under /root/reference/open_seq2seq, where
they lie) on seeded inputs and writes inputs, variables and outputs to tests/golden/ref_exec_*.npz.

TensorFlow is not installed in this container, so `tensorflow` resolves to oracle/ref_shim/tf1 — a restatement
of the TF 1.x LIBRARY primitives on torch CPU tensors (see its header). Everything above the primitives — the
encoder / decoder / loss / lr-policy / loss-scaler logic — executes from the reference's files, unmodified.
The parity tests (tests/test_ref_exec_*.py) then hold the oracle restatements (CPU) and the HIP path (GPU)
against these fixtures; when /root/reference is present they also re-run this generator and check that it
reproduces the committed files.

    python tests/golden/make_ref_exec.py [--check] [name ...]

Only importable pieces of the reference are loaded: its packages' __init__ files import every model family
(cuDNN RNNs, librosa, horovod ...), so the package names are pre-seeded as empty namespace stubs that point at
the reference's directories and the wanted modules are imported one by one.
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OS2S_REFERENCE", "/root/reference")
PKG = os.path.join(REF, "open_seq2seq")


def reference_available():
  return os.path.isdir(PKG)


def _install():
  """tensorflow -> the shim; open_seq2seq.* -> namespace stubs over the reference's directories."""
  sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
  import tf1
  tf = tf1.install()
  for k in [k for k in sys.modules if k == "open_seq2seq" or k.startswith("open_seq2seq.")]:
    del sys.modules[k]
  subs = ["", "encoders", "decoders", "losses", "optimizers", "utils", "parts", "parts.transformer", "parts.cnns",
          "parts.rnns", "data", "data.speech2text", "models"]
  for s in subs:
    name = "open_seq2seq" + ("." + s if s else "")
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(PKG, *s.split("."))] if s else [PKG]
    m.__package__ = name
    sys.modules[name] = m
    if s:
      parent, _, leaf = name.rpartition(".")
      setattr(sys.modules[parent], leaf, m)
  # the data layer class the TDNN encoder imports for a type check only (its module needs librosa / pandas)
  dl = types.ModuleType("open_seq2seq.data.speech2text.speech2text")
  dl.Speech2TextDataLayer = type("Speech2TextDataLayer", (), {})
  sys.modules[dl.__name__] = dl
  for name, attrs in (("open_seq2seq.parts.rnns.weight_drop", ["WeightDropLayerNormBasicLSTMCell"]),
                      ("open_seq2seq.parts.rnns.slstm", ["BasicSLSTMCell"]),
                      ("open_seq2seq.parts.rnns.glstm", ["GLSTMCell"]),
                      ("open_seq2seq.parts.rnns.zoneout", ["ZoneoutWrapper"])):
    # imported by rnn_decoders.py / parts/rnns/utils.py at module level, not used by the configurations executed
    # here; their own imports reach deep into TensorFlow's private modules
    m = types.ModuleType(name)
    for a in attrs:
      setattr(m, a, type(a, (), {}))
    sys.modules[name] = m
  import collections
  import collections.abc
  if not hasattr(collections, "Sequence"):      # the reference predates Python 3.10 (optimizers.py:441)
    collections.Sequence = collections.abc.Sequence
  if "Inf" not in np.__dict__:                  # ... and NumPy 2.0 (rnn_beam_search_decoder.py:324)
    np.Inf = np.inf
  import importlib
  enc = importlib.import_module("open_seq2seq.encoders.encoder")
  sys.modules["open_seq2seq.encoders"].Encoder = enc.Encoder
  dec = importlib.import_module("open_seq2seq.decoders.decoder")
  sys.modules["open_seq2seq.decoders"].Decoder = dec.Decoder
  los = importlib.import_module("open_seq2seq.losses.loss")
  sys.modules["open_seq2seq.losses"].Loss = los.Loss
  return tf, importlib.import_module


def _np(v):
  return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


# ---------------------------------------------------------------------------------------------------------
# Transformer: TransformerEncoder._encode -> TransformerDecoder.decode_pass -> PaddedCrossEntropyLossWithSmoothing
# (encoders/transformer_encoder.py:78-170, decoders/transformer_decoder.py:96-230, parts/transformer/*.py,
# losses/sequence_loss.py:233-309), train mode with every dropout probability 0, forward + all gradients.
# ---------------------------------------------------------------------------------------------------------
def seeded_array(name, shape, seed, scale=None):
  """The value of variable `name` in the fixtures that do not store their variables: N(0, scale^2) from a seed
  derived from the name (scale: 1 / sqrt(fan_in) for matrices, 0.1 around 1 / 0 for LayerNorm scale / bias and
  for biases) — the generator and the tests call this same function."""
  import zlib
  rs = np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 31))
  x = rs.standard_normal(shape).astype(np.float32)
  if len(shape) >= 2:
    return x * np.float32(scale if scale is not None else 1.0 / np.sqrt(shape[-2] if "embedding" not in name else shape[-1]))
  if name.endswith("attention_v"):        # scores of a few units at depth 128: attention that is neither uniform nor
    return np.float32(0.2) * x            # one-hot (a one-hot softmax flips under a bf16 ulp: nothing to compare)
  return (np.float32(1.0) if name.endswith(("scale", "gamma", "attention_g")) else np.float32(0.0)) + np.float32(0.1) * x


def projection(name, g, seed):
  """(norm, <g, r>) with r a seeded N(0,1) direction: two numbers that pin a gradient tensor without storing it."""
  import zlib
  r = np.random.RandomState((zlib.crc32(("proj/" + name).encode()) + seed) % (2 ** 31)).standard_normal(g.shape)
  g = np.asarray(g, np.float64)
  return np.array([np.linalg.norm(g), float((g * r).sum())], np.float64)


def transformer(seed=11, dims=(3, 11, 9, 45, 32, 4, 64, 2), store_vars=True):
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  TransformerEncoder = imp("open_seq2seq.encoders.transformer_encoder").TransformerEncoder
  TransformerDecoder = imp("open_seq2seq.decoders.transformer_decoder").TransformerDecoder
  Loss = imp("open_seq2seq.losses.sequence_loss").PaddedCrossEntropyLossWithSmoothing
  rng = np.random.RandomState(seed)
  B, S, T, V, D, H, F, NL = dims
  src_len = np.array([11, 7, 4], np.int32)
  tgt_len = np.array([6, 9, 3], np.int32)
  src = np.zeros((B, S), np.int32)
  tgt = np.zeros((B, T), np.int32)
  for b in range(B):
    src[b, :src_len[b]] = rng.randint(2, V, size=src_len[b])
    tgt[b, :tgt_len[b]] = rng.randint(2, V, size=tgt_len[b])
  if store_vars:
    src[1, 2] = V + 5            # an id past the vocabulary: mapped to the pad symbol (embedding_layer.py:71-73)
  enc_params = dict(encoder_layers=NL, hidden_size=D, num_heads=H, attention_dropout=0.0, filter_size=F,
                    src_vocab_size=V, relu_dropout=0.0, layer_postprocess_dropout=0.0, remove_padding=True,
                    pad_embeddings_2_eight=True, dtype=tf.float32)
  dec_params = dict(EOS_ID=1, layer_postprocess_dropout=0.0, num_hidden_layers=NL, hidden_size=D, num_heads=H,
                    attention_dropout=0.0, relu_dropout=0.0, filter_size=F, batch_size=B, tgt_vocab_size=V,
                    beam_size=4, alpha=0.6, extra_decode_length=5, GO_SYMBOL=1, PAD_SYMBOL=0, END_SYMBOL=1,
                    dtype=tf.float32)
  loss_params = dict(batch_size=B, tgt_vocab_size=V, label_smoothing=0.1, pad_embeddings_2_eight=True,
                     dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    encoder = TransformerEncoder(enc_params, None, mode="train")
    decoder = TransformerDecoder(dec_params, None, mode="train")
    loss_fn = Loss(loss_params, None)
    src_t, src_len_t = tf.constant(src), tf.constant(src_len)
    tgt_t, tgt_len_t = tf.constant(tgt), tf.constant(tgt_len)
    enc_out = encoder.encode({"source_tensors": [src_t, src_len_t]})
    dec_out = decoder.decode({"encoder_output": enc_out, "target_tensors": [tgt_t, tgt_len_t]})
    loss = loss_fn.compute_loss({"decoder_output": dec_out, "target_tensors": [tgt_t, tgt_len_t]})
  tvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in tvars]
  with tf.Session() as sess:
    for n, v in zip(names, tvars):
      if not store_vars:
        v.load(seeded_array(n, tuple(v._var.shape), seed))
      elif "layer_norm" in v.name:      # LayerNorm parameters away from 1 / 0 (a swapped scale / bias would hide there)
        v.load(_np(v._var) + 0.1 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
    grads = tf.gradients(loss, tvars)
    vals = sess.run({"enc": enc_out["outputs"], "bias": enc_out["inputs_attention_bias"], "logits": dec_out["logits"],
                     "loss": loss, "grads": grads, "vars": list(tvars)})
  out = {"src": src, "src_len": src_len, "tgt": tgt, "tgt_len": tgt_len, "enc_out": vals["enc"],
         "enc_bias": vals["bias"], "logits": vals["logits"], "loss": np.float32(vals["loss"]),
         "config": np.array([B, S, T, V, D, H, F, NL], np.int32), "label_smoothing": np.float32(0.1),
         "var_names": np.array(names), "seed": np.int32(seed)}
  for n, v, g in zip(names, vals["vars"], vals["grads"]):
    if store_vars:
      out["var/" + n] = v.astype(np.float32)
      out["grad/" + n] = g.astype(np.float32)
    else:
      out["shape/" + n] = np.array(v.shape, np.int32)
      out["gproj/" + n] = projection(n, g, seed)
  return out


def transformer_infer(seed=13, dims=(3, 8, 40, 32, 4, 64, 2), beam=3, extra=4):
  """TransformerEncoder + TransformerDecoder.predict in infer mode (decoders/transformer_decoder.py:232-326): the
  cached decode step (_get_symbols_to_logits_fn: embedding of the last id, timing signal row i, self-attention bias
  slice, K / V cache concatenation inside the reference's Attention layers) under sequence_beam_search's tf.while_loop;
  returns the top beam's ids. Variables are stored (no gradients)."""
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  TransformerEncoder = imp("open_seq2seq.encoders.transformer_encoder").TransformerEncoder
  TransformerDecoder = imp("open_seq2seq.decoders.transformer_decoder").TransformerDecoder
  rng = np.random.RandomState(seed)
  B, S, V, D, H, F, NL = dims
  src_len = np.array([8, 5, 3], np.int32)
  src = np.zeros((B, S), np.int32)
  for b in range(B):
    src[b, :src_len[b]] = rng.randint(2, V, size=src_len[b])
  enc_params = dict(encoder_layers=NL, hidden_size=D, num_heads=H, attention_dropout=0.1, filter_size=F,
                    src_vocab_size=V, relu_dropout=0.1, layer_postprocess_dropout=0.1, remove_padding=True,
                    dtype=tf.float32)
  dec_params = dict(EOS_ID=1, layer_postprocess_dropout=0.1, num_hidden_layers=NL, hidden_size=D, num_heads=H,
                    attention_dropout=0.1, relu_dropout=0.1, filter_size=F, batch_size=B, tgt_vocab_size=V,
                    beam_size=beam, alpha=0.6, extra_decode_length=extra, GO_SYMBOL=1, PAD_SYMBOL=0, END_SYMBOL=1,
                    dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    encoder = TransformerEncoder(enc_params, None, mode="infer")
    decoder = TransformerDecoder(dec_params, None, mode="infer")
    enc_out = encoder.encode({"source_tensors": [tf.constant(src), tf.constant(src_len)]})
    dec_out = decoder.decode({"encoder_output": enc_out})
  gvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in gvars]
  with tf.Session() as sess:
    for n, v in zip(names, gvars):
      if "layer_norm" in n:
        v.load(_np(v._var) + 0.1 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
      elif n.endswith("embedding_and_softmax/weights"):
        v.load(3.0 * _np(v._var))          # a sharper output distribution: well separated beams
    vals = sess.run({"ids": dec_out["outputs"][0], "logits": dec_out["logits"], "vars": list(gvars)})
  out = {"src": src, "src_len": src_len, "ids": vals["ids"].astype(np.int32), "logits": vals["logits"],
         "config": np.array(list(dims) + [beam, extra], np.int32), "var_names": np.array(names)}
  for n, v in zip(names, vals["vars"]):
    out["var/" + n] = v.astype(np.float32)
  return out


TRANSFORMER_BEAM = dict(dims=(4, 11, 96, 512, 8, 1024, 2), beam=4, extra=4, seed=29, emb_gain=1.0, mat_gain=3.0, eos_gain=2.0, perturbations=6,
                        src_len=[11, 7, 9, 4])


def transformer_beam_variable(name, shape, seed, perturbation=None):
  """The variable values of the d512 beam-search fixture: seeded_array, the shared embedding (= output layer) times
  emb_gain for a sharper output distribution; perturbation k: every matrix entry times 1 + 2^-7 u, u ~ U(-1, 1)."""
  import zlib
  a = seeded_array(name, shape, seed)
  if name.endswith("embedding_and_softmax/weights"):
    a = a * np.float32(TRANSFORMER_BEAM["emb_gain"])
    a[1] *= np.float32(TRANSFORMER_BEAM["eos_gain"])        # EOS: logits of a wider spread, sometimes the largest
  elif a.ndim == 2:
    a = a * np.float32(TRANSFORMER_BEAM["mat_gain"])
  if a.ndim == 2 and perturbation is not None:
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 1000 * perturbation) % (2 ** 31))
    a = a * (1 + np.float32(2.0 ** -7) * rs.uniform(-1, 1, a.shape).astype(np.float32))
  return a


def transformer_infer_d512():
  """TransformerDecoder.predict (cached decode under sequence_beam_search) at the widths the HIP kernels take: the
  device's beam search is held to these ids. As for the RNN beam search (nmt_beam), the reference is re-run with every
  matrix perturbed by 2^-7 relative and the rows whose winner never changes are recorded (`stable`)."""
  C = TRANSFORMER_BEAM
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(C["seed"])
  TransformerEncoder = imp("open_seq2seq.encoders.transformer_encoder").TransformerEncoder
  TransformerDecoder = imp("open_seq2seq.decoders.transformer_decoder").TransformerDecoder
  rng = np.random.RandomState(C["seed"])
  B, S, V, D, H, F, NL = C["dims"]
  src_len = np.array(C["src_len"], np.int32)
  src = np.zeros((B, S), np.int32)
  for b in range(B):
    src[b, :src_len[b]] = rng.randint(2, V, size=src_len[b])
  enc_params = dict(encoder_layers=NL, hidden_size=D, num_heads=H, attention_dropout=0.1, filter_size=F,
                    src_vocab_size=V, relu_dropout=0.1, layer_postprocess_dropout=0.1, remove_padding=True,
                    dtype=tf.float32)          # V = 96: a vocabulary the data layer already padded to a multiple of 8
  dec_params = dict(EOS_ID=1, layer_postprocess_dropout=0.1, num_hidden_layers=NL, hidden_size=D, num_heads=H,
                    attention_dropout=0.1, relu_dropout=0.1, filter_size=F, batch_size=B, tgt_vocab_size=V,
                    beam_size=C["beam"], alpha=0.6, extra_decode_length=C["extra"], GO_SYMBOL=1, PAD_SYMBOL=0,
                    END_SYMBOL=1, dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    encoder = TransformerEncoder(enc_params, None, mode="infer")
    decoder = TransformerDecoder(dec_params, None, mode="infer")
    enc_out = encoder.encode({"source_tensors": [tf.constant(src), tf.constant(src_len)]})
    dec_out = decoder.decode({"encoder_output": enc_out})
  gvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in gvars]
  with tf.Session() as sess:
    for n, v in zip(names, gvars):
      v.load(transformer_beam_variable(n, tuple(v._var.shape), C["seed"]))
    ids = sess.run(dec_out["outputs"][0]).astype(np.int32)
    stable = np.ones(B, np.bool_)
    for k in range(C["perturbations"]):
      for n, v in zip(names, gvars):
        v.load(transformer_beam_variable(n, tuple(v._var.shape), C["seed"], perturbation=k))
      ids_k = sess.run(dec_out["outputs"][0])
      stable &= np.array([ids_k.shape == ids.shape and np.array_equal(ids_k[b], ids[b]) for b in range(B)])
  out = {"src": src, "src_len": src_len, "ids": ids, "stable": stable, "var_names": np.array(names)}
  for n, v in zip(names, gvars):
    out["shape/" + n] = np.array(tuple(v._var.shape), np.int32)
  return out


def transformer_d512():
  """The same graph at the narrowest widths the HIP kernels take (head dim 64, LayerNorm rows of 512 or 1024):
  d_model 512, 8 heads, filter 1024, 2 + 2 layers, V 90 -> 96. Variables come from seeded_array (not stored),
  gradients are stored as (norm, seeded projection) per variable."""
  return transformer(seed=23, dims=(3, 11, 9, 90, 512, 8, 1024, 2), store_vars=False)


# ---------------------------------------------------------------------------------------------------------
# Jasper / TDNN: TDNNEncoder._encode (encoders/tdnn_encoder.py:87-265) over conv_bn_actv / conv_bn_res_bn_actv
# (parts/cnns/conv_blocks.py:61-232) -> FullyConnectedCTCDecoder (decoders/fc_decoders.py:105-250: dense +
# time-major transpose + tf.nn.ctc_greedy_decoder), train mode (BatchNorm on batch statistics, moving averages
# updated through UPDATE_OPS), dropout keep 1. tf.nn.ctc_loss is TensorFlow-internal: the gradients are those of
# the surrogate loss sum(logits * R) with a seeded R (also stored), which exercises every variable.
# ---------------------------------------------------------------------------------------------------------
def jasper_layers(c, k):
  """Jasper's layer pattern scaled down: stride-2 first layer, dense-residual blocks with repeat 2, a
  dilation-2 layer, a 1x1 last layer (example_configs/speech2text/jasper10x5_LibriSpeech_nvgrad_masks.py:58-147)."""
  def blk(rep, kk, ch, stride=1, dil=1, res=False):
    d = {"type": "conv1d", "repeat": rep, "kernel_size": [kk], "stride": [stride], "num_channels": ch,
         "padding": "SAME", "dilation": [dil], "dropout_keep_prob": 1.0}
    if res:
      d.update(residual=True, residual_dense=True)
    return d
  return [blk(1, k[0], c[0], stride=2), blk(2, k[0], c[0], res=True), blk(2, k[1], c[1], res=True),
          blk(2, k[2], c[2], res=True), blk(1, k[3], c[3], dil=2), blk(1, 1, c[4])]


def tdnn_input(seed, src_len, T, F):
  """The feature batch of the TDNN fixtures (zeros past each length, as the data layer pads)."""
  x = np.random.RandomState(seed + 1000).standard_normal((len(src_len), T, F)).astype(np.float32)
  for b, n in enumerate(src_len):
    x[b, int(n):] = 0.0
  return x


def tdnn(seed=5, F=16, chans=(8, 16, 16, 24, 24), kern=(5, 7, 9, 11), T=48, B=3, store_vars=True, V=29):
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  TDNNEncoder = imp("open_seq2seq.encoders.tdnn_encoder").TDNNEncoder
  Decoder = imp("open_seq2seq.decoders.fc_decoders").FullyConnectedCTCDecoder
  CTCLoss = imp("open_seq2seq.losses.ctc_loss").CTCLoss
  DataLayer = sys.modules["open_seq2seq.data.speech2text.speech2text"].Speech2TextDataLayer
  rng = np.random.RandomState(seed)
  src_len = np.array(([T, T - 13, T // 2 - 3] * B)[:B], np.int32)
  src_len[3:] -= np.arange(B - 3, dtype=np.int32)[:max(B - 3, 0)] * 2 + 1
  x = tdnn_input(seed, src_len, T, F)
  layers = jasper_layers(chans, kern)

  class _Model(object):                 # what the encoder asks its model for (encoder.py:86-112, tdnn_encoder.py:113-117)
    params = {"dtype": tf.float32}

    def get_data_layer(self):
      dl = DataLayer()
      dl.params = {"backend": "librosa", "pad_to": 16}
      return dl
  enc_params = dict(convnet_layers=layers, dropout_keep_prob=1.0, activation_fn=tf.nn.relu, data_format="channels_last",
                    use_conv_mask=True, dtype=tf.float32)
  dec_params = dict(tgt_vocab_size=V, dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    encoder = TDNNEncoder(enc_params, _Model(), name="w2l_encoder", mode="train")
    decoder = Decoder(dec_params, None, mode="train")
    x_t, len_t = tf.constant(x), tf.constant(src_len)
    enc_out = encoder.encode({"source_tensors": [x_t, len_t]})
    dec_out = decoder.decode({"encoder_output": enc_out})
  logits = dec_out["logits"]                      # [T', B, V], time major
  # CTCLoss (losses/ctc_loss.py:44-88: dense_to_sparse, tf.nn.ctc_loss(ignore_longer_outputs_than_inputs=True),
  # mask_nans, mean over the WHOLE batch); the last sample's transcript is longer than its output: zero loss
  out_frames = (src_len + 1) // 2
  label_len = np.minimum(np.maximum(out_frames // 3, 2), 12).astype(np.int32)
  label_len[-1] = out_frames[-1] + 3
  labels = rng.randint(0, V - 1, size=(B, int(label_len.max()))).astype(np.int32)
  with tf.variable_scope("ForwardPass"):
    ctc = CTCLoss({}, None).compute_loss({"decoder_output": dec_out,
                                          "target_tensors": [tf.constant(labels), tf.constant(label_len)]})
  ctc_dlogits = tf.gradients(ctc, [logits])[0]
  R = rng.standard_normal(tuple(int(v) for v in logits.get_shape())).astype(np.float32)
  loss = tf.reduce_sum(logits * tf.constant(R))
  tvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in tvars]
  moving = [v for v in tf.global_variables() if "moving_" in v.name]
  with tf.Session() as sess:
    for n, v in zip(names, tvars):
      v.load(seeded_array(n, tuple(v._var.shape), seed) if not store_vars else
             (seeded_array(n, tuple(v._var.shape), seed) if v._var.dim() == 1 else _np(v._var)))
    grads = tf.gradients(loss, tvars)
    decoded = dec_out["outputs"][0]
    vals = sess.run({"enc": enc_out["outputs"], "len": enc_out["src_length"], "logits": logits, "loss": loss,
                     "grads": grads, "vars": list(tvars), "ids": tf.sparse_tensor_to_dense(decoded, default_value=-1),
                     "ctc": ctc, "ctc_dlogits": ctc_dlogits})
    sess.run(tf.get_collection(tf.GraphKeys.UPDATE_OPS))
    mv = sess.run(list(moving))
  out = {"src_len": src_len, "T": np.int32(T), "out_len": vals["len"].astype(np.int32),
         "logits": vals["logits"], "R": R, "loss": np.float32(vals["loss"]), "greedy_ids": vals["ids"].astype(np.int32),
         "var_names": np.array(names), "seed": np.int32(seed), "chans": np.array(chans, np.int32),
         "kern": np.array(kern, np.int32), "F": np.int32(F), "V": np.int32(V),
         "moving_names": np.array([v.name.split(":")[0] for v in moving]), "labels": labels, "label_len": label_len,
         "ctc_loss": np.float32(vals["ctc"]), "ctc_dlogits": vals["ctc_dlogits"].astype(np.float32)}
  for v, a in zip(moving, mv):
    out["moving/" + v.name.split(":")[0]] = a.astype(np.float32)
  if store_vars:
    out["enc_out"] = vals["enc"]
  for n, v, g in zip(names, vals["vars"], vals["grads"]):
    if store_vars:
      out["var/" + n] = v.astype(np.float32)
      out["grad/" + n] = g.astype(np.float32)
    else:
      out["shape/" + n] = np.array(v.shape, np.int32)
      out["gproj/" + n] = projection(n, g, seed)
  return out


def tdnn_wide():
  """The same graph at widths the HIP convolution kernels run at (channels in multiples of 64, 64 features):
  variables from seeded_array, gradients as (norm, projection); B = 6 (288 rows per BatchNorm: the batch size of the
  device's own end-to-end test); the encoder output is not stored (the logits are its 29-dimensional image)."""
  return tdnn(seed=9, F=64, chans=(128, 192, 256, 320, 384), kern=(11, 13, 17, 29), T=96, B=6, store_vars=False)


# ---------------------------------------------------------------------------------------------------------
# Optimizer side: the seven learning-rate policies (optimizers/lr_policies.py) evaluated at a list of global steps,
# and the two loss scalers (optimizers/automatic_loss_scaler.py: BackoffScaler, LogMaxScaler) driven through a
# scripted sequence of (has_nan, amax) events — update_op built ONCE and run per event, as a training loop does.
# ---------------------------------------------------------------------------------------------------------
LR_CASES = [
    ("fixed_lr", dict(learning_rate=0.3)),
    ("piecewise_constant", dict(learning_rate=0.1, boundaries=[40, 90, 200], decay_rates=[0.5, 0.1, 0.01])),
    ("piecewise_constant", dict(learning_rate=0.1, boundaries=[2, 5], decay_rates=[0.1, 0.01], steps_per_epoch=30)),
    ("exp_decay", dict(learning_rate=0.05, decay_steps=50, decay_rate=0.5, use_staircase_decay=True,
                       begin_decay_at=20, min_lr=1e-3)),
    ("exp_decay", dict(learning_rate=0.05, decay_steps=37, decay_rate=0.9, use_staircase_decay=False)),
    ("poly_decay", dict(learning_rate=0.02, decay_steps=300, power=2.0, min_lr=1e-5)),
    ("poly_decay", dict(learning_rate=0.02, decay_steps=200, power=0.5, begin_decay_at=60, min_lr=1e-4,
                        warmup_steps=25)),
    ("cosine_decay", dict(learning_rate=0.01, decay_steps=250, begin_decay_at=30, min_lr=0.05, warmup_steps=10)),
    ("transformer_policy", dict(learning_rate=2.0, d_model=512, warmup_steps=80)),
    ("transformer_policy", dict(learning_rate=1.0, d_model=1024, warmup_steps=40, max_lr=1e-3, coefficient=2.0)),
    ("inv_poly_decay", dict(learning_rate=0.1, decay_steps=400, min_lr=1e-4, power=2.0)),
]
LR_STEPS = [0, 1, 2, 9, 10, 11, 19, 20, 24, 25, 26, 29, 30, 31, 39, 40, 41, 59, 60, 61, 79, 80, 81, 89, 90, 91, 149, 150,
            151, 199, 200, 201, 260, 299, 300, 301, 399, 400, 1000]


def scaler_events(seed=3, n=260):
  """(has_nan, amax) per step: mostly finite maxima spread over decades, isolated and back-to-back overflows of both
  kinds (NaN seen, Inf maximum), small step windows so that growth happens inside the trace."""
  rs = np.random.RandomState(seed)
  amax = np.exp2(rs.uniform(-6, 14, size=n)).astype(np.float32)
  nan = np.zeros(n, np.bool_)
  for i in (7, 8, 40, 41, 42, 100, 180):
    nan[i] = True
  for i in (20, 101, 150, 151):
    amax[i] = np.inf
  return nan, amax


def optim():
  tf, imp = _install()
  tf.reset_default_graph()
  pol = imp("open_seq2seq.optimizers.lr_policies")
  als = imp("open_seq2seq.optimizers.automatic_loss_scaler")
  out = {"lr_steps": np.array(LR_STEPS, np.int64)}
  gs = tf.train.get_or_create_global_step()
  with tf.Session() as sess:
    for i, (name, params) in enumerate(LR_CASES):
      lr = getattr(pol, name)(global_step=gs, **params)
      vals = []
      for st in LR_STEPS:
        gs.load(np.int64(st))
        vals.append(float(sess.run(lr)) if hasattr(lr, "_eval") else float(lr))
      out["lr/%d/%s" % (i, name)] = np.array(vals, np.float64)
    nan, amax = scaler_events()
    out["ev_nan"], out["ev_amax"] = nan, amax
    cases = [("backoff", dict(step_window=16)), ("backoff", dict(scale_min=8.0, scale_max=4096.0, step_factor=4.0,
                                                              step_window=5)),
             ("logmax", {}), ("logmax", dict(scale_max=2.0 ** 10, beta1=0.9, beta2=0.95, overflow_std_dev=2.0))]
    for i, (algo, params) in enumerate(cases):
      sc = als.AutomaticLossScaler(algorithm=algo, params=dict(params))
      p_nan, p_amax = tf.placeholder(tf.bool, []), tf.placeholder(tf.float32, [])
      up = sc.update_op(p_nan, p_amax)
      trace = [float(sess.run(sc.loss_scale))]
      for h, a in zip(nan, amax):
        sess.run(up, {p_nan: bool(h), p_amax: np.float32(a)})
        trace.append(float(sess.run(sc.loss_scale)))
      out["scale/%d/%s" % (i, algo)] = np.array(trace, np.float32)
      out["scale_params/%d" % i] = np.array(repr(sorted(params.items())))
  return out


# ---------------------------------------------------------------------------------------------------------
# The train op: optimizers.optimize_loss (optimizers/optimizers.py:107-286) with dtype "mixed" — the
# MixedPrecisionOptimizerWrapper (mp_wrapper.py: loss scaling, FP32 master copies, regularisation on the master
# copy, un-scaling, skip on NaN / Inf, saturate_cast back to fp16), post_process_gradients (LARC / global-norm
# clipping), the automatic loss scalers and the optimizer itself (NovoGrad from optimizers/novograd.py over the
# restated tf.train.MomentumOptimizer; Adam; Momentum), built ONCE and run step after step on a toy model whose
# fp16 gradients overflow at the initial loss scale. Recorded per step: the loss scale going in, the learning
# rate, the scaled half-precision gradients the wrapper received, and every variable after the step.
# ---------------------------------------------------------------------------------------------------------
TRAIN_OP_CASES = {
    "novograd_larc_backoff": dict(optimizer="NovoGrad", optimizer_params=dict(beta1=0.95, beta2=0.98, weight_decay=0.001),
                                  larc_params=dict(larc_eta=0.001), loss_scaling="Backoff",
                                  loss_scaling_params=dict(step_window=6), l2=0.0,
                                  lr=("poly_decay", dict(learning_rate=0.02, decay_steps=40, power=2.0, min_lr=1e-5))),
    "adam_backoff_l2": dict(optimizer="Adam", optimizer_params=dict(beta1=0.9, beta2=0.997, epsilon=1e-9),
                            loss_scaling="Backoff", loss_scaling_params=dict(step_window=5), l2=1e-3,
                            lr=("transformer_policy", dict(learning_rate=2.0, d_model=64, warmup_steps=8))),
    "momentum_clip_logmax": dict(optimizer="Momentum", optimizer_params=dict(momentum=0.9), clip_gradients=0.5,
                                 loss_scaling="LogMax", loss_scaling_params={}, l2=5e-4,
                                 lr=("exp_decay", dict(learning_rate=0.05, decay_steps=10, decay_rate=0.7,
                                                       use_staircase_decay=False))),
}
TRAIN_OP_VARS = [("w1", (6, 10), "float16", 3.0), ("w2", (10, 4), "float16", 9.0), ("bias", (4,), "float16", 1.0),
                 ("gamma", (10,), "float32", 2.0)]


def train_op_constants(seed=31):
  """Per toy variable: initial value, A (linear term, magnitude `amp`: 9 x 2^14 overflows fp16) and B > 0 (curvature).
  loss = sum_v sum(cast32(v) * A + 0.5 * cast32(v)^2 * B)."""
  rs = np.random.RandomState(seed)
  out = {}
  for name, shape, dt, amp in TRAIN_OP_VARS:
    out[name] = (rs.standard_normal(shape).astype(dt), (amp * rs.standard_normal(shape)).astype(np.float32),
                 rs.uniform(0.5, 1.5, size=shape).astype(np.float32))
  return out


def train_op(steps=28):
  out = {"steps": np.int32(steps)}
  for case, cfg in TRAIN_OP_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    opt_mod = imp("open_seq2seq.optimizers.optimizers")
    pol = imp("open_seq2seq.optimizers.lr_policies")
    mpw = imp("open_seq2seq.optimizers.mp_wrapper")
    optimizer = imp("open_seq2seq.optimizers.novograd").NovoGrad if cfg["optimizer"] == "NovoGrad" else cfg["optimizer"]
    consts = train_op_constants()
    reg = mpw.mp_regularizer_wrapper(tf.contrib.layers.l2_regularizer(cfg["l2"])) if cfg["l2"] else None
    tvars, terms = [], []
    with tf.variable_scope("ForwardPass"):
      for name, shape, dt, amp in TRAIN_OP_VARS:
        init, A, B = consts[name]
        v = tf.get_variable(name, shape=list(shape), dtype=tf.as_dtype(dt), initializer=tf.constant(init),
                            regularizer=reg if (dt == "float16" and len(shape) == 2) else None)
        tvars.append(v)
        v32 = tf.cast(v, tf.float32)
        terms.append(tf.reduce_sum(v32 * tf.constant(A) + 0.5 * v32 * v32 * tf.constant(B)))
    loss = tf.add_n(terms)
    lr_name, lr_params = cfg["lr"]
    train = opt_mod.optimize_loss(
        loss=loss, optimizer=optimizer, optimizer_params=dict(cfg["optimizer_params"]),
        learning_rate_decay_fn=lambda gs: getattr(pol, lr_name)(global_step=gs, **lr_params), dtype="mixed",
        clip_gradients=cfg.get("clip_gradients"), larc_params=cfg.get("larc_params"),
        loss_scaling=cfg["loss_scaling"], loss_scaling_params=dict(cfg["loss_scaling_params"]), on_horovod=False)
    scaled = [g for g, _ in tf.train.Optimizer.LAST_COMPUTED]          # d(loss * scale) / d(variable), variable dtype
    gs = tf.train.get_global_step()
    masters = {v.name.split(":")[0]: v for v in tf.get_collection("FP32_MASTER_COPIES")}
    scale_var = [v for v in tf.global_variables() if v.name.startswith("Loss_Optimization/Variable")]
    # the scaler's variables in creation order: Backoff (iteration, last_overflow_iteration, scale), LogMax
    # (iteration, scale, ...): the loss scale is the first float32 scalar among them
    scale_var = [v for v in scale_var if v._var.dtype.is_floating_point and v._var.dim() == 0][0]
    lr_t = getattr(pol, lr_name)(global_step=gs, **lr_params)
    rec = {k: [] for k in ("scale_in", "lr", "global_step_after")}
    for name, _, _, _ in TRAIN_OP_VARS:
      rec["g/" + name], rec["w/" + name], rec["m/" + name] = [], [], []
    with tf.Session() as sess:
      for st in range(steps):
        pre = sess.run({"scale": scale_var, "lr": lr_t, "g": scaled})
        sess.run(train)
        rec["scale_in"].append(pre["scale"])
        rec["lr"].append(pre["lr"])
        rec["global_step_after"].append(sess.run(gs))
        for (name, _, dt, _), g, v in zip(TRAIN_OP_VARS, pre["g"], tvars):
          rec["g/" + name].append(np.asarray(g, np.float32))
          rec["w/" + name].append(np.asarray(sess.run(v), np.float32))
          mk = [k for k in masters if k.endswith("/" + name)]
          rec["m/" + name].append(np.asarray(sess.run(masters[mk[0]]), np.float32) if mk
                                   else rec["w/" + name][-1])
      out[case + "/scale_final"] = np.float32(sess.run(scale_var))
    assert (len(masters) == 3), sorted(masters)
    for k, v in rec.items():
      out[case + "/" + k] = np.stack(v)
    out[case + "/master_names"] = np.array(sorted(masters))
  return out


# ---------------------------------------------------------------------------------------------------------
# DeepSpeech2: DeepSpeech2Encoder._encode (encoders/ds2_encoder.py:158-401) — conv2d + BatchNorm + ReLU blocks
# through conv_bn_actv('conv2d'), the [B, T, F, C] -> [T, B, F * C] hand-over to the cuDNN GRU (run over the whole
# padded length: no sequence lengths), row convolution, dense + ReLU. Two cases: bidirectional (ds2_large_8gpus.py)
# and unidirectional + row_conv. The cuDNN RNN is a TensorFlow library object: the stand-in builds it on
# torch.nn.GRU (same cell equations) and exposes its parameters under torch's names.
# ---------------------------------------------------------------------------------------------------------
DS2_CASES = {
    "bidir": dict(unidir=False, row_conv=False, layers=2),
    "unidir_rowconv": dict(unidir=True, row_conv=True, layers=1),
}
DS2_CONV = [{"kernel_size": [5, 7], "stride": [2, 2], "num_channels": 6, "padding": "SAME"},
            {"kernel_size": [5, 5], "stride": [1, 2], "num_channels": 8, "padding": "SAME"}]


def ds2(seed=17, B=3, T=37, F=20, H=12, NH=16):
  out = {"dims": np.array([B, T, F, H, NH], np.int32)}
  for case, cfg in DS2_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    Enc = imp("open_seq2seq.encoders.ds2_encoder").DeepSpeech2Encoder
    rng = np.random.RandomState(seed)
    src_len = np.array([T, T - 9, T // 2], np.int32)
    x = tdnn_input(seed, src_len, T, F)
    params = dict(dropout_keep_prob=1.0, conv_layers=DS2_CONV, activation_fn=tf.nn.relu, num_rnn_layers=cfg["layers"],
                  row_conv=cfg["row_conv"], row_conv_width=4, n_hidden=NH, use_cudnn_rnn=True, rnn_cell_dim=H,
                  rnn_type="cudnn_gru", rnn_unidirectional=cfg["unidir"], bn_momentum=0.99, bn_epsilon=1e-3,
                  dtype=tf.float32)
    with tf.variable_scope("ForwardPass"):
      enc = Enc(params, None, name="ds2_encoder", mode="train")
      res = enc.encode({"source_tensors": [tf.constant(x), tf.constant(src_len)]})
    outputs = res["outputs"]
    R = rng.standard_normal(tuple(int(v) for v in outputs.get_shape())).astype(np.float32)
    loss = tf.reduce_sum(outputs * tf.constant(R))
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    with tf.Session() as sess:
      for n, v in zip(names, tvars):
        if v._var.dim() == 1 and "cudnn" not in n:       # BatchNorm / bias vectors away from 1 / 0
          v.load(seeded_array(n, tuple(v._var.shape), seed))
      vals = sess.run({"out": outputs, "len": res["src_length"], "grads": tf.gradients(loss, tvars),
                       "vars": list(tvars)})
    out.update({case + "/src_len": src_len, case + "/out": vals["out"], case + "/out_len": vals["len"].astype(np.int32),
                case + "/R": R, case + "/var_names": np.array(names), case + "/seed": np.int32(seed)})
    for n, v, g in zip(names, vals["vars"], vals["grads"]):
      out["%s/var/%s" % (case, n)] = v.astype(np.float32)
      out["%s/grad/%s" % (case, n)] = g.astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# RNN NMT decoder: RNNDecoderWithAttention._decode (decoders/rnn_decoders.py:147-321) in train mode — decoder
# embedding, single_cell (parts/rnns/utils.py), BahdanauAttention(normalize=True) / AttentionWrapper
# (parts/rnns/attention_wrapper.py), GNMTAttentionMultiCell + gnmt_residual_fn (parts/rnns/gnmt.py), the output
# projection, TrainingHelper / BasicDecoder / dynamic_decode(impute_finished=True) — followed by BasicSequenceLoss
# (losses/sequence_loss.py:10-114). The cell classes, dynamic_decode and the helpers are TensorFlow library code
# (restated in oracle/ref_shim/tf1/rnn.py: one traced step, replayed); everything else runs from the reference.
# ---------------------------------------------------------------------------------------------------------
NMT_CASES = {
    "gnmt_v2": dict(attention_type="gnmt_v2", skip=False, layers=2),
    "gnmt": dict(attention_type="gnmt", skip=False, layers=2),
    "gnmt_v2_skip": dict(attention_type="gnmt_v2", skip=True, layers=3),
}
NMT_DIMS = dict(B=3, S=7, T=6, V=24, E=12, H=14, M=16, U=10)


def nmt_decoder(seed=41):
  out = {"dims": np.array([NMT_DIMS[k] for k in ("B", "S", "T", "V", "E", "H", "M", "U")], np.int32)}
  D = NMT_DIMS
  for case, cfg in NMT_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    Dec = imp("open_seq2seq.decoders.rnn_decoders").RNNDecoderWithAttention
    Loss = imp("open_seq2seq.losses.sequence_loss").BasicSequenceLoss
    rng = np.random.RandomState(seed)
    B, S, T, V, E, H, M, U = [D[k] for k in ("B", "S", "T", "V", "E", "H", "M", "U")]
    src_len = np.array([7, 4, 5], np.int32)
    tgt_len = np.array([6, 3, 5], np.int32)
    tgt = rng.randint(3, V, size=(B, T)).astype(np.int32)
    tgt[:, 0] = 1
    for b in range(B):
      tgt[b, tgt_len[b] - 1] = 2
      tgt[b, tgt_len[b]:] = 0
    enc = rng.standard_normal((B, S, M)).astype(np.float32)
    params = dict(GO_SYMBOL=1, END_SYMBOL=2, tgt_vocab_size=V, tgt_emb_size=E, attention_layer_size=U,
                  attention_type=cfg["attention_type"], core_cell=tf.nn.rnn_cell.LSTMCell,
                  core_cell_params={"num_units": H, "forget_bias": 1.0}, decoder_layers=cfg["layers"],
                  decoder_use_skip_connections=cfg["skip"], batch_size=B, decoder_dp_input_keep_prob=1.0,
                  decoder_dp_output_keep_prob=1.0, dtype=tf.float32)
    with tf.variable_scope("ForwardPass"):
      enc_var = tf.get_variable("encoder_outputs", initializer=tf.constant(enc))
      dec = Dec(params, None, mode="train")
      res = dec.decode({"encoder_output": {"outputs": enc_var, "src_lengths": tf.constant(src_len)},
                        "target_tensors": [tf.constant(tgt), tf.constant(tgt_len)]})
      loss = Loss(dict(tgt_vocab_size=V, batch_size=B, offset_target_by_one=True, average_across_timestep=False,
                       do_mask=True, dtype=tf.float32), None).compute_loss(
          {"decoder_output": res, "target_tensors": [tf.constant(tgt), tf.constant(tgt_len)]})
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    with tf.Session() as sess:
      for n, v in zip(names, tvars):
        if v._var.dim() == 1:                 # biases / attention vectors away from their 0 / constant initial values
          v.load(_np(v._var) + 0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
      vals = sess.run({"logits": res["logits"], "loss": loss, "lens": res["final_sequence_lengths"],
                       "grads": tf.gradients(loss, tvars), "vars": list(tvars)})
    out.update({case + "/src_len": src_len, case + "/tgt_len": tgt_len, case + "/tgt": tgt,
                case + "/logits": vals["logits"], case + "/loss": np.float32(vals["loss"]),
                case + "/final_sequence_lengths": vals["lens"].astype(np.int32), case + "/var_names": np.array(names)})
    for n, v, g in zip(names, vals["vars"], vals["grads"]):
      out["%s/var/%s" % (case, n)] = v.astype(np.float32)
      out["%s/grad/%s" % (case, n)] = g.astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# RNN NMT encoders: BidirectionalRNNEncoderWithEmbedding and GNMTLikeEncoderWithEmbedding
# (encoders/rnn_encoders.py:221-305, 380-470): embedding, single_cell stacks, bidirectional_dynamic_rnn /
# dynamic_rnn with sequence lengths, ResidualWrapper on the upper unidirectional layers.
# ---------------------------------------------------------------------------------------------------------
NMT_ENC_CASES = {"bidir": dict(cls="BidirectionalRNNEncoderWithEmbedding", layers=2),
                 "gnmt_like": dict(cls="GNMTLikeEncoderWithEmbedding", layers=3)}


def nmt_encoder(seed=43, B=3, S=7, V=20, E=10, H=12):
  return _nmt_encoder(seed, B, S, V, E, H, NMT_ENC_CASES, [7, 3, 5])


def nmt_encoder_dev(seed=71):
  """GNMTLikeEncoderWithEmbedding (encoders/rnn_encoders.py:320-470: one bidirectional LSTM layer, then
  unidirectional layers, residual connections from the third layer on) at widths the device's recurrent kernels
  take — embedding 64, 64 units, three layers, a ragged batch of 4 x 12: what tests/test_ref_exec_nmt_gpu.py holds
  the HIP encoder against (outputs and every variable's gradient)."""
  return _nmt_encoder(seed, 4, 12, 40, 64, 64, {"gnmt_like": NMT_ENC_CASES["gnmt_like"]}, [12, 5, 9, 7])


def _nmt_encoder(seed, B, S, V, E, H, cases, lens):
  out = {"dims": np.array([B, S, V, E, H], np.int32)}
  for case, cfg in cases.items():
    tf, imp = _install()
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    Enc = getattr(imp("open_seq2seq.encoders.rnn_encoders"), cfg["cls"])
    rng = np.random.RandomState(seed)
    src_len = np.array(lens, np.int32)
    src = rng.randint(3, V, size=(B, S)).astype(np.int32)
    for b in range(B):
      src[b, src_len[b]:] = 0
    params = dict(src_vocab_size=V, src_emb_size=E, core_cell=tf.nn.rnn_cell.LSTMCell,
                  core_cell_params={"num_units": H, "forget_bias": 1.0}, encoder_layers=cfg["layers"],
                  encoder_use_skip_connections=False, encoder_dp_input_keep_prob=1.0,
                  encoder_dp_output_keep_prob=1.0, dtype=tf.float32)
    with tf.variable_scope("ForwardPass"):
      res = Enc(params, None, mode="train").encode({"source_tensors": [tf.constant(src), tf.constant(src_len)]})
    outputs = res["outputs"]
    R = rng.standard_normal(tuple(int(v) for v in outputs.get_shape())).astype(np.float32)
    loss = tf.reduce_sum(outputs * tf.constant(R))
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    with tf.Session() as sess:
      for n, v in zip(names, tvars):
        if v._var.dim() == 1:
          v.load(0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
      vals = sess.run({"out": outputs, "grads": tf.gradients(loss, tvars), "vars": list(tvars)})
    out.update({case + "/src": src, case + "/src_len": src_len, case + "/out": vals["out"], case + "/R": R,
                case + "/var_names": np.array(names)})
    for n, v, g in zip(names, vals["vars"], vals["grads"]):
      out["%s/var/%s" % (case, n)] = v.astype(np.float32)
      out["%s/grad/%s" % (case, n)] = g.astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# Tacotron 2 decoder: Tacotron2Decoder._decode in train mode (decoders/tacotron2_decoder.py:257-567) — pre-net with
# its always-on dropout, single_cell LSTM stack inside AttentionWrapper(output_attention="both",
# alignment_history=True) over LocationSensitiveAttention (parts/rnns/attention_wrapper.py:641-878: Chorowski
# location layer on the CUMULATIVE alignments, optional attention bias), TacotronDecoder + TacotronTrainingHelper
# (parts/tacotron/*.py) under dynamic_decode(impute_finished=False), output / stop-token projections, the five-layer
# post-net through conv_bn_actv, the magnitude branch of "both" mode. The pre-net's dropout masks are random in the
# reference: the stand-in's dropout records them (tf1.DROPOUT_TAP) and they are part of the fixture.
# ---------------------------------------------------------------------------------------------------------
TACO_DIMS = dict(B=3, S=9, T=7, M=12, H=16, U=10, P=8, NMEL=8, NMAG=12, K=5, F=6)


def tacotron_decoder(seed=47):
  out = {"dims": np.array([TACO_DIMS[k] for k in ("B", "S", "T", "M", "H", "U", "P", "NMEL", "NMAG", "K", "F")],
                          np.int32)}
  D = TACO_DIMS
  for case, bias in (("location", False), ("location_bias", True)):
    tf, imp = _install()
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    Dec = imp("open_seq2seq.decoders.tacotron2_decoder").Tacotron2Decoder
    rng = np.random.RandomState(seed)
    B, S, T, M, H, U, P, NMEL, NMAG, K, F = [D[k] for k in ("B", "S", "T", "M", "H", "U", "P", "NMEL", "NMAG", "K", "F")]
    src_len = np.array([9, 5, 7], np.int32)
    spec_len = np.array([7, 4, 6], np.int32)
    enc = rng.standard_normal((B, S, M)).astype(np.float32)
    spec = rng.standard_normal((B, T, NMEL + NMAG)).astype(np.float32)

    class _DL(object):
      params = {"num_audio_features": {"mel": NMEL, "magnitude": NMAG}, "output_type": "both"}
      _exp_mag = True

    class _Model(object):
      params = {"dtype": tf.float32}

      def get_data_layer(self):
        return _DL()
    postnet = [{"kernel_size": [5], "stride": [1], "num_channels": 10, "padding": "SAME", "activation_fn": tf.nn.tanh},
               {"kernel_size": [5], "stride": [1], "num_channels": 10, "padding": "SAME", "activation_fn": tf.nn.tanh},
               {"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]
    params = dict(attention_layer_size=U, attention_type="location", attention_bias=bias, decoder_cell_units=H,
                  decoder_cell_type=tf.nn.rnn_cell.LSTMCell, decoder_layers=2, enable_prenet=True, prenet_layers=2,
                  prenet_units=P, enable_postnet=True, postnet_conv_layers=postnet, postnet_keep_dropout_prob=1.0,
                  postnet_bn_momentum=0.1, postnet_bn_epsilon=1e-5, mask_decoder_sequence=True, zoneout_prob=0.0,
                  dropout_prob=0.0, dtype=tf.float32)
    with tf.variable_scope("ForwardPass"):
      enc_var = tf.get_variable("encoder_outputs", initializer=tf.constant(enc))
      dec = Dec(params, _Model(), mode="train")
      # the location layer's sizes are fixed to 32 x 32 in the reference unless location_attention_params is given,
      # which _build_attention does not pass on: the fixture runs at the reference's 32 taps / 32 filters
      res = dec.decode({"encoder_output": {"outputs": enc_var, "src_length": tf.constant(src_len)},
                        "target_tensors": [tf.constant(spec), tf.constant(np.zeros((B, T), np.float32)),
                                           tf.constant(spec_len)]})
    dec_out, post, align, stop_sig, seq_lens, mag = res["outputs"]
    stop_logits = res["stop_token_prediction"]
    Rs = [rng.standard_normal(tuple(int(v) for v in t.get_shape())).astype(np.float32) for t in (dec_out, post, stop_logits, mag)]
    loss = tf.add_n([tf.reduce_sum(t * tf.constant(r)) for t, r in zip((dec_out, post, stop_logits, mag), Rs)])
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    with tf.Session() as sess:
      big = {n for n, v in zip(names, tvars) if v._var.numel() > 4096}   # the magnitude branch's fixed 256 / 512 widths
      for n, v in zip(names, tvars):
        if n in big:
          v.load(seeded_array(n, tuple(v._var.shape), seed))
        elif v._var.dim() == 1:
          v.load(_np(v._var) + 0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
      tf1 = tf
      tf1.DROPOUT_TAP = []
      vals = sess.run({"mel": dec_out, "post": post, "align": align, "stop": stop_logits, "lens": seq_lens, "mag": mag,
                       "loss": loss, "grads": tf.gradients(loss, tvars), "vars": list(tvars)})
      masks = [_np(m) for m in tf1.DROPOUT_TAP]
      tf1.DROPOUT_TAP = None
    assert len(masks) == 2 * T, len(masks)
    out.update({case + "/src_len": src_len, case + "/spec_len": spec_len, case + "/spec": spec,
                case + "/mel": vals["mel"], case + "/post": vals["post"], case + "/align": vals["align"],
                case + "/stop": vals["stop"], case + "/mag": vals["mag"], case + "/lens": vals["lens"].astype(np.int32),
                case + "/loss": np.float32(vals["loss"]), case + "/var_names": np.array(names),
                case + "/prenet_mask0": np.stack(masks[0::2], 1), case + "/prenet_mask1": np.stack(masks[1::2], 1)})
    out[case + "/seed"] = np.int32(seed)
    for i, r in enumerate(Rs):
      out["%s/R%d" % (case, i)] = r
    for n, v, g in zip(names, vals["vars"], vals["grads"]):
      if n in big:                       # regenerated from the seed by the tests; gradient as (norm, projection)
        out["%s/shape/%s" % (case, n)] = np.array(v.shape, np.int32)
        out["%s/gproj/%s" % (case, n)] = projection(n, g, seed)
      else:
        out["%s/var/%s" % (case, n)] = v.astype(np.float32)
        out["%s/grad/%s" % (case, n)] = g.astype(np.float32)
  return out


TACO_ENC = dict(B=3, S=9, V=30, E=10, C=12, H=7, NS=16, TS=21, SC1=4, SC2=6, SH=9, NTOK=5, TOKE=8, ATT=8, HEADS=2)
TACO_STYLE_CONV = [{"kernel_size": [3, 3], "stride": [2, 2], "num_channels": 4, "padding": "SAME"},
                   {"kernel_size": [3, 3], "stride": [2, 2], "num_channels": 6, "padding": "SAME"}]


def tacotron_encoder(seed=61):
  """Tacotron2Encoder._encode with global style tokens (encoders/tacotron2_encoder.py:104-505): embedding, three
  conv_bn_actv layers, the cuDNN bidirectional LSTM over the whole padded length, and _embed_style — conv2d + BatchNorm
  + ReLU blocks over the style spectrogram, tf.nn.rnn_cell.GRUCell under dynamic_rnn (final state at each sample's
  length), Dense(128, tanh), the reference's multi-head Attention in "bahdanau" mode over tanh(style tokens) — tiled
  over time and concatenated. Train mode, dropout probabilities 0."""
  D = TACO_ENC
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  Enc = imp("open_seq2seq.encoders.tacotron2_encoder").Tacotron2Encoder
  rng = np.random.RandomState(seed)
  B, S, V, E, C, H, NS, TS = [D[k] for k in ("B", "S", "V", "E", "C", "H", "NS", "TS")]
  text_len = np.array([9, 6, 4], np.int32)
  text = rng.randint(1, V, size=(B, S)).astype(np.int32)
  for b in range(B):
    text[b, text_len[b]:] = 0
  style_len = np.array([21, 13, 17], np.int32)
  style = rng.standard_normal((B, TS, NS)).astype(np.float32)
  for b in range(B):
    style[b, style_len[b]:] = 0.0

  class _DL(object):
    params = {"src_vocab_size": V, "style_input": "wav"}

  class _Model(object):
    params = {"dtype": tf.float32}

    def get_data_layer(self):
      return _DL()
  conv = [{"kernel_size": [5], "stride": [1], "num_channels": C, "padding": "SAME"} for _ in range(3)]
  params = dict(src_emb_size=E, conv_layers=conv, activation_fn=tf.nn.relu, num_rnn_layers=1, rnn_cell_dim=H,
                rnn_type=tf.contrib.cudnn_rnn.CudnnLSTM, use_cudnn_rnn=True, rnn_unidirectional=False,
                cnn_dropout_prob=0.0, rnn_dropout_prob=0.0, zoneout_prob=0.0, data_format="channels_last",
                style_embedding_enable=True,
                style_embedding_params=dict(conv_layers=TACO_STYLE_CONV, num_rnn_layers=1, rnn_cell_dim=D["SH"],
                                            rnn_unidirectional=True, rnn_type=tf.nn.rnn_cell.GRUCell,
                                            emb_size=D["TOKE"], attention_layer_size=D["ATT"], num_tokens=D["NTOK"],
                                            num_heads=D["HEADS"]),
                dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    res = Enc(params, _Model(), mode="train").encode(
        {"source_tensors": [tf.constant(text), tf.constant(text_len), tf.constant(style), tf.constant(style_len)]})
  outputs = res["outputs"]
  R = rng.standard_normal(tuple(int(v) for v in outputs.get_shape())).astype(np.float32)
  loss = tf.reduce_sum(outputs * tf.constant(R))
  gvars = [v for v in tf.global_variables() if "moving_" not in v.name and "global_step" not in v.name]
  names = [v.name.split(":")[0] for v in gvars]
  with tf.Session() as sess:
    for n, v in zip(names, gvars):
      if v._var.dim() == 1 and "cudnn" not in n:
        v.load(_np(v._var) + 0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
    vals = sess.run({"out": outputs, "len": res["src_length"], "grads": tf.gradients(loss, gvars), "vars": list(gvars)})
  out = {"dims": np.array([D[k] for k in sorted(D)], np.int32), "dim_names": np.array(sorted(D)), "text": text,
         "text_len": text_len, "style": style, "style_len": style_len, "out": vals["out"],
         "out_len": vals["len"].astype(np.int32), "R": R, "var_names": np.array(names)}
  for n, v, g in zip(names, vals["vars"], vals["grads"]):
    out["var/" + n] = v.astype(np.float32)
    out["grad/" + n] = g.astype(np.float32)
  return out


def tacotron_infer(seed=59):
  """Tacotron2Decoder._decode in eval mode (decoders/tacotron2_decoder.py:378-428): TacotronHelper feeds every
  projected frame back through the pre-net, finished = round(sigmoid(stop logit)), the loop ends when every sample has
  finished (or after 10 x max(src_len) steps); sequence lengths as dynamic_decode counts them. The stop projection's
  bias and gain are searched (deterministically) until a sample finishes in the middle of the run while another one keeps
  decoding (random LSTM trajectories settle quickly: most settings finish at step 1 or never)."""
  D = TACO_DIMS
  B, S, M, H, U, P, NMEL, NMAG = [D[k] for k in ("B", "S", "M", "H", "U", "P", "NMEL", "NMAG")]
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  Dec = imp("open_seq2seq.decoders.tacotron2_decoder").Tacotron2Decoder
  rng = np.random.RandomState(seed)
  src_len = np.array([3, 2, 3], np.int32)                       # limit = 10 * 3 = 30 steps
  S = 3
  enc = rng.standard_normal((B, S, M)).astype(np.float32)

  class _DL(object):
    params = {"num_audio_features": NMEL, "output_type": "mel"}

  class _Model(object):
    params = {"dtype": tf.float32}

    def get_data_layer(self):
      return _DL()
  postnet = [{"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]
  params = dict(attention_layer_size=U, attention_type="location", attention_bias=True, decoder_cell_units=H,
                decoder_cell_type=tf.nn.rnn_cell.LSTMCell, decoder_layers=2, enable_prenet=True, prenet_layers=2,
                prenet_units=P, enable_postnet=True, postnet_conv_layers=postnet, postnet_keep_dropout_prob=1.0,
                mask_decoder_sequence=True, zoneout_prob=0.0, dropout_prob=0.0, dtype=tf.float32)
  with tf.variable_scope("ForwardPass"):
    dec = Dec(params, _Model(), mode="eval")
    res = dec.decode({"encoder_output": {"outputs": tf.constant(enc), "src_length": tf.constant(src_len)}})
  dec_out, post, align, stop_sig, seq_lens, _ = res["outputs"]
  gvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in gvars]
  stop_b = [v for v in gvars if v.name.endswith("stop_token_proj/bias:0")][0]
  stop_k = [v for v in gvars if v.name.endswith("stop_token_proj/kernel:0")][0]
  out_k = [v for v in gvars if v.name.endswith("output_proj/kernel:0")][0]
  with tf.Session() as sess:
    for n, v in zip(names, gvars):
      if v._var.dim() == 1:
        v.load(_np(v._var) + 0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
    stop_k0, out_k0 = _np(stop_k._var).copy(), _np(out_k._var).copy()
    chosen = None
    for bias, gain in [(b, g) for g in (3.0, 10.0, 30.0) for b in np.linspace(-3.0, 3.0, 61)]:
      stop_b.load(np.array([bias], np.float32))
      stop_k.load(gain * stop_k0)
      out_k.load(2.0 * out_k0)
      tf._RNG.manual_seed(seed)
      tf.DROPOUT_TAP = []
      vals = sess.run({"mel": dec_out, "align": align, "stop": res["stop_token_prediction"], "lens": seq_lens,
                       "vars": list(gvars)})
      masks = [_np(m) for m in tf.DROPOUT_TAP]
      tf.DROPOUT_TAP = None
      lens = vals["lens"]
      mid = [int(v) for v in lens if 3 <= v <= 28]
      if len(set(lens.tolist())) >= 2 and mid:
        chosen = (float(bias), float(gain))
        break
  assert chosen is not None, "no stop bias / gain lets a sample finish in the middle of the run"
  steps = vals["mel"].shape[1]
  assert len(masks) == 2 * steps, (len(masks), steps)
  out = {"dims": np.array([B, S, M, H, U, P, NMEL], np.int32), "src_len": src_len, "enc": enc, "mel": vals["mel"],
         "align": vals["align"], "stop": vals["stop"], "lens": vals["lens"].astype(np.int32),
         "steps": np.int32(steps), "var_names": np.array(names),
         "prenet_mask0": np.stack(masks[0::2], 0), "prenet_mask1": np.stack(masks[1::2], 0)}
  for n, v in zip(names, vals["vars"]):
    out["var/" + n] = v.astype(np.float32)
  return out


TACO_DEV = dict(B=4, S=12, M=64, H=64, U=128, P=64, NMEL=16, src_len=[12, 8, 10, 6])


def tacotron_infer_dev(seed=67):
  """The same free-running decode (Tacotron2Decoder._decode in eval mode with TacotronHelper,
  decoders/tacotron2_decoder.py:378-428, parts/tacotron/tacotron_helper.py:138-226) at widths the device's fused decode
  kernels take (decoder_cell_units = memory = 64, attention layer 128, 32 location filters of 31 taps, pre-net 2 x 64,
  16 mel bins, a two-layer post-net): tests/test_ref_exec_tacotron_gpu.py holds the HIP decode against these frames,
  stop logits, alignments and lengths. The pre-net's always-on dropout (tacotron2_decoder.py: Prenet at keep 0.5) is
  switched off on both sides — the device draws its masks from a counter hash inside the step kernel, the reference
  from tf.nn.dropout: the hash is pinned separately (test_tacotron_infer_gpu.py against the oracle). The stop
  projection's bias / gain are searched until a sample finishes in the middle of the run while another keeps decoding."""
  D = TACO_DEV
  B, S, M, H, U, P, NMEL = [D[k] for k in ("B", "S", "M", "H", "U", "P", "NMEL")]
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  Dec = imp("open_seq2seq.decoders.tacotron2_decoder").Tacotron2Decoder
  rng = np.random.RandomState(seed)
  src_len = np.array(D["src_len"], np.int32)                  # limit = 10 * 12 = 120 steps
  enc = (0.5 * rng.standard_normal((B, S, M))).astype(np.float32)
  enc *= (np.arange(S)[None, :, None] < src_len[:, None, None])

  class _DL(object):
    params = {"num_audio_features": NMEL, "output_type": "mel"}

  class _Model(object):
    params = {"dtype": tf.float32}

    def get_data_layer(self):
      return _DL()
  postnet = [{"kernel_size": [5], "stride": [1], "num_channels": 64, "padding": "SAME", "activation_fn": tf.nn.tanh},
             {"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]
  params = dict(attention_layer_size=U, attention_type="location", attention_bias=True, decoder_cell_units=H,
                decoder_cell_type=tf.nn.rnn_cell.LSTMCell, decoder_layers=2, enable_prenet=True, prenet_layers=2,
                prenet_units=P, enable_postnet=True, postnet_conv_layers=postnet, postnet_keep_dropout_prob=1.0,
                mask_decoder_sequence=True, zoneout_prob=0.0, dropout_prob=0.0, dtype=tf.float32)
  tf.DROPOUT_OFF = True
  try:
    with tf.variable_scope("ForwardPass"):
      dec = Dec(params, _Model(), mode="eval")
      res = dec.decode({"encoder_output": {"outputs": tf.constant(enc), "src_length": tf.constant(src_len)}})
    dec_out, post, align, stop_sig, seq_lens, _ = res["outputs"]
    gvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in gvars]
    stop_b = [v for v in gvars if v.name.endswith("stop_token_proj/bias:0")][0]
    stop_k = [v for v in gvars if v.name.endswith("stop_token_proj/kernel:0")][0]
    out_k = [v for v in gvars if v.name.endswith("output_proj/kernel:0")][0]
    with tf.Session() as sess:
      for n, v in zip(names, gvars):
        if v._var.dim() == 1:
          v.load(_np(v._var) + 0.2 * rng.standard_normal(tuple(v._var.shape)).astype(np.float32))
      stop_k0, out_k0 = _np(stop_k._var).copy(), _np(out_k._var).copy()
      chosen = None
      # (random LSTM trajectories settle within a few steps and the first step — the all-zero go frame — carries the
      # largest stop logit: with the frame projection at 16 x its initial scale the fed-back frames keep the state
      # moving and one sample's logit crosses zero around step 30)
      for bias, gain in [(b, g) for g in (1.0, 3.0) for b in np.linspace(-0.5, 0.5, 41)]:
        stop_b.load(np.array([bias], np.float32))
        stop_k.load(gain * stop_k0)
        out_k.load(16.0 * out_k0)
        tf._RNG.manual_seed(seed)
        vals = sess.run({"mel": dec_out, "post": post, "align": align, "stop": res["stop_token_prediction"],
                         "lens": seq_lens, "vars": list(gvars)})
        lens = [int(v) for v in vals["lens"]]
        if any(8 <= v <= 100 for v in lens) and max(lens) == 10 * int(src_len.max()):
          chosen = (float(bias), float(gain))
          break
  finally:
    tf.DROPOUT_OFF = False
  assert chosen is not None, "no stop bias / gain lets a sample finish in the middle of the run while another keeps decoding"
  steps = vals["mel"].shape[1]
  out = {"dims": np.array([B, S, M, H, U, P, NMEL], np.int32), "src_len": src_len, "enc": enc, "mel": vals["mel"],
         "post": vals["post"], "align": vals["align"], "stop": vals["stop"], "lens": vals["lens"].astype(np.int32),
         "steps": np.int32(steps), "var_names": np.array(names), "stop_search": np.array(chosen, np.float32)}
  for n, v in zip(names, vals["vars"]):
    out["var/" + n] = v.astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# Text2SpeechLoss (losses/text2speech_loss.py:35-209) on synthetic predictions: "both" mode, predictions shorter and
# longer than the targets (the pad-to-common-length branch), mask on / off, l1 / l2, weights and scale.
# ---------------------------------------------------------------------------------------------------------
T2S_CASES = {
    "masked_l2_pred_longer": dict(Tp=9, Tt=7, use_mask=True, l1_norm=False),
    "masked_l1_pred_shorter": dict(Tp=5, Tt=8, use_mask=True, l1_norm=True, mel_weight=0.7, mag_weight=1.3,
                                   stop_token_weight=2.0, scale=0.5),
    "unmasked_l2_equal": dict(Tp=6, Tt=6, use_mask=False, l1_norm=False),
}


def t2s_loss(seed=53, B=3, NMEL=5, NMAG=7):
  out = {"dims": np.array([B, NMEL, NMAG], np.int32)}
  for case, cfg in T2S_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    Loss = imp("open_seq2seq.losses.text2speech_loss").Text2SpeechLoss
    rng = np.random.RandomState(seed)
    Tp, Tt = cfg["Tp"], cfg["Tt"]
    mel, post = [rng.standard_normal((B, Tp, NMEL)).astype(np.float32) for _ in range(2)]
    stop = rng.standard_normal((B, Tp, 1)).astype(np.float32)
    mag = rng.standard_normal((B, Tp, NMAG)).astype(np.float32)
    spec = rng.standard_normal((B, Tt, NMEL + NMAG)).astype(np.float32)
    spec_len = np.array([Tt, max(Tt - 3, 1), max(Tt - 1, 1)], np.int32)
    stop_t = (np.arange(Tt)[None, :] >= (spec_len[:, None] - 1)).astype(np.float32)

    class _DL(object):
      params = {"num_audio_features": {"mel": NMEL, "magnitude": NMAG}, "output_type": "both"}

    class _Model(object):
      def get_data_layer(self):
        return _DL()

      def get_tf_dtype(self):
        return tf.float32
    params = {k: v for k, v in cfg.items() if k not in ("Tp", "Tt")}
    vs = [tf.Variable(a) for a in (mel, post, stop, mag)]
    loss = Loss(params, _Model()).compute_loss(
        {"decoder_output": {"outputs": [vs[0], vs[1], None, None, None, vs[3]], "stop_token_prediction": vs[2]},
         "target_tensors": [tf.constant(spec), tf.constant(stop_t), tf.constant(spec_len)]})
    with tf.Session() as sess:
      vals = sess.run({"loss": loss, "grads": tf.gradients(loss, vs)})
    out.update({case + "/mel": mel, case + "/post": post, case + "/stop": stop, case + "/mag": mag, case + "/spec": spec,
                case + "/spec_len": spec_len, case + "/stop_target": stop_t, case + "/loss": np.float32(vals["loss"])})
    for k, g in zip(("mel", "post", "stop", "mag"), vals["grads"]):
      out["%s/grad/%s" % (case, k)] = g.astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# Transformer beam search: parts/transformer/beam_search.py:sequence_beam_search (the tf.while_loop over
# _continue_search / _search_step with its alive / finished bookkeeping, length normalisation, 2 x beam candidates)
# driven by a TABLE symbols_to_logits_fn: logits = table[step][last id] + cache["bias"] — the cache entry travels
# through the search's expand / flatten / gather plumbing like the decoder's K / V tensors do.
# ---------------------------------------------------------------------------------------------------------
BEAM_CASES = {
    "finishes": dict(B=3, V=9, beam=4, alpha=0.6, L=7, eos=1, eos_boost=1.5, seed=71),
    "never_finishes": dict(B=2, V=7, beam=3, alpha=1.0, L=5, eos=1, eos_boost=-30.0, seed=73),
    "early_stop": dict(B=2, V=8, beam=2, alpha=0.0, L=9, eos=1, eos_boost=6.0, seed=79),
}


def beam_tables(cfg):
  rs = np.random.RandomState(cfg["seed"])
  table = rs.standard_normal((cfg["L"] + 1, cfg["V"], cfg["V"])).astype(np.float32)
  table[:, :, cfg["eos"]] += np.float32(cfg["eos_boost"])
  bias = (0.5 * rs.standard_normal((cfg["B"], cfg["V"]))).astype(np.float32)
  return table, bias


def beam_search():
  out = {}
  for case, cfg in BEAM_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    bs = imp("open_seq2seq.parts.transformer.beam_search")
    table, bias = beam_tables(cfg)
    tab = tf.constant(table)

    def fn(ids, i, cache):
      last = ids[:, -1]
      logits = tf.gather(tf.gather(tab, i), last) + cache["bias"]
      return logits, cache
    ids, scores = bs.sequence_beam_search(fn, tf.zeros([cfg["B"]], dtype=tf.int32), {"bias": tf.constant(bias)},
                                          cfg["V"], cfg["beam"], cfg["alpha"], cfg["L"], cfg["eos"])
    with tf.Session() as sess:
      v = sess.run({"ids": ids, "scores": scores})
    out[case + "/ids"], out[case + "/scores"] = v["ids"].astype(np.int32), v["scores"].astype(np.float32)
  return out


# ---------------------------------------------------------------------------------------------------------
# The whole RNN NMT model at widths the HIP kernels take (attention depth 128): BidirectionalRNNEncoderWithEmbedding ->
# RNNDecoderWithAttention (gnmt_v2) -> BasicSequenceLoss, chained as models/encoder_decoder.py chains them. Variables
# from seeded_array (not stored), gradients as (norm, projection).
# ---------------------------------------------------------------------------------------------------------
NMT_FULL = dict(B=4, S=11, T=9, V=30, E=64, H=64, U=128, layers=2)


def nmt_full(seed=67):
  D = NMT_FULL
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  Enc = imp("open_seq2seq.encoders.rnn_encoders").BidirectionalRNNEncoderWithEmbedding
  Dec = imp("open_seq2seq.decoders.rnn_decoders").RNNDecoderWithAttention
  Loss = imp("open_seq2seq.losses.sequence_loss").BasicSequenceLoss
  rng = np.random.RandomState(seed)
  B, S, T, V, E, H, U, NL = [D[k] for k in ("B", "S", "T", "V", "E", "H", "U", "layers")]
  src_len = np.array([11, 6, 9, 3], np.int32)
  tgt_len = np.array([9, 4, 7, 2], np.int32)
  src = rng.randint(4, V, size=(B, S)).astype(np.int32)
  tgt = rng.randint(4, V, size=(B, T)).astype(np.int32)
  for b in range(B):
    src[b, src_len[b]:] = 0
    tgt[b, 0] = 2
    tgt[b, tgt_len[b] - 1] = 1
    tgt[b, tgt_len[b]:] = 0
  cellp = {"num_units": H, "forget_bias": 1.0}
  with tf.variable_scope("ForwardPass"):
    enc = Enc(dict(src_vocab_size=V, src_emb_size=E, core_cell=tf.nn.rnn_cell.LSTMCell, core_cell_params=cellp,
                   encoder_layers=NL, encoder_use_skip_connections=False, encoder_dp_input_keep_prob=1.0,
                   encoder_dp_output_keep_prob=1.0, dtype=tf.float32), None, mode="train")
    dec = Dec(dict(GO_SYMBOL=2, END_SYMBOL=1, tgt_vocab_size=V, tgt_emb_size=E, attention_layer_size=U,
                   attention_type="gnmt_v2", core_cell=tf.nn.rnn_cell.LSTMCell, core_cell_params=dict(cellp),
                   decoder_layers=NL, decoder_use_skip_connections=False, batch_size=B,
                   decoder_dp_input_keep_prob=1.0, decoder_dp_output_keep_prob=1.0, dtype=tf.float32), None,
              mode="train")
    eo = enc.encode({"source_tensors": [tf.constant(src), tf.constant(src_len)]})
    do = dec.decode({"encoder_output": eo, "target_tensors": [tf.constant(tgt), tf.constant(tgt_len)]})
    loss = Loss(dict(tgt_vocab_size=V, batch_size=B, offset_target_by_one=True, average_across_timestep=False,
                     do_mask=True, dtype=tf.float32), None).compute_loss(
        {"decoder_output": do, "target_tensors": [tf.constant(tgt), tf.constant(tgt_len)]})
  tvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in tvars]
  with tf.Session() as sess:
    for n, v in zip(names, tvars):
      v.load(seeded_array(n, tuple(v._var.shape), seed))
    vals = sess.run({"enc": eo["outputs"], "logits": do["logits"], "loss": loss, "grads": tf.gradients(loss, tvars)})
  out = {"src": src, "src_len": src_len, "tgt": tgt, "tgt_len": tgt_len, "enc_out": vals["enc"].astype(np.float16),
         "logits": vals["logits"], "loss": np.float32(vals["loss"]), "var_names": np.array(names),
         "seed": np.int32(seed)}
  for n, v, g in zip(names, tvars, vals["grads"]):
    out["shape/" + n] = np.array(tuple(v._var.shape), np.int32)
    out["gproj/" + n] = projection(n, g, seed)
  return out


# ---------------------------------------------------------------------------------------------------------
# Beam-search inference of the RNN NMT model: BidirectionalRNNEncoderWithEmbedding (infer) ->
# BeamSearchRNNDecoderWithAttention (decoders/rnn_decoders.py:324-532) over the reference's OWN BeamSearchDecoder
# (parts/rnns/rnn_beam_search_decoder.py: initialize / step / _beam_search_step / _mask_probs / _get_scores /
# finalize) under dynamic_decode(maximum_iterations = 2 * max source length). tile_batch, gather_tree, top_k and
# dynamic_decode are TensorFlow library code (oracle/ref_shim/tf1). Same dims, source batch and variable seed as
# nmt_full.
# ---------------------------------------------------------------------------------------------------------
NMT_BEAM_CASES = {
    # matrices scaled by `gain`; END among the symbols this random model emits. A random recurrent model is an
    # ill-conditioned beam-search problem (near-ties at the beam boundary at most steps): the cases are the ones, of
    # a scan over (gain, END, penalty), where most rows' winners survive perturbations of bf16 size — see `stable`
    "lp0_beam4": dict(beam=4, lp=0.0, END=4, att="gnmt_v2", gain=2.5),     # two rows finish (lengths 4 and 1), two run to the cap
    "lp06_beam4": dict(beam=4, lp=0.6, END=18, att="gnmt_v2", gain=8.0),   # winners of lengths 2, 3, 4, 13
    "lp1_beam3_gnmt": dict(beam=3, lp=1.0, END=18, att="gnmt", gain=8.0),
}
NMT_BEAM_PERTURBATIONS = 6


def nmt_beam_variable(name, shape, seed, gain, perturbation=None):
  """The variable values of the beam-search fixture; perturbation k: every matrix entry times 1 + 2^-7 u, u ~ U(-1, 1)
  (four times the rounding error of a bf16 weight)."""
  import zlib
  a = seeded_array(name, shape, seed)
  if a.ndim == 2:
    a = a * np.float32(gain)
    if perturbation is not None:
      rs = np.random.RandomState((zlib.crc32(name.encode()) + 1000 * perturbation) % (2 ** 31))
      a = a * (1 + np.float32(2.0 ** -7) * rs.uniform(-1, 1, a.shape).astype(np.float32))
  return a


def nmt_beam(seed=67):
  D = NMT_FULL
  B, S, T, V, E, H, U, NL = [D[k] for k in ("B", "S", "T", "V", "E", "H", "U", "layers")]
  rng = np.random.RandomState(seed)
  src_len = np.array([11, 6, 9, 3], np.int32)
  src = rng.randint(4, V, size=(B, S)).astype(np.int32)
  for b in range(B):
    src[b, src_len[b]:] = 0
  out = {"src": src, "src_len": src_len, "seed": np.int32(seed)}
  for case, cfg in NMT_BEAM_CASES.items():
    tf, imp = _install()
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    Enc = imp("open_seq2seq.encoders.rnn_encoders").BidirectionalRNNEncoderWithEmbedding
    Dec = imp("open_seq2seq.decoders.rnn_decoders").BeamSearchRNNDecoderWithAttention
    cellp = {"num_units": H, "forget_bias": 1.0}
    with tf.variable_scope("ForwardPass"):
      enc = Enc(dict(src_vocab_size=V, src_emb_size=E, core_cell=tf.nn.rnn_cell.LSTMCell, core_cell_params=cellp,
                     encoder_layers=NL, encoder_use_skip_connections=False, encoder_dp_input_keep_prob=1.0,
                     encoder_dp_output_keep_prob=1.0, dtype=tf.float32), None, mode="infer")
      dec = Dec(dict(GO_SYMBOL=2, END_SYMBOL=cfg["END"], tgt_vocab_size=V, tgt_emb_size=E, attention_layer_size=U,
                     attention_type=cfg["att"], core_cell=tf.nn.rnn_cell.LSTMCell, core_cell_params=dict(cellp),
                     decoder_layers=NL, decoder_use_skip_connections=False, batch_size=B,
                     decoder_dp_input_keep_prob=1.0, decoder_dp_output_keep_prob=1.0, dtype=tf.float32,
                     beam_width=cfg["beam"], length_penalty=cfg["lp"]), None, mode="infer")
      eo = enc.encode({"source_tensors": [tf.constant(src), tf.constant(src_len)]})
      do = dec.decode({"encoder_output": eo})
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    st = do["final_state"]
    with tf.Session() as sess:
      for n, v in zip(names, tvars):
        v.load(nmt_beam_variable(n, tuple(v._var.shape), seed, cfg["gain"]))
      vals = sess.run({"top": do["logits"], "lengths": st.lengths, "log_probs": st.log_probs, "finished": st.finished,
                       "seq_len": do["final_sequence_lengths"]})
      # which rows' winners are properties of the model rather than of the last bit: the reference's own answer under
      # perturbations of bf16 size (the device computes in bf16; it is held to the `stable` rows exactly)
      stable = np.ones(B, np.bool_)
      for k in range(NMT_BEAM_PERTURBATIONS):
        for n, v in zip(names, tvars):
          v.load(nmt_beam_variable(n, tuple(v._var.shape), seed, cfg["gain"], perturbation=k))
        top_k = sess.run(do["logits"])
        stable &= np.array([top_k.shape == vals["top"].shape and np.array_equal(top_k[b], vals["top"][b])
                            for b in range(B)])
    out.update({case + "/top_ids": vals["top"].astype(np.int32), case + "/lengths": vals["lengths"].astype(np.int32),
                case + "/log_probs": vals["log_probs"].astype(np.float32),
                case + "/finished": vals["finished"].astype(np.bool_),
                case + "/final_sequence_lengths": vals["seq_len"].astype(np.int32), case + "/stable": stable,
                case + "/var_names": np.array(names)})
    for n, v in zip(names, tvars):
      out["%s/shape/%s" % (case, n)] = np.array(tuple(v._var.shape), np.int32)
  return out


# ---------------------------------------------------------------------------------------------------------
# The whole Tacotron 2 model at widths the HIP kernels take: Tacotron2Encoder (no style tokens) -> Tacotron2Decoder
# ("both" mode, attention depth 128, attention bias) -> Text2SpeechLoss, train mode, the pre-net's dropout passed
# through (tf1.DROPOUT_OFF: the device test runs with its pre-net dropout off as well). Variables from seeded_array,
# gradients as (norm, projection).
# ---------------------------------------------------------------------------------------------------------
TACO_FULL = dict(B=3, S=12, T=24, V=40, E=64, Henc=32, H=64, NM=16, NG=24, pre=64, C=64, U=128,
                 text_len=[12, 7, 10], spec_len=[24, 16, 8])


def tacotron_full(seed=83):
  D = TACO_FULL
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  tf.DROPOUT_OFF = True
  try:
    Enc = imp("open_seq2seq.encoders.tacotron2_encoder").Tacotron2Encoder
    Dec = imp("open_seq2seq.decoders.tacotron2_decoder").Tacotron2Decoder
    Loss = imp("open_seq2seq.losses.text2speech_loss").Text2SpeechLoss
    rng = np.random.RandomState(seed)
    B, S, T, V, E, NM, NG = [D[k] for k in ("B", "S", "T", "V", "E", "NM", "NG")]
    text_len, spec_len = np.array(D["text_len"], np.int32), np.array(D["spec_len"], np.int32)
    text = rng.randint(3, V, size=(B, S)).astype(np.int32)
    spec = np.concatenate([rng.standard_normal((B, T, NM)) - 1.0, np.exp(rng.standard_normal((B, T, NG)) - 2.0)],
                          -1).astype(np.float32)
    spec = _np(__import__("torch").from_numpy(spec).to(__import__("torch").bfloat16).float())   # bf16 teacher frames
    stop = (np.arange(T)[None, :] >= (spec_len[:, None] - 2)).astype(np.float32)

    class _DL(object):
      params = {"src_vocab_size": V, "num_audio_features": {"mel": NM, "magnitude": NG}, "output_type": "both"}
      _exp_mag = True

    class _Model(object):
      params = {"dtype": tf.float32}

      def get_data_layer(self):
        return _DL()

      def get_tf_dtype(self):
        return tf.float32
    conv = [{"kernel_size": [5], "stride": [1], "num_channels": D["C"], "padding": "SAME"} for _ in range(2)]
    post = [{"kernel_size": [5], "stride": [1], "num_channels": D["C"], "padding": "SAME", "activation_fn": tf.nn.tanh},
            {"kernel_size": [5], "stride": [1], "num_channels": D["C"], "padding": "SAME", "activation_fn": tf.nn.tanh},
            {"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]
    with tf.variable_scope("ForwardPass"):
      enc = Enc(dict(src_emb_size=E, conv_layers=conv, activation_fn=tf.nn.relu, num_rnn_layers=1,
                     rnn_cell_dim=D["Henc"], rnn_type=tf.contrib.cudnn_rnn.CudnnLSTM, use_cudnn_rnn=True,
                     rnn_unidirectional=False, cnn_dropout_prob=0.0, rnn_dropout_prob=0.0, zoneout_prob=0.0,
                     data_format="channels_last", dtype=tf.float32), _Model(), mode="train")
      dec = Dec(dict(attention_layer_size=D["U"], attention_type="location", attention_bias=True,
                     decoder_cell_units=D["H"], decoder_cell_type=tf.nn.rnn_cell.LSTMCell, decoder_layers=2,
                     enable_prenet=True, prenet_layers=2, prenet_units=D["pre"], enable_postnet=True,
                     postnet_conv_layers=post, postnet_keep_dropout_prob=1.0, postnet_bn_momentum=0.1,
                     postnet_bn_epsilon=1e-5, mask_decoder_sequence=True, zoneout_prob=0.0, dropout_prob=0.0,
                     dtype=tf.float32), _Model(), mode="train")
      tgt = [tf.constant(spec), tf.constant(stop), tf.constant(spec_len)]
      eo = enc.encode({"source_tensors": [tf.constant(text), tf.constant(text_len)]})
      do = dec.decode({"encoder_output": eo, "target_tensors": tgt})
      loss = Loss(dict(use_mask=True), _Model()).compute_loss(
          {"decoder_output": do, "target_tensors": tgt})
    tvars = tf.trainable_variables()
    names = [v.name.split(":")[0] for v in tvars]
    mel, post_o, align, _, lens, mag = do["outputs"]
    with tf.Session() as sess:
      for n, v in zip(names, tvars):
        v.load(seeded_array(n, tuple(v._var.shape), seed))
      vals = sess.run({"enc": eo["outputs"], "mel": mel, "post": post_o, "stop": do["stop_token_prediction"],
                       "mag": mag, "align": align, "loss": loss, "grads": tf.gradients(loss, tvars)})
  finally:
    tf.DROPOUT_OFF = False
  out = {"text": text, "spec": spec, "stop_target": stop, "enc_out": vals["enc"].astype(np.float16),
         "mel": vals["mel"], "post": vals["post"], "stop": vals["stop"], "mag": vals["mag"],
         "align": vals["align"].astype(np.float16), "loss": np.float32(vals["loss"]), "var_names": np.array(names),
         "seed": np.int32(seed)}
  for n, v, g in zip(names, tvars, vals["grads"]):
    out["shape/" + n] = np.array(tuple(v._var.shape), np.int32)
    out["gproj/" + n] = projection(n, g, seed)
  return out


# ---------------------------------------------------------------------------------------------------------
# DeepSpeech2 at widths the HIP kernels take (the device test's scaled-down case: 32 features, the configuration's own
# conv2d kernels [11, 41] / [11, 21] with 32 channels, two bidirectional GRU-64 layers, dense 128) + the CTC decoder's
# dense layer, surrogate loss sum(logits * R). Variables from seeded_array, gradients as (norm, projection).
# ---------------------------------------------------------------------------------------------------------
DS2_FULL = dict(B=3, T=48, F=32, H=64, NH=128, layers=2, V=29, lens=[48, 35, 22])
DS2_FULL_CONV = [{"kernel_size": [11, 41], "stride": [2, 2], "num_channels": 32, "padding": "SAME"},
                 {"kernel_size": [11, 21], "stride": [1, 2], "num_channels": 32, "padding": "SAME"}]


def ds2_full(seed=89):
  D = DS2_FULL
  tf, imp = _install()
  tf.reset_default_graph()
  tf.set_random_seed(seed)
  Enc = imp("open_seq2seq.encoders.ds2_encoder").DeepSpeech2Encoder
  Dec = imp("open_seq2seq.decoders.fc_decoders").FullyConnectedCTCDecoder
  rng = np.random.RandomState(seed)
  B, T, F, H, NH, V = [D[k] for k in ("B", "T", "F", "H", "NH", "V")]
  src_len = np.array(D["lens"], np.int32)
  x = tdnn_input(seed, src_len, T, F)
  with tf.variable_scope("ForwardPass"):
    enc = Enc(dict(dropout_keep_prob=1.0, conv_layers=DS2_FULL_CONV, activation_fn=tf.nn.relu,
                   num_rnn_layers=D["layers"], row_conv=False, n_hidden=NH, use_cudnn_rnn=True, rnn_cell_dim=H,
                   rnn_type="cudnn_gru", rnn_unidirectional=False, bn_momentum=0.99, bn_epsilon=1e-3,
                   dtype=tf.float32), None, name="ds2_encoder", mode="train")
    eo = enc.encode({"source_tensors": [tf.constant(x), tf.constant(src_len)]})
    do = Dec(dict(tgt_vocab_size=V, dtype=tf.float32), None, mode="train").decode({"encoder_output": eo})
  logits = do["logits"]
  R = rng.standard_normal(tuple(int(v) for v in logits.get_shape())).astype(np.float32)
  loss = tf.reduce_sum(logits * tf.constant(R))
  tvars = tf.trainable_variables()
  names = [v.name.split(":")[0] for v in tvars]
  with tf.Session() as sess:
    for n, v in zip(names, tvars):
      v.load(seeded_array(n, tuple(v._var.shape), seed))
    vals = sess.run({"out": eo["outputs"], "len": eo["src_length"], "logits": logits, "grads": tf.gradients(loss, tvars)})
  out = {"src_len": src_len, "out": vals["out"].astype(np.float16), "out_len": vals["len"].astype(np.int32),
         "logits": vals["logits"], "R": R, "var_names": np.array(names), "seed": np.int32(seed)}
  for n, v, g in zip(names, tvars, vals["grads"]):
    out["shape/" + n] = np.array(tuple(v._var.shape), np.int32)
    out["gproj/" + n] = projection(n, g, seed)
  return out


# ---------------------------------------------------------------------------------------------------------
# ASR front end: get_speech_features (data/speech2text/speech_utils.py:274-535) executed from the reference's file with
# stand-ins for librosa / python_speech_features (oracle/ref_shim/audio_libs): the librosa 'logfbank' path of the
# Jasper configs (64 mel bands, n_fft 512, per-feature normalisation, dither with a seeded np.random), the librosa
# 'spectrogram' path, and the psf 'spectrogram' (DeepSpeech2 configs, padded to a multiple of 8 frames) and 'logfbank'
# (the toy Wave2Letter config) paths.
# ---------------------------------------------------------------------------------------------------------
FRONTEND_CASES = {
    "jasper_logfbank": dict(backend="librosa", input_type="logfbank", num_audio_features=64, window_size=20e-3,
                            window_stride=10e-3, window="hanning", dither=1e-5, num_fft=512, norm_per_feature=True),
    "librosa_spectrogram": dict(backend="librosa", input_type="spectrogram", num_audio_features=96, window="hanning"),
    "ds2_psf_spectrogram": dict(input_type="spectrogram", num_audio_features=160, pad_to=8),
    "w2l_psf_logfbank": dict(backend="psf", input_type="logfbank", num_audio_features=40, pad_to=8),
}


TTS_CASES = {
    "tts_both_power2": dict(num_features={"mel": 80, "magnitude": 401}, n_fft=800, mag_power=2,
                            data_min={"mel": 1e-5, "magnitude": 1e-5}),
    "tts_both_power1": dict(num_features={"mel": 40, "magnitude": 257}, n_fft=512, mag_power=1,
                            data_min={"mel": 1e-2, "magnitude": 1e-3}),
}


def frontend_signal(seed=97, n=7013):
  rs = np.random.RandomState(seed)
  t = np.arange(n) / 16000.0
  sig = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t * (1 + 0.2 * t)) + \
      0.05 * rs.standard_normal(n)
  return (sig * 20000).astype(np.int16)


def frontend():
  sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
  import audio_libs
  audio_libs.install()
  _install()
  import importlib
  for k in [k for k in sys.modules if k.startswith("open_seq2seq.data.speech2text.speech_utils")]:
    del sys.modules[k]
  su = importlib.import_module("open_seq2seq.data.speech2text.speech_utils")
  assert "librosa" in su.BACKENDS
  sig = frontend_signal()
  out = {"signal": sig}
  # the TTS features of the Tacotron configs (data/text2speech/speech_utils.py:98-182: periodic-Hann STFT at hop
  # n_fft / 4, |D| ^ power, log(clip(., data_min)), HTK mel basis without normalisation, "both" = [mel, magnitude])
  for k in [k for k in sys.modules if k.startswith("open_seq2seq.data.text2speech")]:
    del sys.modules[k]
  pkg = types.ModuleType("open_seq2seq.data.text2speech")
  pkg.__path__ = [os.path.join(PKG, "data", "text2speech")]
  sys.modules["open_seq2seq.data.text2speech"] = pkg
  tts = importlib.import_module("open_seq2seq.data.text2speech.speech_utils")
  fsig = sig.astype(np.float32) / 32768.0
  for case, kw in TTS_CASES.items():
    mel_f, mag_f = tts.get_speech_features(fsig, 22050, dict(kw["num_features"]), "both", kw["n_fft"], None,
                                           kw["mag_power"], False, 0., 1., dict(kw["data_min"]))
    out[case + "/mel"], out[case + "/mag"] = np.asarray(mel_f, np.float32), np.asarray(mag_f, np.float32)
  for case, params in FRONTEND_CASES.items():
    np.random.seed(1234)                     # the dither draw of the librosa path
    feats, dur = su.get_speech_features(sig.copy(), 16000, dict(params))
    out[case + "/features"] = np.asarray(feats, np.float32)
    out[case + "/duration"] = np.float64(dur)
  return out


GENERATORS = {"transformer": transformer, "transformer_d512": transformer_d512, "tdnn": tdnn,
              "tdnn_wide": tdnn_wide, "optim": optim, "train_op": train_op, "ds2": ds2, "nmt_decoder": nmt_decoder, "nmt_encoder": nmt_encoder, "nmt_encoder_dev": nmt_encoder_dev, "tacotron_decoder": tacotron_decoder, "t2s_loss": t2s_loss, "tacotron_infer": tacotron_infer, "tacotron_infer_dev": tacotron_infer_dev, "tacotron_encoder": tacotron_encoder, "beam_search": beam_search, "transformer_infer": transformer_infer, "nmt_full": nmt_full, "tacotron_full": tacotron_full, "ds2_full": ds2_full, "frontend": frontend, "nmt_beam": nmt_beam, "transformer_infer_d512": transformer_infer_d512}


def generate(name):
  out = GENERATORS[name]()
  return {k: np.asarray(v) for k, v in out.items()}


def fixture_path(name):
  return os.path.join(HERE, "ref_exec_%s.npz" % name)


def compare(a, b, rtol=2e-5, atol=2e-6):
  """Names that differ between two fixture dicts (float arrays within rtol / atol, everything else exact)."""
  bad = [k for k in set(a) ^ set(b)]
  for k in set(a) & set(b):
    x, y = np.asarray(a[k]), np.asarray(b[k])
    if x.shape != y.shape:
      bad.append(k)
    elif x.dtype.kind == "f":
      if not np.allclose(x, y, rtol=rtol, atol=atol * max(1.0, float(np.abs(y).max()) if y.size else 1.0)):
        bad.append(k)
    elif not np.array_equal(x, y):
      bad.append(k)
  return sorted(bad)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("names", nargs="*", default=sorted(GENERATORS))
  ap.add_argument("--check", action="store_true", help="regenerate and compare with the committed files")
  args = ap.parse_args()
  if not reference_available():
    raise SystemExit("%s not found: fixtures can only be generated where the reference checkout is" % PKG)
  rc = 0
  for n in args.names:
    out = generate(n)
    if args.check:
      bad = compare(out, dict(np.load(fixture_path(n))))
      print("%s: %s" % (n, "reproduced" if not bad else "DIFFERS in %s" % bad))
      rc |= bool(bad)
    else:
      np.savez_compressed(fixture_path(n), **out)
      print("%s: %d arrays, %.1f KB -> %s" % (n, len(out), os.path.getsize(fixture_path(n)) / 1e3,
                                             os.path.relpath(fixture_path(n), REPO)))
  return rc


if __name__ == "__main__":
  sys.exit(main())
