"""ORACLE (test infrastructure only): CPU fp32 restatement (PyTorch, autograd for
reference gradients) of the Transformer NMT path on PADDED [B, L] batches, exactly
as the reference computes it:
  parts/transformer/utils.py:28-129   position encoding, padding, biases
  parts/transformer/embedding_layer.py:59-105  shared embedding / tied softmax
  parts/transformer/common.py:41-106  LayerNormalization (L2), PrePostProcessingWrapper
  parts/transformer/attention_layer.py:104-227  Attention / SelfAttention ("loung")
  parts/transformer/ffn_layer.py:51-85  FeedFowardNetwork (padding removal = no-op on
                                        the non-pad rows)
  encoders/transformer_encoder.py:78-170, decoders/transformer_decoder.py:155-230
  losses/sequence_loss.py:257-309      PaddedCrossEntropyLossWithSmoothing
Known-answer pins from the reference's own tests (parts/transformer/utils_test.py:27-61)
are reproduced in tests/test_oracle_transformer.py. PARITY STATUS (round 5): pinned to the
reference's OWN CODE — TransformerEncoder / TransformerDecoder.decode_pass /
PaddedCrossEntropyLossWithSmoothing executed from their files on the TF-primitive stand-in
oracle/ref_shim/tf1 (tests/golden/make_ref_exec.py); this module reproduces their encoder
output, logits, loss (1e-5) and all 65 variable gradients (1e-6) on the same inputs and
variables (tests/test_ref_exec_transformer.py). The primitives under the reference (matmul,
softmax, Dense ...) are restated there: "pinned modulo TF primitives".
Dropout masks are passed in explicitly (keep masks), never drawn here.
"""
import math

import torch
import torch.nn.functional as F

NEG_INF = -1e9


def get_position_encoding(length, hidden_size, min_timescale=1.0, max_timescale=1.0e4):
  position = torch.arange(length, dtype=torch.float32)
  num_timescales = hidden_size // 2
  log_inc = math.log(float(max_timescale) / float(min_timescale)) / float(num_timescales - 1)
  inv = min_timescale * torch.exp(torch.arange(num_timescales, dtype=torch.float32) * -log_inc)
  scaled = position[:, None] * inv[None, :]
  return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)


def get_padding(x, padding_value=0):
  return (x == padding_value).float()


def get_padding_bias(x, pad_sym=0):
  return (get_padding(x, pad_sym) * NEG_INF)[:, None, None, :]


def get_decoder_self_attention_bias(length):
  valid = torch.tril(torch.ones(length, length))
  return (NEG_INF * (1.0 - valid))[None, None]


def embedding(ids, table, embed_scale=True, pad_sym=0):
  V, D = table.shape
  ids = torch.where(ids > V - 1, torch.full_like(ids, pad_sym), ids)
  emb = table[ids.long()]
  if embed_scale:
    emb = emb * D ** 0.5
  return emb * (1.0 - get_padding(ids, pad_sym))[..., None]


def layer_norm(x, scale, bias, eps=1e-6):
  mean = x.mean(-1, keepdim=True)
  var = ((x - mean) ** 2).mean(-1, keepdim=True)
  return (x - mean) * torch.rsqrt(var + eps) * scale + bias


def _drop(x, mask, keep):
  return x if mask is None else x * mask.float() / keep


def attention(x, y, w, bias, num_heads, att_mask=None, att_keep=1.0):
  """w: dict q,k,v,o each [in, out] (tf.layers.Dense kernel layout, no bias)."""
  B, Lx, D = x.shape
  Ly = y.shape[1]
  dh = D // num_heads
  q = (x @ w["q"]).view(B, Lx, num_heads, dh).transpose(1, 2) * dh ** -0.5
  k = (y @ w["k"]).view(B, Ly, num_heads, dh).transpose(1, 2)
  v = (y @ w["v"]).view(B, Ly, num_heads, dh).transpose(1, 2)
  logits = q @ k.transpose(-1, -2) + bias
  weights = torch.softmax(logits, -1)
  weights = _drop(weights, att_mask, att_keep)
  out = (weights @ v).transpose(1, 2).reshape(B, Lx, D)
  return out @ w["o"]


def ffn(x, w, relu_mask=None, relu_keep=1.0):
  h = torch.relu(x @ w["w1"] + w["b1"])
  h = _drop(h, relu_mask, relu_keep)
  return h @ w["w2"] + w["b2"]


def prepost(x, ln, fn, post_mask=None, post_keep=1.0):
  y = fn(layer_norm(x, ln["scale"], ln["bias"]))
  return x + _drop(y, post_mask, post_keep)


def encoder(src_ids, P, num_heads, masks=None, keeps=None):
  """P: dict with 'emb' [V,D], 'layers': [{ 'ln1','att','ln2','ffn' }], 'ln_out'.
  masks: optional dict of keep masks keyed like the device RNG streams."""
  m = masks or {}
  kp = keeps or {}
  D = P["emb"].shape[1]
  x = embedding(src_ids, P["emb"]) + get_position_encoding(src_ids.shape[1], D)
  x = _drop(x, m.get("emb"), kp.get("post", 1.0))
  bias = get_padding_bias(src_ids)
  for i, L in enumerate(P["layers"]):
    x = prepost(x, L["ln1"], lambda t: attention(t, t, L["att"], bias, num_heads,
                                                 m.get(("att", i)), kp.get("att", 1.0)),
                m.get(("post1", i)), kp.get("post", 1.0))
    x = prepost(x, L["ln2"], lambda t: ffn(t, L["ffn"], m.get(("relu", i)), kp.get("relu", 1.0)),
                m.get(("post2", i)), kp.get("post", 1.0))
  return layer_norm(x, P["ln_out"]["scale"], P["ln_out"]["bias"]), bias


def decoder_pass(tgt_ids, enc_out, enc_bias, P, num_heads, masks=None, keeps=None):
  m = masks or {}
  kp = keeps or {}
  emb = P["emb"]
  D = emb.shape[1]
  x = embedding(tgt_ids, emb)
  x = F.pad(x, (0, 0, 1, 0))[:, :-1, :]          # shift right (transformer_decoder.py:199-202)
  L = x.shape[1]
  x = x + get_position_encoding(L, D)
  x = _drop(x, m.get("emb"), kp.get("post", 1.0))
  self_bias = get_decoder_self_attention_bias(L)
  for i, Lyr in enumerate(P["layers"]):
    x = prepost(x, Lyr["ln1"], lambda t: attention(t, t, Lyr["self"], self_bias, num_heads,
                                                   m.get(("self", i)), kp.get("att", 1.0)),
                m.get(("post1", i)), kp.get("post", 1.0))
    x = prepost(x, Lyr["ln2"], lambda t: attention(t, enc_out, Lyr["cross"], enc_bias, num_heads,
                                                   m.get(("cross", i)), kp.get("att", 1.0)),
                m.get(("post2", i)), kp.get("post", 1.0))
    x = prepost(x, Lyr["ln3"], lambda t: ffn(t, Lyr["ffn"], m.get(("relu", i)), kp.get("relu", 1.0)),
                m.get(("post3", i)), kp.get("post", 1.0))
  x = layer_norm(x, P["ln_out"]["scale"], P["ln_out"]["bias"])
  return x @ emb.t()                              # tied softmax (embedding_layer.py:90-105)


def padded_xent_smoothing(logits, labels, smoothing):
  """sequence_loss.py:257-309 (logits [B,L,V], labels [B,L] of the same L)."""
  V = logits.shape[-1]
  conf = 1.0 - smoothing
  low = (1.0 - conf) / float(V - 1)
  soft = torch.full_like(logits, low)
  soft.scatter_(-1, labels.long()[..., None], conf)
  xent = -(soft * torch.log_softmax(logits, -1)).sum(-1)
  norm = -(conf * math.log(conf) + float(V - 1) * low * math.log(low + 1e-20))
  xent = xent - norm
  w = (labels != 0).float()
  return (xent * w).sum() / w.sum()
