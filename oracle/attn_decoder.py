"""ORACLE (test infrastructure only): CPU fp32 attention-RNN decoders with autograd.

Restates, step by step, what tf.contrib.seq2seq.dynamic_decode executes for
  * RNNDecoderWithAttention with attention_type gnmt / gnmt_v2
    (open_seq2seq/decoders/rnn_decoders.py:147-321; GNMTAttentionMultiCell
    parts/rnns/gnmt.py:32-79; AttentionWrapper.call parts/rnns/attention_wrapper.py:1720-1823;
    normalised Bahdanau score :482-539; memory preparation :83-156,250-257; score mask :158-170)
  * the Tacotron2 decoder cell: AttentionWrapper(MultiRNNCell[LSTMCell x L],
    LocationSensitiveAttention, output_attention="both")
    (decoders/tacotron2_decoder.py:257-420; attention_wrapper.py:641-715, 749-878).

Parameter tensors use the DEVICE layout of openseq2seq_amd (weights [out, in], gate order
i, j, f, o as tf.nn.rnn_cell.LSTMCell, forget_bias inside the sigmoid); dropout masks are
passed in explicitly (already scaled by 1/keep) so GPU and oracle share them.

PARITY STATUS (round 5): pinned to the reference's OWN CODE — parts/rnns/attention_wrapper.py
(AttentionWrapper, BahdanauAttention normalised, LocationSensitiveAttention with the Chorowski
location layer, memory preparation, score masking), parts/rnns/gnmt.py, decoders/rnn_decoders.py
and decoders/tacotron2_decoder.py executed from their files on the TF-primitive stand-in
oracle/ref_shim/tf1 (its rnn.py restates the LSTM cell class and dynamic_decode as one traced step);
through oracle/nmt.py and oracle/tacotron.py this loop reproduces their outputs (1e-5) and all
gradients (3e-6): tests/test_ref_exec_nmt.py, tests/test_ref_exec_tacotron.py. The LSTM cell
equations themselves are TensorFlow library code (i, j, f, o; forget_bias inside the sigmoid):
restated on both sides."""
import math

import torch
import torch.nn.functional as F


def lstm_cell(x_cat, c, w, bias, forget_bias):
  """x_cat [B, In+H] = concat(inputs, h_prev); w [4H, In+H]; gate order i, j, f, o."""
  z = x_cat @ w.t()
  if bias is not None:
    z = z + bias
  i, j, f, o = z.chunk(4, dim=-1)
  cn = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
  return torch.tanh(cn) * torch.sigmoid(o), cn


def prepare_memory(memory, src_len):
  """_prepare_memory: values = memory zeroed past memory_sequence_length."""
  S = memory.shape[1]
  mask = (torch.arange(S)[None, :] < torch.as_tensor(src_len)[:, None]).to(memory.dtype)
  return memory * mask[:, :, None], mask


def bahdanau_score(q, keys, v, g=None, b=None):
  """q [B,U] processed query, keys [B,S,U]; normalised form when g is given."""
  if g is not None:
    nv = g * v * torch.rsqrt((v * v).sum())
    return (nv * torch.tanh(keys + q[:, None, :] + b)).sum(-1)
  return (v * torch.tanh(keys + q[:, None, :])).sum(-1)


def location_features(cum, conv_w, conv_b, dense_w):
  """ChorowskiLocationLayer: Conv1D(filters F, kernel K, SAME, bias) over the cumulative
  alignments [B,S,1], then a bias-free k=1 Conv1D to U. conv_w [K,F], dense_w [F,U]."""
  K = conv_w.shape[0]
  S = cum.shape[1]
  pad_total = K - 1
  pl = pad_total // 2
  x = F.pad(cum, (pl, pad_total - pl))                          # [B, S+K-1]
  win = x.unfold(1, K, 1)                                       # [B, S, K]
  feat = win @ conv_w + conv_b                                  # [B, S, F]
  return feat @ dense_w                                         # [B, S, U]


def masked_softmax(score, mask):
  score = score.masked_fill(mask == 0, float("-inf"))
  return torch.softmax(score, dim=-1)


def attention_decoder(p, gx0, memory, src_len, tgt_len=None, attn_in_mask=None, out_masks=None,
                      forget_bias=1.0, mode="bahdanau_norm", keys_override=None,
                      values_override=None):
  """Runs T steps. p: dict of parameters
      wcat: list of L tensors, wcat[0] [4H, M+H] (columns: attention, h), wcat[l>0] [4H, 2H]
      bias: list of L tensors [4H] or None (layer 0's bias is part of gx0)
      wq [U,H], wmem [U,M], v [U]; g [1], b [U] (bahdanau_norm / location with bias);
      conv_w [K,F], conv_b [F], dense_w [F,U] (location).
    gx0 [B,T,4H]: input projection of layer 0 (inputs @ W_in^T + bias), attention excluded.
    attn_in_mask [B,T,M]: dropout mask applied to the attention part of layer-0's input at
      step t (DropoutWrapper input dropout around the attention cell).
    out_masks: list of L [B,T,H] masks on each cell's OUTPUT (state h stays undropped).
    tgt_len: steps >= tgt_len[b] leave the state untouched and output zeros
      (TrainingHelper + impute_finished=True); None: every sample runs all T steps.
    Returns dict(y [B,T,H] top cell outputs, ctx [B,T,M], align [B,T,S])."""
  B, T, _ = gx0.shape
  L = len(p["wcat"])
  H = p["wq"].shape[1]
  values, mask = prepare_memory(memory, src_len)
  if values_override is not None:   # tests: gradients w.r.t. keys / values as separate leaves
    values = values_override
  keys = keys_override if keys_override is not None else values @ p["wmem"].t()
  M = values.shape[2]
  S = values.shape[1]
  h = [gx0.new_zeros(B, H) for _ in range(L)]
  c = [gx0.new_zeros(B, H) for _ in range(L)]
  attn = gx0.new_zeros(B, M)
  cum = gx0.new_zeros(B, S)
  ys, ctxs, aligns = [], [], []
  for t in range(T):
    live = None
    if tgt_len is not None:
      live = (t < torch.as_tensor(tgt_len)).to(gx0.dtype)[:, None]
    a_in = attn if attn_in_mask is None else attn * attn_in_mask[:, t]
    x = None
    nh, nc = [], []
    for l in range(L):
      if l == 0:
        # cell_inputs = concat(inputs, prev attention); the `inputs` part of the kernel
        # product is gx0[:, t] (computed for all steps at once by the caller)
        hn, cn = lstm_cell(torch.cat([a_in, h[0]], -1), c[0], p["wcat"][0], gx0[:, t], forget_bias)
      else:
        hn, cn = lstm_cell(torch.cat([x, h[l]], -1), c[l], p["wcat"][l], p["bias"][l], forget_bias)
      nh.append(hn)
      nc.append(cn)
      x = hn if out_masks is None else hn * out_masks[l][:, t]
    q = x @ p["wq"].t()
    if mode == "location":
      loc = location_features(cum, p["conv_w"], p["conv_b"], p["dense_w"])
      pre = keys + q[:, None, :] + loc
      if p.get("b") is not None:
        pre = pre + p["b"]
      score = (p["v"] * torch.tanh(pre)).sum(-1)
    elif mode == "bahdanau_norm":
      score = bahdanau_score(q, keys, p["v"], p["g"], p["b"])
    elif mode == "luong":
      # LuongAttention (attention_wrapper.py, _luong_score): score = keys . query, the query is
      # the cell output itself (no query layer; depth must equal num_units), scale=False
      score = (keys * x[:, None, :]).sum(-1)
    else:
      score = bahdanau_score(q, keys, p["v"])
    al = masked_softmax(score, mask)
    ctx = (al[:, :, None] * values).sum(1)
    if live is not None:
      for l in range(L):
        h[l] = live * nh[l] + (1 - live) * h[l]
        c[l] = live * nc[l] + (1 - live) * c[l]
      attn = live * ctx + (1 - live) * attn
      cum = cum + live * al
      ys.append(x * live)
      ctxs.append(ctx * live)
      aligns.append(al * live)
    else:
      h, c, attn = nh, nc, ctx
      cum = cum + al
      ys.append(x)
      ctxs.append(ctx)
      aligns.append(al)
  return dict(y=torch.stack(ys, 1), ctx=torch.stack(ctxs, 1), align=torch.stack(aligns, 1),
              keys=keys, values=values)
