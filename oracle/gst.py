"""ORACLE (test infrastructure only): CPU fp32 restatement of the global-style-token encoder,
Tacotron2Encoder._embed_style (open_seq2seq/encoders/tacotron2_encoder.py:341-505):
conv2d(3x3, stride 2, SAME) + BN + ReLU stack, tf.nn.rnn_cell.GRUCell under
dynamic_rnn(sequence_length) (final state), Dense(128, tanh), multi-head attention in
"bahdanau" mode over tanh(style tokens) (parts/transformer/attention_layer.py:104-196).
PARITY STATUS (round 5): pinned to the reference's OWN CODE — Tacotron2Encoder._embed_style executed from
its file on the TF-primitive stand-in oracle/ref_shim/tf1 (tf.nn.rnn_cell.GRUCell is TF library code,
restated there): style embedding and all gradients 7e-7 (tests/test_ref_exec_tacotron.py). The GRUCell
restatement is also cross-checked against a direct per-step formula in tests/test_oracle_gst.py."""
import torch

from . import cnn
from .ds2 import conv2d_tf


def gru_cell_tf(x, lens, wg, bg, wc, bc):
  """tf GRUCell: wg [In+H, 2H] (r | u), wc [In+H, H]; returns the state at each sample's last
  valid step [B, H] (dynamic_rnn copies the state through past sequence_length)."""
  B, T, In = x.shape
  H = wc.shape[1]
  h = x.new_zeros(B, H)
  for t in range(T):
    ru = torch.sigmoid(torch.cat([x[:, t], h], -1) @ wg + bg)
    r, u = ru[:, :H], ru[:, H:]
    c = torch.tanh(torch.cat([x[:, t], r * h], -1) @ wc + bc)
    hn = u * h + (1 - u) * c
    live = (t < torch.as_tensor(lens)).to(x.dtype)[:, None]
    h = live * hn + (1 - live) * h
  return h


def token_attention(ref, tokens, wq, wk, wv, wo, att_v, heads):
  """ref [B, 128]; tokens [N, E] (already tanh'd); dense kernels [in, out] without bias."""
  B = ref.shape[0]
  N = tokens.shape[0]
  q = (ref @ wq).view(B, 1, heads, -1).permute(0, 2, 1, 3)              # [B,h,1,d]
  k = (tokens @ wk).view(1, N, heads, -1).permute(0, 2, 1, 3)          # [1,h,N,d]
  v = (tokens @ wv).view(1, N, heads, -1).permute(0, 2, 1, 3)
  w = torch.tanh(att_v * torch.tanh(k + q)).sum(-1)                    # [B,h,N]
  w = torch.softmax(w, -1)
  out = (w[..., None] * v).sum(2)                                      # [B,h,d]
  return out.reshape(B, -1) @ wo


def style_encoder(P, spec, lens, conv_layers, heads, bn_eps=1e-5):
  """P: convs [(w_tf [KT,KF,Cin,Cout], gamma, beta)], wg, bg, wc, bc, ref_w [H,128], ref_b,
  tokens [N,E], wq, wk, wv, wo, att_v."""
  h = spec[..., None]
  lens = torch.as_tensor(lens)
  for (w, g, b), cl in zip(P["convs"], conv_layers):
    y = conv2d_tf(h, w, cl["stride"], cl["padding"])
    B, T, Fr, C = y.shape
    yn = cnn.batch_norm_train(y.reshape(B, T * Fr, C), g, b, bn_eps)[0]
    h = torch.relu(yn).reshape(B, T, Fr, C)
    s = cl["stride"][0]
    lens = (lens + s - 1) // s
  B, T, Fr, C = h.shape
  x = h.reshape(B, T, Fr * C)
  hf = gru_cell_tf(x, lens, P["wg"], P["bg"], P["wc"], P["bc"])
  ref = torch.tanh(hf @ P["ref_w"] + P["ref_b"])
  return token_attention(ref, torch.tanh(P["tokens"]), P["wq"], P["wk"], P["wv"], P["wo"], P["att_v"], heads)
