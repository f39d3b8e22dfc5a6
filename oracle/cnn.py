"""ORACLE (test infrastructure only): CPU fp32 restatement of the conv / BatchNorm
building blocks of open_seq2seq/parts/cnns/conv_blocks.py.

PARITY STATUS: the reference ships no test that pins conv / BN *values*
("parity unpinned"); TensorFlow is not installable here. The
restatement follows the TF-1.13 semantics the reference calls into and is
cross-checked against an independent direct-loop NumPy implementation
(tests/test_oracle_cnn.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def same_pad(tin, k, stride, dil):
  """TF 'SAME' padding: (tout, pad_left, pad_right). TF pads asymmetrically when
  the total is odd — the extra element goes to the END (right). [TF-internal,
  not in /root/reference; relied upon by tf.layers.conv1d at conv_blocks.py:195-206]"""
  tout = -(-tin // stride)
  total = max((tout - 1) * stride + (k - 1) * dil + 1 - tin, 0)
  return tout, total // 2, total - total // 2


def conv1d_tf(x, w_tf, stride=1, dil=1, padding="SAME", mask_len=None):
  """tf.layers.conv1d(use_bias=False, data_format=channels_last)
  (conv_blocks.py:195-206). x [B,T,Cin] fp32, w_tf [K,Cin,Cout] (TF layout).
  mask_len: optional [B] lengths; rows t >= len are zeroed first
  (tdnn_encoder.py:138-143,185-186: conv_feats * mask)."""
  x = torch.as_tensor(x, dtype=torch.float32)
  w = torch.as_tensor(w_tf, dtype=torch.float32)
  B, T, Cin = x.shape
  K = w.shape[0]
  if mask_len is not None:
    m = (torch.arange(T)[None, :] < torch.as_tensor(mask_len)[:, None]).to(x.dtype)
    x = x * m[:, :, None]
  xc = x.permute(0, 2, 1)  # B C T
  if padding == "SAME":
    _, pl, pr = same_pad(T, K, stride, dil)
    xc = F.pad(xc, (pl, pr))
  y = F.conv1d(xc, w.permute(2, 1, 0).contiguous(), stride=stride, dilation=dil)
  return y.permute(0, 2, 1).contiguous()


def conv1d_direct_numpy(x, w_tf, stride=1, dil=1, padding="SAME"):
  """Independent direct-loop implementation (float64) used only to pin conv1d_tf."""
  x = np.asarray(x, np.float64)
  w = np.asarray(w_tf, np.float64)
  B, T, Cin = x.shape
  K, _, Cout = w.shape
  if padding == "SAME":
    tout, pl, _ = same_pad(T, K, stride, dil)
  else:
    tout, pl = (T - (K - 1) * dil - 1) // stride + 1, 0
  y = np.zeros((B, tout, Cout))
  for t in range(tout):
    for k in range(K):
      ti = t * stride + k * dil - pl
      if 0 <= ti < T:
        y[:, t, :] += x[:, ti, :] @ w[k]
  return y


def to_dev_layout(w_tf):
  """TF kernel [K,Cin,Cout] -> device layout [K,Cout,Cin]."""
  return torch.as_tensor(w_tf).permute(0, 2, 1).contiguous()
