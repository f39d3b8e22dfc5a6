"""ORACLE (test infrastructure only): CPU fp32 restatement of DeepSpeech2Encoder._encode
(open_seq2seq/encoders/ds2_encoder.py:158-401) for the cuDNN-GRU configuration
(example_configs/speech2text/ds2_large_8gpus.py:53-72): conv2d(SAME, TF asymmetric
padding) + BN + ReLU, bidirectional multi-layer cuDNN-form GRU without sequence lengths,
dense + ReLU (+ dropout mask). PARITY STATUS (round 5): the encoder's WIRING is pinned to the
reference's own code — DeepSpeech2Encoder._encode executed from its file on the TF-primitive stand-in
oracle/ref_shim/tf1 (bidirectional and unidirectional + row_conv cases): outputs 1e-5, all gradients 1e-6
(tests/test_ref_exec_ds2.py). The cuDNN GRU cell itself is a TensorFlow library object (torch.nn.GRU on
both sides): its equations stay "parity unpinned" (cross-checked in tests/test_oracle_rnn.py)."""
import torch
import torch.nn.functional as F

from . import cnn


def conv2d_tf(x, w_tf, stride, padding="SAME"):
  """x [B,T,Fr,Cin]; w_tf [KT,KF,Cin,Cout]; stride [sT,sF]."""
  xc = x.permute(0, 3, 1, 2)
  if padding == "SAME":
    _, pt_l, pt_r = cnn.same_pad(x.shape[1], w_tf.shape[0], stride[0], 1)
    _, pf_l, pf_r = cnn.same_pad(x.shape[2], w_tf.shape[1], stride[1], 1)
    xc = F.pad(xc, (pf_l, pf_r, pt_l, pt_r))
  y = F.conv2d(xc, w_tf.permute(3, 2, 0, 1).contiguous(), stride=tuple(stride))
  return y.permute(0, 2, 3, 1).contiguous()


def ds2_encode(x, conv_layers, W, gru, fc_w, fc_b, bn_eps=1e-3, keep_mask=None, keep=1.0):
  """x [B,T,F]; W: dict 'convN/kernel', 'convN/bn/gamma', 'convN/bn/beta';
  gru: torch.nn.GRU (batch_first, configured by the caller); fc_w [in,out]."""
  h = x[..., None]
  for i, cl in enumerate(conv_layers):
    n = "conv%d" % (i + 1)
    y = conv2d_tf(h, W[n + "/kernel"], cl["stride"], cl["padding"])
    B, T, Fr, C = y.shape
    yn = cnn.batch_norm_train(y.reshape(B, T * Fr, C), W[n + "/bn/gamma"], W[n + "/bn/beta"], bn_eps)[0]
    h = torch.relu(yn).reshape(B, T, Fr, C)
  B, T, Fr, C = h.shape
  r = h.reshape(B, T, Fr * C)
  if gru is not None:
    r, _ = gru(r)
  o = torch.relu(r @ fc_w + fc_b)
  if keep_mask is not None:
    o = o * keep_mask.float() / keep
  return o


def row_conv(x, w, gamma, beta, eps=1e-3, act=torch.relu):
  """row_conv (ds2_encoder.py:38-83): depthwise conv over time, filter w [K, C] (TF [K,1,C,1]),
  SAME padding (left (K-1)//2), batch norm with batch statistics, activation. x [B,T,C]."""
  B, T, C = x.shape
  K = w.shape[0]
  _, pl, pr = cnn.same_pad(T, K, 1, 1)
  xp = F.pad(x.transpose(1, 2), (pl, pr))
  y = F.conv1d(xp, w.t()[:, None, :], groups=C).transpose(1, 2)
  yn = cnn.batch_norm_train(y, gamma, beta, eps)[0]
  return act(yn)
