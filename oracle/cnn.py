"""ORACLE (test infrastructure only): CPU fp32 restatement of the conv / BatchNorm
building blocks of open_seq2seq/parts/cnns/conv_blocks.py.

PARITY STATUS: the reference ships no test that pins conv / BN *values*
("parity unpinned"); TensorFlow is not installable here. The
restatement follows the TF-1.13 semantics the reference calls into and is
cross-checked against an independent direct-loop NumPy implementation
(tests/test_oracle_cnn.py). Round 5: through oracle/tdnn.py these functions reproduce the
reference's own conv_bn_actv / conv_bn_res_bn_actv, executed on a TF-primitive stand-in
(tests/test_ref_exec_tdnn.py) — the padding and the BatchNorm conventions ("SAME" with the
extra element at the end, biased variance to normalise, Bessel-corrected variance in the
moving average of the 4-D fused path) are the stand-in's restatement of TF 1.13, written
independently of this file.
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


def sep_conv1d_tf(x, depthwise, pointwise, stride=1, dil=1, padding="SAME", between=None):
  """tf.layers.separable_conv1d(use_bias=False, depth_multiplier=1) (layer type
  'sep_conv1d', conv_blocks.py:11-16): depthwise [K, Cin] (TF depthwise_kernel [K, Cin, 1])
  applied per channel with the block's stride / dilation / padding, then the pointwise
  kernel [1, Cin, Cout]. `between`: optional function applied to the depthwise output before the pointwise
  product (identity in exact arithmetic) — the device stores that tensor, and the gradient flowing back
  through it, in bf16; oracle/tdnn.py passes its storage-rounding node here when it emulates the device's
  storage points."""
  x = torch.as_tensor(x, dtype=torch.float32)
  B, T, C = x.shape
  K = depthwise.shape[0]
  xc = x.permute(0, 2, 1)
  if padding == "SAME":
    _, pl, pr = same_pad(T, K, stride, dil)
    xc = F.pad(xc, (pl, pr))
  z = F.conv1d(xc, depthwise.t()[:, None, :].contiguous(), stride=stride, dilation=dil, groups=C)
  if between is not None:
    z = between(z)
  y = F.conv1d(z, pointwise.permute(2, 1, 0).contiguous())
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


# ---------------------------------------------------------------------------
# BatchNorm / conv blocks (conv_blocks.py:61-232)
# ---------------------------------------------------------------------------
def batch_norm_train(y, gamma, beta, eps, momentum=None, moving_mean=None,
                     moving_var=None):
  """tf.layers.batch_normalization(training=True) on [B,T,1,C] (fused path,
  conv_blocks.py:208-227): statistics over ALL B*T positions (padded frames
  included), biased variance for normalisation, Bessel-corrected variance in the
  moving average, moving = moving*momentum + batch*(1-momentum).
  Returns (out, mean, var_biased, new_moving_mean, new_moving_var)."""
  y = torch.as_tensor(y, dtype=torch.float32)
  C = y.shape[-1]
  flat = y.reshape(-1, C).double()
  n = flat.shape[0]
  mean = flat.mean(0)
  var = flat.var(0, unbiased=False)
  out = ((flat - mean) * torch.rsqrt(var + eps)).float() * gamma + beta
  new_mm = new_mv = None
  if moving_mean is not None:
    unb = var * n / max(n - 1, 1)
    new_mm = moving_mean * momentum + mean.float() * (1 - momentum)
    new_mv = moving_var * momentum + unb.float() * (1 - momentum)
  return out.reshape(y.shape), mean.float(), var.float(), new_mm, new_mv


def batch_norm_eval(y, gamma, beta, eps, moving_mean, moving_var):
  return (y - moving_mean) * torch.rsqrt(moving_var + eps) * gamma + beta


def layer_norm_tf(y, gamma, beta, eps=1e-12):
  """tf.contrib.layers.layer_norm(inputs=conv) as conv_ln_actv calls it (conv_blocks.py:262-264; TensorFlow 1.x
  contrib, a dependency that is not vendored in the reference — "parity unpinned" against TensorFlow itself; its
  published algorithm, tensorflow/contrib/layers/python/layers/layers.py `layer_norm`): begin_norm_axis = 1 —
  moments over ALL axes but the batch axis, here the T x C values of a sample (padded frames included) —,
  begin_params_axis = -1 — gamma / beta of shape [C] —, tf.nn.batch_normalization with variance_epsilon 1e-12.
  y [B, T, C] fp32 (differentiable)."""
  mean = y.mean(dim=(1, 2), keepdim=True)
  var = ((y - mean) ** 2).mean(dim=(1, 2), keepdim=True)
  return (y - mean) * torch.rsqrt(var + eps) * gamma + beta


def instance_norm_tf(y, gamma, beta, eps=1e-6):
  """tf.contrib.layers.instance_norm(inputs=conv, data_format="NHWC") as conv_in_actv calls it
  (conv_blocks.py:298-301; same unvendored dependency): moments per sample and channel over the remaining
  (time) axis of the padded tensor, gamma / beta of shape [C], epsilon 1e-6. y [B, T, C] fp32."""
  mean = y.mean(dim=1, keepdim=True)
  var = ((y - mean) ** 2).mean(dim=1, keepdim=True)
  return (y - mean) * torch.rsqrt(var + eps) * gamma + beta


def act_fn(x, act):
  if act in (None, "none", 0):
    return x
  if act in ("relu", 1):
    return torch.relu(x)
  if act in ("tanh", 2):
    return torch.tanh(x)
  if act in ("relu20", 3):
    # the clipped ReLU of the reference's DeepSpeech2 / Wave2Letter configs:
    # activation_fn = lambda x: tf.minimum(tf.nn.relu(x), 20.0)  (example_configs/speech2text/ds2_toy_config.py:79)
    return torch.clamp(torch.relu(x), max=20.0)
  raise ValueError(act)


def seq_mask(lens, T):
  return (torch.arange(T)[None, :] < torch.as_tensor(lens)[:, None]).float()[:, :, None]


def bn_res_act(ys, gammas, betas, eps, act, keep_mask=None, keep_prob=1.0, out_len=None):
  """act(sum_j BN_j(y_j)) -> dropout -> (mask of the next conv's input).
  conv_bn_res_bn_actv (conv_blocks.py:61-168) + tf.nn.dropout (tdnn_encoder.py:255:
  x * mask / keep_prob) + conv_feats * mask (tdnn_encoder.py:185-186,204-205)."""
  tot = 0
  for y, g, b in zip(ys, gammas, betas):
    tot = tot + batch_norm_train(y, g, b, eps)[0]
  out = act_fn(tot, act)
  if keep_mask is not None:
    out = out * keep_mask.float() / keep_prob
  if out_len is not None:
    out = out * seq_mask(out_len, out.shape[1])
  return out
