"""ORACLE (test infrastructure only): CPU fp32 restatement of the Jasper/TDNN ASR
forward path with torch autograd supplying reference gradients.

Follows open_seq2seq/encoders/tdnn_encoder.py:87-265 (mask before each conv,
length bookkeeping, dense residuals, dropout per layer, no mask on the final
output), parts/cnns/conv_blocks.py:61-232, decoders/fc_decoders.py:105-158 and
losses/ctc_loss.py:44-88. PARITY STATUS (round 5): pinned to the reference's OWN CODE —
TDNNEncoder._encode, conv_bn_actv / conv_bn_res_bn_actv, FullyConnectedCTCDecoder and the
CTCLoss wrapper executed from their files on the TF-primitive stand-in oracle/ref_shim/tf1
(tests/golden/make_ref_exec.py); this module reproduces their outputs, lengths, logits,
greedy ids (exact), BatchNorm moving statistics and all variable gradients (1e-6) on a
Jasper-shaped stack (tests/test_ref_exec_tdnn.py). The CTC recursion itself (tf.nn.ctc_loss,
TensorFlow-internal) stays pinned to torch.nn.functional.ctc_loss only.
"""
import torch

from . import cnn, ctc


class _Round(torch.autograd.Function):
  """Optional emulation of the device's bf16 STORAGE points (identity in exact
  arithmetic): rounds the forward value and/or the gradient flowing back through
  this point to bfloat16. Used only to separate 'bf16 storage noise' from logic
  errors when comparing against the HIP path; the fp32 oracle is emulate=False."""

  @staticmethod
  def forward(ctx, x, fwd, bwd):
    ctx.bwd = bwd
    return x.to(torch.bfloat16).float() if fwd else x

  @staticmethod
  def backward(ctx, g):
    return (g.to(torch.bfloat16).float() if ctx.bwd else g), None, None


def _r(x, emulate, fwd=True, bwd=True):
  return _Round.apply(x, fwd, bwd) if emulate else x


def tdnn_layer(feats, layer_res, blk, name, weights, out_mask=None, activation="relu", bn_eps=1e-3,
               keep_mask=None, keep_prob=1.0, emulate_bf16=False):
  """ONE conv_bn_actv / conv_bn_res_bn_actv call (parts/cnns/conv_blocks.py:61-232) + the
  encoder's dropout and output mask (tdnn_encoder.py:248-256): act(BN(conv(feats)) + sum_i
  BN_i(conv1x1_i(layer_res[i]))) -> dropout -> mask. `feats` / `layer_res` are already masked
  (the encoder masks every conv input); `layer_res` is empty unless this is the last repeat of a
  residual block; `name` = 'convIJ'. Also the unit of the layer-by-layer device tests, which feed
  both sides the same inputs."""
  K, s = blk["kernel_size"][0], blk["stride"][0]
  d = blk.get("dilation", [1])[0]
  dense = blk.get("residual_dense", False)
  em = emulate_bf16
  # conv outputs are stored in bf16; their gradients (BN backward) too
  sep = blk.get("type", "conv1d") == "sep_conv1d"
  if sep:
    y = _r(cnn.sep_conv1d_tf(feats, weights[name + "/depthwise_kernel"],
                             weights[name + "/pointwise_kernel"], s, d, blk["padding"],
                             between=(lambda z: _r(z, em))), em)
  else:
    y = _r(cnn.conv1d_tf(feats, weights[name + "/kernel"], s, d, blk["padding"]), em)
  tot = cnn.batch_norm_train(y, weights[name + "/bn/gamma"], weights[name + "/bn/beta"], bn_eps)[0]
  for i, r in enumerate(layer_res):
    rn = (name + "/res_%d" % i) if dense else (name + "/res")
    bn = (name + "/res_bn_%d" % i) if dense else (name + "/res_bn")
    if sep:   # residual branches use the block's layer type with k = 1 (conv_blocks.py:66,79-85)
      ry = _r(cnn.sep_conv1d_tf(r, weights[rn + "/depthwise_kernel"],
                                weights[rn + "/pointwise_kernel"], 1, 1, "SAME", between=(lambda z: _r(z, em))), em)
    else:
      ry = _r(cnn.conv1d_tf(r, weights[rn + "/kernel"], 1, 1, "SAME"), em)
    tot = tot + cnn.batch_norm_train(ry, weights[bn + "/gamma"], weights[bn + "/beta"], bn_eps)[0]
  tot = _r(tot, em, fwd=False, bwd=True)      # dz is stored in bf16
  out = cnn.act_fn(tot, activation)
  if keep_mask is not None:
    out = out * keep_mask.float() / keep_prob
  if out_mask is not None:
    out = out * out_mask   # idempotent w.r.t. the reference's later multiply
  return _r(out, em)                           # block outputs / their grads: bf16


def tdnn_encode(x, src_len, convnet_layers, weights, activation="relu", use_conv_mask=True,
                bn_eps=1e-3, keep_masks=None, keep_probs=None, emulate_bf16=False):
  """x [B,T,F] fp32; weights: dict name -> tensor in TF layouts
  ('convIJ/kernel' [K,Cin,Cout], 'convIJ/bn/gamma', '.../res_N/kernel', ...).
  keep_masks: optional list (per layer) of dropout keep masks [B,T',C]."""
  src_len = torch.as_tensor(src_len).clone()
  T = x.shape[1]
  mask = cnn.seq_mask(src_len, T) if use_conv_mask else None
  feats = x
  res_agg = []
  li = 0
  for ib, blk in enumerate(convnet_layers):
    K, s = blk["kernel_size"][0], blk["stride"][0]
    d = blk.get("dilation", [1])[0]
    residual, dense = blk.get("residual", False), blk.get("residual_dense", False)
    if use_conv_mask:
      feats = feats * mask
    if residual:
      layer_res = feats
      if dense:
        res_agg.append(layer_res)
        layer_res = list(res_agg)
      else:
        layer_res = [layer_res]
    for ir in range(blk["repeat"]):
      name = "conv%d%d" % (ib + 1, ir + 1)
      if blk["padding"] == "VALID":
        src_len = (src_len - K) // s + 1
        T = (T - K) // s + 1
      else:
        src_len = (src_len + s - 1) // s
        T = (T + s - 1) // s
      if ir > 0 and use_conv_mask:
        feats = feats * mask
      if use_conv_mask and (blk["padding"] == "VALID" or s > 1):
        mask = cnn.seq_mask(src_len, T)
      last = ib == len(convnet_layers) - 1 and ir == blk["repeat"] - 1
      feats = tdnn_layer(feats, layer_res if (residual and ir == blk["repeat"] - 1) else [], blk, name,
                         weights, mask if (use_conv_mask and not last) else None, activation, bn_eps,
                         keep_masks[li] if keep_masks is not None else None,
                         keep_probs[li] if keep_probs is not None else 1.0, emulate_bf16)
      li += 1
  return feats, src_len


def fc_ctc(outputs, src_len, fc_kernel, fc_bias, labels, label_len):
  """FullyConnectedTimeDecoder + CTCLoss: returns (logits [T,B,V], mean loss)."""
  logits = (outputs @ fc_kernel + fc_bias).permute(1, 0, 2)
  T, B, V = logits.shape
  lp = torch.log_softmax(logits.double(), -1)
  feas = torch.tensor([ctc._feasible([int(v) for v in labels[b][:label_len[b]]], int(src_len[b]))
                       for b in range(B)])
  loss = torch.nn.functional.ctc_loss(
      lp, torch.as_tensor(labels, dtype=torch.long), torch.as_tensor(src_len, dtype=torch.long),
      torch.as_tensor(label_len, dtype=torch.long), blank=V - 1, reduction="none",
      zero_infinity=True)
  loss = torch.where(feas, loss, torch.zeros_like(loss))
  return logits, loss.mean().float()
