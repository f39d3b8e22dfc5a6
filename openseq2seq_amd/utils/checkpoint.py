"""Checkpoint I/O under the reference's variable names and TensorFlow layouts.

The reference saves tf.train.Saver checkpoints (utils/funcs.py:117-144, utils/hooks.py:227-236)
and restores by NAME and SHAPE (utils/helpers.py:462-553), falling back from a half-precision
variable to its 'Loss_Optimization/FP32-master-copy/<name>' twin. The container is the
reference's own: a TensorFlow V2 checkpoint (<prefix>.index + <prefix>.data-00000-of-00001),
read and written by utils/tensor_bundle.py without TensorFlow; a NumPy .npz with the same keys
(format='npz', the round-1 container) is still read and written. Every array is stored under the
reference's variable name, in the reference's layout:

  conv1d kernel   device [K, Cout, Cin]        -> tf.layers.conv1d  [K, Cin, Cout]
  dense kernel    device [1, Cout, Cin]        -> tf.layers.dense   [Cin, Cout]
  fused qkv / kv  device [1, 3D|2D, D]         -> .../q/kernel, .../k/kernel, .../v/kernel  [D, D]
  shared embedding device [1, V, D]            -> embedding_and_softmax/weights [V, D]
  depthwise       device [K, C]                -> separable_conv1d depthwise_kernel [K, C, 1]
  BatchNorm       gamma / beta / moving_mean / moving_variance  (fp32, as is)
  LSTM cell       device wx_0 [1, 4H, in0], wx_1 ..., wh [1, 4H, H] (or kernel_inputs +
                  kernel_attention_state)      -> ONE lstm_cell/kernel [in0 + in1 + ... + H, 4H] per cell
                  (rows: the cell's inputs in order, then h; gate order i, j, f, o on both sides), under the
                  scope names the reference's GRAPH gives them (bidirectional_rnn/fw/..., rnn/...,
                  decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/..., AttentionMechanism/...):
                  the RNN NMT encoders, RNNDecoderWithAttention (gnmt / gnmt_v2) and Tacotron2Decoder, names
                  checked against the reference's executed graphs (tests/golden/ref_exec_{nmt,tacotron}_*.npz)
  anything else   as is (conv2d kernels are already [KT, KF, Cin, Cout]; the cuDNN layers — DeepSpeech2's
                  recurrent layers, Tacotron 2's encoder LSTM — keep the device names and gate order: they are
                  one opaque buffer in a TensorFlow checkpoint)

In mixed precision a half-precision variable is written as DT_HALF (float16) under its plain
name and as fp32 under the master-copy name — the dtypes a reference fp16 graph holds, so a
plain tf.train.Saver restore and the casting restore of helpers.py both accept the file;
import prefers the fp32 twin. The byte format has never been compared with a file written by
TensorFlow itself (none ships with the reference): "TF-V2-shaped, unverified". Optimizer slots, the
global step and the loss-scaler state are stored under 'OS2S/...' keys (own format).
"""
from __future__ import absolute_import, division, print_function

import os
import re

import numpy as np
import torch

MASTER_PREFIX = "Loss_Optimization/FP32-master-copy/"
LATEST_FILENAME = "checkpoint"


# K = 1 tf.layers.conv1d layers of the Tacotron 2 decoder that sit under no conv... scope: the location-sensitive
# attention's memory layer (parts/rnns/attention_wrapper.py: LocationSensitiveAttention, memory_layer = Conv1D) and
# the magnitude head (decoders/tacotron2_decoder.py: post_net_proj) — rank 3 in tf.trainable_variables() of the
# reference's executed graph (tests/golden/ref_exec_tacotron_full.npz)
_TACOTRON_CONV1D = re.compile(r"/tacotron_2_decoder/(AttentionMechanism/memory_layer|post_net_proj)/kernel$")


def _is_dense(name):
  """Is this [1, Cout, Cin] matrix a tf.layers.dense kernel ([Cin, Cout] in a checkpoint) rather than a K = 1
  tf.layers.conv1d kernel ([1, Cin, Cout])? Convolution variables sit under a scope component named conv...
  ('conv61/kernel', and the residual branches 'conv22/res_0/kernel', 'conv22/res/kernel' of
  parts/cnns/conv_blocks.py:78-85 — until round 5 the branches were taken for dense kernels and written / read
  with the wrong rank; found by running the reference's own TDNNEncoder, tests/test_ref_exec_tdnn_gpu.py)."""
  parts = name.split("/")
  if parts[-1] == "pointwise_kernel":
    return False
  if _TACOTRON_CONV1D.search(name):
    return False
  return not any(re.match(r"conv(_|\d|$)", c) for c in parts[:-1])


def export_param(name, shape, kind, arr, logical_out=None):
  """device array -> [(tf_name, tf_array)]. Output layers padded to an MFMA-friendly width
  (FullyConnectedCTCDecoder V=29 -> 32 rows, RNN decoders V -> multiple of 8) are written with
  the reference's LOGICAL shape: only the first `logical_out` output units."""
  if logical_out is not None:
    if kind == "conv" and len(shape) == 3 and shape[0] == 1:
      arr = arr[:, :logical_out, :]
      shape = (1, logical_out, shape[2])
    elif len(shape) == 1:
      arr = arr[:logical_out]
      shape = (logical_out,)
  if name.endswith("/qkv/kernel") or name.endswith("/kv/kernel"):
    base = name[:name.rindex("/", 0, len(name) - len("/kernel"))]
    parts = ("q", "k", "v") if name.endswith("/qkv/kernel") else ("k", "v")
    D = shape[2]
    return [("%s/%s/kernel" % (base, t), arr[0, i * D:(i + 1) * D, :].T.copy()) for i, t in enumerate(parts)]
  if name.endswith("embedding_and_softmax/weights"):
    return [(name, arr[0].copy())]
  if kind == "conv":
    if shape[0] == 1 and _is_dense(name):
      return [(name, arr[0].T.copy())]
    return [(name, np.transpose(arr, (0, 2, 1)).copy())]
  if name.endswith("/depthwise_kernel"):
    return [(name, arr[:, :, None].copy())]
  if name.endswith("/location_conv/kernel") and arr.ndim == 2:       # [K, F] -> conv1d over the one alignment channel
    return [(name, arr[:, None, :].copy())]
  if name.endswith("/location_dense/kernel") and arr.ndim == 2:      # [F, U] -> the K = 1 conv1d the reference uses
    return [(name, arr[None].copy())]
  return [(name, arr.copy())]


def import_param(name, shape, kind, tf_arrays, logical_out=None):
  """inverse of export_param; returns the device-layout array or None if names are missing.
  A layer stored with its logical width is zero-padded back to the device width."""
  if logical_out is not None:
    if kind == "conv" and len(shape) == 3 and shape[0] == 1:
      a = import_param(name, (1, logical_out, shape[2]), kind, tf_arrays)
      if a is None or a.shape[1] not in (logical_out, shape[1]):
        return a
      out = np.zeros(shape, np.float32)
      out[:, :a.shape[1], :] = a
      return out
    if len(shape) == 1:
      a = import_param(name, (logical_out,), kind, tf_arrays)
      if a is None or a.shape[0] not in (logical_out, shape[0]):
        return a
      out = np.zeros(shape, np.float32)
      out[:a.shape[0]] = a
      return out

  def get(n):
    # the fp32 master twin first (exact), the (possibly half-precision) plain name otherwise
    if MASTER_PREFIX + n in tf_arrays:
      return np.asarray(tf_arrays[MASTER_PREFIX + n], np.float32)
    if n in tf_arrays:
      return np.asarray(tf_arrays[n], np.float32)
    return None
  if name.endswith("/qkv/kernel") or name.endswith("/kv/kernel"):
    base = name[:name.rindex("/", 0, len(name) - len("/kernel"))]
    parts = ("q", "k", "v") if name.endswith("/qkv/kernel") else ("k", "v")
    mats = [get("%s/%s/kernel" % (base, t)) for t in parts]
    if any(m is None for m in mats):
      return None
    return np.concatenate([m.T for m in mats], axis=0)[None]
  a = get(name)
  if a is None and name.endswith("/embedding_shared_weights/embedding_and_softmax/weights"):
    a = get("ForwardPass/embedding_and_softmax/weights")      # checkpoints this repository wrote before round 5
  if a is None:
    return None
  if name.endswith("embedding_and_softmax/weights"):
    return a[None]
  if kind == "conv":
    if a.ndim == 2:                       # tf.layers.dense [Cin, Cout]
      return a.T[None]
    return np.transpose(a, (0, 2, 1))     # tf.layers.conv1d [K, Cin, Cout] (the array's own rank decides)
  if name.endswith("/depthwise_kernel"):
    return a[:, :, 0]
  if name.endswith(("/location_conv/kernel", "/location_dense/kernel")) and a.ndim == 3 and len(shape) == 2:
    return a.reshape(shape) if a.size == int(np.prod(shape)) else a
  return a


# ---------------------------------------------------------------------------------------------------------
# RNN NMT models: device scope names -> the names of the reference's graph. The reference BUILDS its cells under
# 'FW' / 'BW' / 'Level1FW' / 'UniDirLevel' (encoders/rnn_encoders.py:276-282, 392-417) but the variables are created
# when bidirectional_dynamic_rnn / dynamic_rnn / dynamic_decode first call the cells, under THEIR scopes.
# ---------------------------------------------------------------------------------------------------------
_RNN_SCOPES = [
    (r"/bidir_rnn_encoder_with_emb/FW/", "/bidir_rnn_encoder_with_emb/bidirectional_rnn/fw/"),
    (r"/bidir_rnn_encoder_with_emb/BW/", "/bidir_rnn_encoder_with_emb/bidirectional_rnn/bw/"),
    (r"/gnmt_encoder_with_emb/Level1FW/", "/gnmt_encoder_with_emb/bidirectional_rnn/fw/"),
    (r"/gnmt_encoder_with_emb/Level1BW/", "/gnmt_encoder_with_emb/bidirectional_rnn/bw/"),
    (r"/gnmt_encoder_with_emb/UniDirLevel/", "/gnmt_encoder_with_emb/rnn/"),
    (r"/rnn_decoder_with_attention/attention_cell/cell_0/",
     "/rnn_decoder_with_attention/decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/lstm_cell/"),
    (r"/rnn_decoder_with_attention/attention_cell/attention/memory_layer/",
     "/rnn_decoder_with_attention/AttentionMechanism/memory_layer/"),
    (r"/rnn_decoder_with_attention/attention_cell/attention/",
     "/rnn_decoder_with_attention/decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/bahdanau_attention/"),
    (r"/rnn_decoder_with_attention/multi_rnn_cell/", "/rnn_decoder_with_attention/decoder/multi_rnn_cell/"),
    (r"/rnn_decoder_with_attention/dense/", "/rnn_decoder_with_attention/decoder/dense/"),
    # Tacotron2Decoder (decoders/tacotron2_decoder.py: everything the decoder step touches lives under dynamic_decode's
    # 'decoder' scope; the memory layer is created with the attention mechanism, outside it)
    (r"/tacotron_2_decoder/attention_wrapper/attention/memory_layer/", "/tacotron_2_decoder/AttentionMechanism/memory_layer/"),
    (r"/tacotron_2_decoder/attention_wrapper/attention/", "/tacotron_2_decoder/decoder/attention_wrapper/location_attention/"),
    (r"/tacotron_2_decoder/attention_wrapper/cell_0/",
     "/tacotron_2_decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/lstm_cell/"),
    (r"/tacotron_2_decoder/attention_wrapper/cell_", "/tacotron_2_decoder/decoder/attention_wrapper/multi_rnn_cell/cell_"),
    (r"/tacotron_2_decoder/prenet_", "/tacotron_2_decoder/decoder/prenet_"),
    (r"/tacotron_2_decoder/output_proj/", "/tacotron_2_decoder/decoder/output_proj/"),
    (r"/tacotron_2_decoder/stop_token_proj/", "/tacotron_2_decoder/decoder/stop_token_proj/"),
]
_UPPER_CELL = re.compile(r"^(.*/tacotron_2_decoder/decoder/attention_wrapper/multi_rnn_cell/cell_[1-9]\d*)/(kernel|bias)$")
_LSTM_PART = re.compile(r"^(.*)/(wx_(\d+)|wh|kernel_inputs|kernel_attention_state)$")


def reference_name(name):
  """device parameter name -> (reference variable name, rank of this parameter among the row blocks of that
  variable or None when the parameter IS the variable)."""
  for pat, rep in _RNN_SCOPES:
    if pat in name:
      name = name.replace(pat, rep, 1)
      break
  else:
    return name, None
  m = _UPPER_CELL.match(name)
  if m:                                   # the upper decoder cells are one [1, 4H, 2H] matrix here: a rename
    return "%s/lstm_cell/%s" % (m.group(1), m.group(2)), None
  m = _LSTM_PART.match(name)
  if not m:
    return name, None
  part = m.group(2)
  order = int(m.group(3)) if m.group(3) is not None else {"kernel_inputs": 0, "kernel_attention_state": 1,
                                                           "wh": 1 << 20}[part]
  return m.group(1) + "/kernel", order


def lstm_kernel_groups(params):
  """{reference kernel name: [device parameters, in row order]} for the LSTM cells stored as one kernel by the
  reference and as several matrices here."""
  groups = {}
  for p in params:
    ref, order = reference_name(p.name)
    if order is not None:
      groups.setdefault(ref, []).append((order, p))
  return {k: [p for _, p in sorted(v, key=lambda t: t[0])] for k, v in groups.items()}


# ---------------------------------------------------------------------------------------------------------
# cuDNN layers (tf.contrib.cudnn_rnn.CudnnGRU / CudnnLSTM: DeepSpeech2's GRU stack, encoders/ds2_encoder.py:294-328;
# Tacotron 2's encoder LSTM, encoders/tacotron2_encoder.py:254-263). The graph holds ONE opaque buffer per layer
# stack; its Saveable (tensorflow/contrib/cudnn_rnn/python/ops/cudnn_rnn_ops.py, CudnnOpaqueParamsSaveable) writes
# the checkpoint in the "canonical" form a CudnnCompatible{GRU,LSTM}Cell restores from:
#   <layer scope>/stack_bidirectional_rnn/cell_<l>/bidirectional_rnn/{fw,bw}/<cell>/...        (bidirectional)
#   <layer scope>/rnn/multi_rnn_cell/cell_<l>/<cell>/...                                        (unidirectional)
#   GRU  (<cell> = cudnn_compatible_gru_cell):  gates/kernel [in + H, 2H] (columns: reset | update; rows: input,
#        then state), gates/bias [2H] = b_W + b_R of the two gates, candidate/input_projection/{kernel [in, H],
#        bias}, candidate/hidden_projection/{kernel [H, H], bias} (cuDNN applies the reset gate to R_h h + b_Rh,
#        so the two candidate biases stay apart)
#   LSTM (<cell> = cudnn_compatible_lstm_cell): kernel [in + H, 4H] in LSTMBlockCell's gate order i, c, f, o
#        (cuDNN's is i, f, c, o), bias [4H] = b_W + b_R.
# The device keeps cuDNN's own form per direction — wx_0 [1, G H, in], wh [1, G H, H], bias (b_W), bias_h (b_R),
# gate rows r, z, n / i, f, g, o (csrc/rnn.hip) — so the exchange is a stack / split / transpose; on the way IN
# the summed gate biases are halved over b_W and b_R, as the Saveable does. Canonical tensors carry the dtype of the
# opaque variable (half in a mixed-precision graph, see model_variables). TensorFlow is not in /root/reference: the layout
# above is restated from the TF 1.x source; tests/test_checkpoint_shapes.py holds the exported tensors to the
# CudnnCompatible cells' equations against the CPU restatement of the cells (tests/).
# ---------------------------------------------------------------------------------------------------------
_CUDNN_PART = re.compile(r"^(.*/(?:cudnn_gru|cudnn_lstm|cudnn_rnn))/layer_(\d+)/(fw|bw)/(wx_0|wh|bias|bias_h)$")
_CUDNN_TF_ORDER = {3: (0, 1, 2), 4: (0, 2, 1, 3)}       # device gate index of TF's k-th gate block


def cudnn_groups(params):
  """{(layer scope, layer, 'fw' | 'bw'): {'wx_0': p, 'wh': p, 'bias': p, 'bias_h': p}} for the complete cuDNN-form
  directions among `params` (objects with .name and .shape)."""
  groups = {}
  for p in params:
    m = _CUDNN_PART.match(p.name)
    if m:
      groups.setdefault((m.group(1), int(m.group(2)), m.group(3)), {})[m.group(4)] = p
  return {k: v for k, v in groups.items() if len(v) == 4}


def cudnn_canonical_prefix(scope, layer, tag, bidirectional, gates):
  cell = "cudnn_compatible_gru_cell" if gates == 3 else "cudnn_compatible_lstm_cell"
  if bidirectional:
    return "%s/stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/%s" % (scope, layer, tag, cell)
  return "%s/rnn/multi_rnn_cell/cell_%d/%s" % (scope, layer, cell)


def cudnn_to_canonical(wx, wh, bx, bh):
  """Device arrays of one direction (wx [1, G H, in], wh [1, G H, H], bx / bh [G H]) -> {suffix: TF tensor}."""
  H = wh.shape[2]
  G = wh.shape[1] // H
  wx, wh = wx[0], wh[0]
  blk = lambda a, g: a[g * H:(g + 1) * H]
  if G == 3:
    rz = lambda a: np.concatenate([blk(a, 0), blk(a, 1)], axis=0)
    return {"gates/kernel": np.concatenate([rz(wx).T, rz(wh).T], axis=0),
            "gates/bias": rz(bx) + rz(bh),
            "candidate/input_projection/kernel": blk(wx, 2).T.copy(),
            "candidate/input_projection/bias": blk(bx, 2).copy(),
            "candidate/hidden_projection/kernel": blk(wh, 2).T.copy(),
            "candidate/hidden_projection/bias": blk(bh, 2).copy()}
  order = _CUDNN_TF_ORDER[4]
  st = lambda a: np.concatenate([blk(a, g) for g in order], axis=0)
  return {"kernel": np.concatenate([st(wx).T, st(wh).T], axis=0), "bias": st(bx) + st(bh)}


def canonical_to_cudnn(get, gates, n_in, H):
  """Inverse: get(suffix) -> TF tensor or None. Returns (wx, wh, bx, bh) in device layout, or None when a tensor is
  missing or mis-shaped."""
  def stack(blocks):
    return np.concatenate(blocks, axis=0)
  if gates == 3:
    gk, gb = get("gates/kernel"), get("gates/bias")
    ik, ib = get("candidate/input_projection/kernel"), get("candidate/input_projection/bias")
    hk, hb = get("candidate/hidden_projection/kernel"), get("candidate/hidden_projection/bias")
    if any(a is None for a in (gk, gb, ik, ib, hk, hb)):
      return None
    if gk.shape != (n_in + H, 2 * H) or ik.shape != (n_in, H) or hk.shape != (H, H) or gb.shape != (2 * H,):
      return None
    wx = stack([gk[:n_in].T, ik.T])[None]
    wh = stack([gk[n_in:].T, hk.T])[None]
    return wx, wh, stack([0.5 * gb, ib]), stack([0.5 * gb, hb])
  k, b = get("kernel"), get("bias")
  if k is None or b is None or k.shape != (n_in + H, 4 * H) or b.shape != (4 * H,):
    return None
  order = _CUDNN_TF_ORDER[4]                       # TF block j holds device gate order[j]
  dev_of = [order.index(g) for g in range(4)]       # device gate g sits in TF block dev_of[g]
  kt = k.T
  pick = lambda a: stack([a[j * H:(j + 1) * H] for j in dev_of])
  return pick(kt[:, :n_in])[None], pick(kt[:, n_in:])[None], pick(0.5 * b), pick(0.5 * b)


def _half_in_reference(p):
  """Is this variable DT_HALF with an fp32 master twin in a mixed-precision graph of the reference? Everything
  the mixed-precision wrapper sees (optimizers/mp_wrapper.py:55-82) — kernels AND the biases of dense / recurrent
  layers — except the variables the reference creates as fp32 explicitly: BatchNorm gamma / beta, the row
  convolution (encoders/ds2_encoder.py:54-57) and the LayerNorm scale / bias of the Transformer
  (parts/transformer/common.py:48-53). The reference's own count for its toy models: 7 of 14 (DeepSpeech2),
  6 of 14 (Wave2Letter) — models/speech2text_{ds2,w2l}_test.py."""
  if p.kind != "vector":
    return True
  n = p.name
  # tf.contrib.layers.layer_norm / instance_norm create gamma / beta in the dtype of their INPUT (conv_ln_actv /
  # conv_in_actv, parts/cnns/conv_blocks.py:234-309): half in a mixed-precision graph, with fp32 master twins
  if re.search(r"/(LayerNorm|InstanceNorm)(_\d+)?/(gamma|beta)$", n):
    return True
  return (n.endswith("/bias") or n.endswith("/bias_h")) and "/bn/" not in n and "/row_conv/" not in n


def model_variables(model):
  """{reference name: array in TF layout} for every variable of the model; dtypes as a
  reference checkpoint of the same precision mode holds them."""
  store = model.store
  out = {}
  mixed = model.params.get("dtype", "mixed") == "mixed"
  groups = lstm_kernel_groups(store.params)
  grouped = {p.name for ps in groups.values() for p in ps}
  cgroups = cudnn_groups(store.params)
  bidir = {scope for scope, _, tag in cgroups if tag == "bw"}
  for (scope, layer, tag), parts in cgroups.items():
    dev = {k: parts[k].master.detach().cpu().numpy() for k in parts}
    gates = parts["wh"].shape[1] // parts["wh"].shape[2]
    prefix = cudnn_canonical_prefix(scope, layer, tag, scope in bidir, gates)
    for suffix, arr in cudnn_to_canonical(dev["wx_0"], dev["wh"], dev["bias"], dev["bias_h"]).items():
      arr = np.ascontiguousarray(arr, dtype=np.float32)
      if mixed:
        # the opaque buffer of a mixed-precision graph is DT_HALF (the reference counts it among its FP32 master
        # copies: speech2text_test.py mp_collection_test, 7 for DeepSpeech2), so its Saveable writes half-precision
        # canonical tensors; the fp32 twin the reference keeps is the OPAQUE master buffer, which has no portable
        # layout — this repository stores the canonical tensors again in fp32 under the master-copy prefix (exact
        # round trips of its own files; a TensorFlow-written file simply has no such entries)
        out["%s/%s" % (prefix, suffix)] = arr.astype(np.float16)
        out["%s%s/%s" % (MASTER_PREFIX, prefix, suffix)] = arr
      else:
        out["%s/%s" % (prefix, suffix)] = arr
    grouped |= {p.name for p in parts.values()}
  for ref, ps in groups.items():
    # [1, 4H, in_k] blocks -> one [sum in_k, 4H] kernel, the cell's inputs first, h last
    k = np.concatenate([p.master.detach().cpu().numpy()[0].T for p in ps], axis=0)
    if mixed:
      out[ref] = k.astype(np.float16)
      out[MASTER_PREFIX + ref] = k
    else:
      out[ref] = k
  for p in store.params:
    if p.name in grouped:
      continue
    arr = p.master.detach().cpu().numpy()
    for tf_name, tf_arr in export_param(reference_name(p.name)[0], p.shape, p.kind, arr,
                                        getattr(p, "logical_out", None)):
      if mixed and _half_in_reference(p):
        # a mixed-precision graph of the reference holds this variable as DT_HALF and its fp32
        # twin under the master-copy name (optimizers/mp_wrapper.py:55-82): a plain
        # tf.train.Saver restore checks the dtype, so the plain name carries float16
        out[tf_name] = tf_arr.astype(np.float16)
        out[MASTER_PREFIX + tf_name] = tf_arr
      else:
        out[tf_name] = tf_arr
  for name, t in store.state.items():
    out[name] = t.detach().cpu().numpy().copy()
  return out


def open_checkpoint(prefix):
  """name -> array mapping of a checkpoint prefix: a TensorBundle (tf.train.Saver V2 files) when
  <prefix>.index exists, else <prefix>.npz."""
  from . import tensor_bundle
  if prefix.endswith(".npz"):
    return np.load(prefix)
  if tensor_bundle.is_bundle(prefix):
    return tensor_bundle.BundleReader(prefix)
  if os.path.exists(prefix + ".npz"):
    return np.load(prefix + ".npz")
  return tensor_bundle.BundleReader(prefix)       # raises the reference-style "not found" error


def save(model, logdir, step=None, format="tf"):
  """Writes <logdir>/model.ckpt-<step> (format 'tf': .index + .data-00000-of-00001; 'npz': .npz)
  and the 'checkpoint' state file tf.train.latest_checkpoint reads."""
  os.makedirs(logdir, exist_ok=True)
  arrays = model_variables(model)
  store = model.store
  train_op = getattr(model, "_train_op", None)
  if step is None:
    step = model.global_step() if train_op is not None else 0
  arrays["global_step"] = np.asarray(step, np.int64)
  if train_op is not None:
    arrays["OS2S/opt/m1"] = store.m1.detach().cpu().numpy()
    if store.m2 is not None:
      arrays["OS2S/opt/m2"] = store.m2.detach().cpu().numpy()
    arrays["OS2S/opt/state"] = train_op.state.detach().cpu().numpy()
    arrays["OS2S/opt/t_v"] = store.t_v.detach().cpu().numpy()
    arrays["OS2S/opt/layout"] = slot_layout(store)
  prefix = "model.ckpt-%d" % int(step)
  if format == "npz":
    np.savez(os.path.join(logdir, prefix + ".npz"), **arrays)
  elif format == "tf":
    from . import tensor_bundle
    tensor_bundle.write_bundle(os.path.join(logdir, prefix), arrays)
  else:
    raise ValueError("checkpoint format must be 'tf' or 'npz'")
  with open(os.path.join(logdir, LATEST_FILENAME), "w") as f:
    f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (prefix, prefix))
  return os.path.join(logdir, prefix)


def latest_checkpoint(logdir):
  """tf.train.latest_checkpoint: the prefix named by <logdir>/checkpoint, or None."""
  path = os.path.join(logdir, LATEST_FILENAME)
  if not os.path.exists(path):
    return None
  m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(path).read())
  if not m:
    return None
  prefix = m.group(1)
  return prefix if os.path.isabs(prefix) else os.path.join(logdir, prefix)


def read_step(prefix):
  """The training step stored with the checkpoint (global_step)."""
  return int(open_checkpoint(prefix)["global_step"])


def load(model, prefix, restore_optimizer=True, strict=True):
  """Restores by name and shape (helpers.py:462-553). Returns the list of variables that were
  not found (empty with strict=True, which raises instead)."""
  path = prefix
  data = open_checkpoint(prefix)
  store = model.store
  missing = []
  split = {}
  for ref, ps in lstm_kernel_groups(store.params).items():
    k = None
    for n in (MASTER_PREFIX + ref, ref):
      if n in data:
        k = np.asarray(data[n], np.float32)
        break
    if k is None or k.ndim != 2 or k.shape != (sum(p.shape[2] for p in ps), ps[0].shape[1]):
      continue                      # falls through to the per-parameter lookup below (and its diagnostics)
    row = 0
    for p in ps:
      split[p.name] = k[row:row + p.shape[2]].T[None]
      row += p.shape[2]
  cgroups = cudnn_groups(store.params)
  bidir = {scope for scope, _, tag in cgroups if tag == "bw"}
  for (scope, layer, tag), parts in cgroups.items():
    H = parts["wh"].shape[2]
    gates, n_in = parts["wh"].shape[1] // H, parts["wx_0"].shape[2]
    prefix = cudnn_canonical_prefix(scope, layer, tag, scope in bidir, gates)

    def get(suffix, prefix=prefix):
      for n in (MASTER_PREFIX + prefix + "/" + suffix, prefix + "/" + suffix):
        if n in data:
          return np.asarray(data[n], np.float32)
      return None
    got = canonical_to_cudnn(get, gates, n_in, H)
    if got is not None:               # else: the per-parameter names of this repository's older files, below
      for key, a in zip(("wx_0", "wh", "bias", "bias_h"), got):
        split[parts[key].name] = a
  for p in store.params:
    a = split.get(p.name)
    if a is None:
      a = import_param(reference_name(p.name)[0], p.shape, p.kind, data, getattr(p, "logical_out", None))
    if a is None and reference_name(p.name)[0] != p.name:     # files this repository wrote before the translation
      a = import_param(p.name, p.shape, p.kind, data, getattr(p, "logical_out", None))
    if a is None or tuple(a.shape) != tuple(p.shape):
      missing.append(p.name)
      continue
    p.master.copy_(torch.from_numpy(np.ascontiguousarray(a)))
  for name, t in store.state.items():
    if name in data and tuple(data[name].shape) == tuple(t.shape):
      t.copy_(torch.from_numpy(np.asarray(data[name], np.float32)))
    else:
      missing.append(name)
  if missing and strict:
    raise ValueError("checkpoint %s lacks (or mis-shapes) variables: %s" % (path, ", ".join(missing[:8])))
  store.refresh_compute_copies()
  train_op = getattr(model, "_train_op", None)
  if restore_optimizer and train_op is not None and "OS2S/opt/state" in data:
    restore_slots(store, train_op, data, path)
  return missing


def slot_layout(store):
  """[n, 3] int64: (crc32 of the variable name, offset, element count) of every trainable variable in the flat
  buffers, in creation order — what the optimizer slots (OS2S/opt/m1, m2: flat fp32 buffers; t_v: one value per
  variable) are laid out by. The total of the chunk-padded sizes does not depend on the creation order, so the
  shapes of the flat buffers alone cannot tell a file written under a different order (round 5 moved the
  Transformer decoder's encoder-decoder K / V kernels in front of the per-layer variables)."""
  import zlib
  return np.asarray([[zlib.crc32(p.name.encode()) & 0xffffffff, p.offset, p.numel] for p in store.params], np.int64)


def restore_slots(store, train_op, data, path=""):
  """Optimizer moments of a checkpoint into the store: flat copies when the file's layout IS the store's, variable
  by variable (matched by name hash and size) when the creation order differs, not at all — with a warning; the
  moments then restart from zero — when the file carries no layout (written before round 6: nothing says which
  variable a slot element belongs to). The scalar optimizer state (step count, loss scale) is restored in all
  three cases."""
  import warnings
  train_op.state.copy_(torch.from_numpy(data["OS2S/opt/state"]))
  if "OS2S/opt/m1" not in data:
    return "none"
  cur = slot_layout(store)
  if "OS2S/opt/layout" not in data:
    warnings.warn("checkpoint %s holds optimizer moments without a layout record (written before round 6): they "
                  "are NOT restored (moments restart from zero; weights, step count and loss scale are restored)"
                  % path)
    return "skipped"
  old = np.asarray(data["OS2S/opt/layout"], np.int64).reshape(-1, 3)
  m1 = torch.from_numpy(np.asarray(data["OS2S/opt/m1"], np.float32))
  m2 = torch.from_numpy(np.asarray(data["OS2S/opt/m2"], np.float32)) if (store.m2 is not None and
                                                                          "OS2S/opt/m2" in data) else None
  t_v = torch.from_numpy(np.asarray(data["OS2S/opt/t_v"], np.float32)) if "OS2S/opt/t_v" in data else None
  if old.shape == cur.shape and np.array_equal(old, cur) and tuple(m1.shape) == tuple(store.m1.shape):
    store.m1.copy_(m1)
    if m2 is not None:
      store.m2.copy_(m2)
    if t_v is not None:
      store.t_v.copy_(t_v)
    return "flat"
  # a different creation order (or a different set of variables): slot by slot
  by_key = {(int(h), int(n)): (i, int(off)) for i, (h, off, n) in enumerate(old)}
  lost = []
  for i, (h, off, n) in enumerate(cur):
    hit = by_key.get((int(h), int(n)))
    if hit is None or hit[1] + int(n) > m1.numel():
      lost.append(store.params[i].name)
      continue
    j, ooff = hit
    store.m1[int(off):int(off) + int(n)].copy_(m1[ooff:ooff + int(n)])
    if m2 is not None:
      store.m2[int(off):int(off) + int(n)].copy_(m2[ooff:ooff + int(n)])
    if t_v is not None and j < t_v.numel():
      store.t_v[i] = t_v[j]
  if lost:
    warnings.warn("checkpoint %s: no optimizer moments for %d variable(s) (e.g. %s): they restart from zero"
                  % (path, len(lost), ", ".join(lost[:4])))
  return "by_name"
