"""Checkpoint exchange with the reference for layers whose device copy is padded: the reference's
FullyConnectedCTCDecoder stores dense kernel [H, V] and bias [V] with V = 29
(decoders/fc_decoders.py:135-148), the device keeps [1, 32, H] / [32]. export_param must write the
logical shape, import_param must accept it (and the padded one) — utils/helpers.py:462-553
restores by NAME and SHAPE."""
import numpy as np

from openseq2seq_amd.utils import checkpoint as ck


def test_padded_output_layer_round_trips_with_reference_shapes():
  H, V, Vpad = 48, 29, 32
  rng = np.random.RandomState(0)
  w_dev = np.zeros((1, Vpad, H), np.float32)
  w_dev[0, :V] = rng.randn(V, H)
  b_dev = np.zeros((Vpad,), np.float32)
  b_dev[:V] = rng.randn(V)
  name = "ForwardPass/fully_connected_ctc_decoder/fully_connected/kernel"
  (n1, a1), = ck.export_param(name, w_dev.shape, "conv", w_dev, logical_out=V)
  (n2, a2), = ck.export_param(name[:-6] + "bias", b_dev.shape, "vector", b_dev, logical_out=V)
  assert n1 == name and a1.shape == (H, V) and a2.shape == (V,)          # TF layouts / shapes
  np.testing.assert_array_equal(a1, w_dev[0, :V].T)
  # a checkpoint written by the reference (logical shapes) loads into the padded device layout
  back_w = ck.import_param(name, w_dev.shape, "conv", {n1: a1}, logical_out=V)
  back_b = ck.import_param(n2, b_dev.shape, "vector", {n2: a2}, logical_out=V)
  np.testing.assert_array_equal(back_w, w_dev)
  np.testing.assert_array_equal(back_b, b_dev)
  # an older .npz of this repository (padded shapes) still loads
  old = ck.import_param(name, w_dev.shape, "conv", {n1: w_dev[0].T.copy()}, logical_out=V)
  np.testing.assert_array_equal(old, w_dev)
  # FP32 master-copy twin (mixed precision checkpoints of the reference)
  tw = ck.import_param(name, w_dev.shape, "conv", {ck.MASTER_PREFIX + n1: a1}, logical_out=V)
  np.testing.assert_array_equal(tw, w_dev)


def test_unpadded_layers_are_untouched():
  w = np.random.RandomState(1).randn(11, 40, 24).astype(np.float32)
  (n, a), = ck.export_param("ForwardPass/w2l_encoder/conv11/kernel", w.shape, "conv", w)
  assert a.shape == (11, 24, 40)
  np.testing.assert_array_equal(ck.import_param(n, w.shape, "conv", {n: a}), w)


def test_residual_branch_kernels_keep_the_conv1d_rank():
  """The 1x1 residual branches of conv_bn_res_bn_actv are tf.layers.conv1d(res, filters, 1, name='<layer>/res_<i>')
  (parts/cnns/conv_blocks.py:78-85): their checkpoint variable is [1, Cin, Cout], like any conv1d kernel, not the
  [Cin, Cout] of a tf.layers.dense. The variable list of the reference's own TDNNEncoder
  (tests/golden/ref_exec_tdnn.npz, produced by executing it) is the witness: every '.../kernel' under the encoder has
  rank 3, the decoder's fully_connected/kernel rank 2 — and export_param / import_param must write and read exactly
  those shapes."""
  import os
  d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_tdnn.npz"))
  seen_res = 0
  for n in [str(v) for v in d["var_names"]]:
    if not n.endswith("/kernel"):
      continue
    tf_arr = d["var/" + n]
    if "/w2l_encoder/" in n:
      assert tf_arr.ndim == 3, n
      K, cin, cout = tf_arr.shape
      dev = ck.import_param(n, (K, cout, cin), "conv", {n: tf_arr})
      assert dev.shape == (K, cout, cin), (n, dev.shape)
      np.testing.assert_array_equal(dev, np.transpose(tf_arr, (0, 2, 1)))
      (n2, back), = ck.export_param(n, dev.shape, "conv", dev)
      assert n2 == n and back.shape == tf_arr.shape, (n, back.shape, tf_arr.shape)
      np.testing.assert_array_equal(back, tf_arr)
      seen_res += "/res_" in n
    else:
      assert tf_arr.ndim == 2, n
      cin, cout = tf_arr.shape
      dev = ck.import_param(n, (1, cout, cin), "conv", {n: tf_arr})
      np.testing.assert_array_equal(dev, tf_arr.T[None])
      (n2, back), = ck.export_param(n, dev.shape, "conv", dev)
      np.testing.assert_array_equal(back, tf_arr)
  assert seen_res >= 6


def test_mixed_precision_dtypes_match_a_reference_checkpoint():
  """A reference mixed-precision graph holds weight matrices as DT_HALF and their fp32 twins under
  Loss_Optimization/FP32-master-copy/ (mp_wrapper.py:55-82); BatchNorm vectors stay fp32 with no
  twin. model_variables writes exactly those dtypes; import takes the exact fp32 twin."""
  import torch
  from openseq2seq_amd.utils import checkpoint as ck

  class P(object):
    def __init__(self, name, kind, arr):
      self.name, self.kind, self.shape, self.master = name, kind, arr.shape, torch.from_numpy(arr)

  rng = np.random.RandomState(0)
  w = (rng.randn(3, 8, 4) * 0.1).astype(np.float32)
  g = rng.randn(8).astype(np.float32)

  class Store(object):
    params = [P("ForwardPass/enc/conv11/kernel", "conv", w), P("ForwardPass/enc/conv11/bn/gamma", "vector", g)]
    state = {"ForwardPass/enc/conv11/bn/moving_mean": torch.zeros(8)}

  class M(object):
    store = Store()
    params = {"dtype": "mixed"}

  out = ck.model_variables(M())
  k = "ForwardPass/enc/conv11/kernel"
  assert out[k].dtype == np.float16 and out[k].shape == (3, 4, 8)
  assert out[ck.MASTER_PREFIX + k].dtype == np.float32
  assert out["ForwardPass/enc/conv11/bn/gamma"].dtype == np.float32
  assert ck.MASTER_PREFIX + "ForwardPass/enc/conv11/bn/gamma" not in out
  back = ck.import_param(k, w.shape, "conv", out)
  np.testing.assert_array_equal(back, w)                    # the fp32 twin, not the rounded half
  M.params = {"dtype": "float32"}
  out32 = ck.model_variables(M())
  assert out32[k].dtype == np.float32 and ck.MASTER_PREFIX + k not in out32


def test_shared_embedding_is_stored_under_the_reference_name():
  """'ForwardPass/transformer_encoder/embedding_shared_weights/embedding_and_softmax/weights' — the name the
  reference's code gives the variable when executed (first variable of tests/golden/ref_exec_transformer.npz);
  checkpoints this repository wrote earlier ('ForwardPass/embedding_and_softmax/weights') still load."""
  import os
  d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_transformer.npz"))
  name = str(d["var_names"][0])
  assert name == "ForwardPass/transformer_encoder/embedding_shared_weights/embedding_and_softmax/weights"
  tab = d["var/" + name]
  dev = ck.import_param(name, (1,) + tab.shape, "conv", {name: tab})
  np.testing.assert_array_equal(dev[0], tab)
  old = ck.import_param(name, (1,) + tab.shape, "conv", {"ForwardPass/embedding_and_softmax/weights": tab})
  np.testing.assert_array_equal(old[0], tab)
  (n2, back), = ck.export_param(name, dev.shape, "conv", dev)
  assert n2 == name and back.shape == tab.shape
