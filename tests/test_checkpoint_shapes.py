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


def _cpu_store(monkeypatch):
  """A FlatParams whose finalize() only gives every parameter a seeded CPU master (no device buffers)."""
  import torch
  from openseq2seq_amd.optimizers import flat_params

  def finalize(self, need_m2=False):
    g = torch.Generator().manual_seed(5)
    for p in self.params:
      p.master = torch.randn(tuple(p.shape), generator=g)
    self.finalized = True
  monkeypatch.setattr(flat_params.FlatParams, "finalize", finalize)
  monkeypatch.setattr(flat_params.FlatParams, "refresh_compute_copies", lambda self: None)
  return flat_params.FlatParams(torch.device("cpu"))


def _nmt_device_model(monkeypatch, encoder, V, E, H, U, enc_layers, dec_layers, att, dtype="float32", M=None):
  from openseq2seq_amd import encoders, decoders
  store = _cpu_store(monkeypatch)
  cellp = {"num_units": H, "forget_bias": 1.0}
  if encoder is not None:
    getattr(encoders, encoder)(
        {"src_vocab_size": V, "src_emb_size": E, "encoder_layers": enc_layers, "encoder_use_skip_connections": False,
         "core_cell": "LSTMCell", "core_cell_params": cellp, "encoder_dp_input_keep_prob": 1.0, "dtype": "mixed"},
        None, mode="train").build(store)
  if dec_layers:
    dec = decoders.RNNDecoderWithAttention(
        {"GO_SYMBOL": 2, "END_SYMBOL": 1, "tgt_vocab_size": V, "tgt_emb_size": E, "attention_layer_size": U,
         "attention_type": att, "core_cell": "LSTMCell", "core_cell_params": cellp, "decoder_layers": dec_layers,
         "decoder_use_skip_connections": False, "decoder_dp_input_keep_prob": 1.0, "batch_size": 4, "dtype": "mixed"},
        None, mode="train")
    dec.build(store, memory_dim=M if M is not None else 2 * H)
  store.finalize()

  class M(object):
    params = {"dtype": dtype}
  M.store = store
  return M()


def test_rnn_nmt_checkpoints_carry_the_reference_graphs_variables(monkeypatch):
  """The variable lists of the reference's EXECUTED graphs (tests/golden/ref_exec_nmt_*.npz: names and shapes of
  tf.trainable_variables() after open_seq2seq's own encoders / RNNDecoderWithAttention ran) against what
  utils/checkpoint.py writes for the device models of the same configuration: the same names, the same shapes —
  an LSTM cell is ONE kernel [inputs + H, 4H] there and two or three matrices here — and reading the reference's
  arrays back puts every row block where the GPU test (tests/test_ref_exec_nmt_gpu.py) puts it by hand."""
  import os
  here = os.path.dirname(os.path.abspath(__file__))
  # bidirectional encoder + gnmt_v2 decoder at the dims of ref_exec_nmt_full
  d = np.load(os.path.join(here, "golden", "ref_exec_nmt_full.npz"))
  want = {str(n): tuple(int(v) for v in d["shape/" + str(n)]) for n in d["var_names"]}
  m = _nmt_device_model(monkeypatch, "BidirectionalRNNEncoderWithEmbedding", 30, 64, 64, 128, 2, 2, "gnmt_v2")
  got = {k: v.shape for k, v in ck.model_variables(m).items()}
  assert got == want, sorted(set(got) ^ set(want))
  # ... and the GNMT-like encoder (one bidirectional level + unidirectional levels) with a three-layer gnmt decoder
  e = np.load(os.path.join(here, "golden", "ref_exec_nmt_encoder.npz"))
  B, S, V, E, H = [int(v) for v in e["dims"]]
  want = {str(n): e["gnmt_like/var/" + str(n)].shape for n in e["gnmt_like/var_names"]}
  m2 = _nmt_device_model(monkeypatch, "GNMTLikeEncoderWithEmbedding", V, E, H, 16, 3, 0, None)
  got = {k: v.shape for k, v in ck.model_variables(m2).items()}
  assert got == want, sorted(set(got) ^ set(want))
  dd = np.load(os.path.join(here, "golden", "ref_exec_nmt_decoder.npz"))
  B, S, T, V, E, H, M, U = [int(v) for v in dd["dims"]]
  case = "gnmt_v2_skip"
  want = {str(n): dd["%s/var/%s" % (case, n)].shape for n in dd[case + "/var_names"]
          if str(n) != "ForwardPass/encoder_outputs"}
  m3 = _nmt_device_model(monkeypatch, None, V, E, H, U, 0, 3, "gnmt_v2", M=M)
  got = {k: v.shape for k, v in ck.model_variables(m3).items()}
  # the device pads the output layer to a multiple of 8 rows and writes the logical V
  assert {k: v for k, v in got.items()} == want, sorted(set(got) ^ set(want))


def test_rnn_nmt_checkpoint_round_trip_and_row_blocks(monkeypatch, tmp_path):
  import torch
  m = _nmt_device_model(monkeypatch, "BidirectionalRNNEncoderWithEmbedding", 30, 64, 64, 128, 2, 2, "gnmt_v2",
                        dtype="mixed")
  arrays = ck.model_variables(m)
  k = "ForwardPass/rnn_decoder_with_attention/decoder/multi_rnn_cell/cell_1/lstm_cell/kernel"
  assert arrays[k].dtype == np.float16 and arrays[ck.MASTER_PREFIX + k].dtype == np.float32
  by = {p.name: p for p in m.store.params}
  pre = "ForwardPass/rnn_decoder_with_attention/multi_rnn_cell/cell_1/lstm_cell/"
  full = arrays[ck.MASTER_PREFIX + k]                       # [H + M + H, 4H]: layer below | attention | h
  np.testing.assert_array_equal(full[:64].T, by[pre + "wx_0"].master[0].numpy())
  np.testing.assert_array_equal(full[64:192].T, by[pre + "wx_1"].master[0].numpy())
  np.testing.assert_array_equal(full[192:].T, by[pre + "wh"].master[0].numpy())
  k0 = "ForwardPass/rnn_decoder_with_attention/decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/lstm_cell/kernel"
  c0 = "ForwardPass/rnn_decoder_with_attention/attention_cell/cell_0/"
  np.testing.assert_array_equal(arrays[ck.MASTER_PREFIX + k0][:64].T, by[c0 + "kernel_inputs"].master[0].numpy())
  np.testing.assert_array_equal(arrays[ck.MASTER_PREFIX + k0][64:].T, by[c0 + "kernel_attention_state"].master[0].numpy())
  # through a TensorBundle file and back into a second model with different values
  from openseq2seq_amd.utils import tensor_bundle
  prefix = str(tmp_path / "model.ckpt-0")
  tensor_bundle.write_bundle(prefix, arrays)
  before = {p.name: p.master.clone() for p in m.store.params}
  for p in m.store.params:
    p.master = torch.zeros_like(p.master)
  assert ck.load(m, prefix, restore_optimizer=False) == []
  for p in m.store.params:
    n = getattr(p, "logical_out", None) or p.master.shape[1 if p.master.dim() == 3 else 0]
    a, b = (p.master[:, :n], before[p.name][:, :n]) if p.master.dim() == 3 else (p.master, before[p.name])
    assert torch.equal(a, b), p.name              # the rows past the logical vocabulary come back as zeros
  # a file this repository wrote before the translation (device names) still loads
  legacy = {}
  for p in m.store.params:
    for n, a in ck.export_param(p.name, p.shape, p.kind, before[p.name].numpy(), getattr(p, "logical_out", None)):
      legacy[n] = a
  np.savez(str(tmp_path / "old.npz"), **legacy)
  for p in m.store.params:
    p.master = torch.zeros_like(p.master)
  assert ck.load(m, str(tmp_path / "old.npz"), restore_optimizer=False) == []
  for p in m.store.params:
    n = getattr(p, "logical_out", None) or p.master.shape[1 if p.master.dim() == 3 else 0]
    a, b = (p.master[:, :n], before[p.name][:, :n]) if p.master.dim() == 3 else (p.master, before[p.name])
    assert torch.equal(a, b), p.name              # the rows past the logical vocabulary come back as zeros


def test_tacotron_decoder_checkpoints_carry_the_reference_graphs_variables(monkeypatch, tmp_path):
  """The same for Tacotron 2: tf.trainable_variables() of the reference's executed Tacotron2Encoder + Tacotron2Decoder
  + "both"-mode heads (tests/golden/ref_exec_tacotron_full.npz) against what utils/checkpoint.py writes for the device
  model of that configuration — the decoder-step variables under dynamic_decode's 'decoder' scope, the attention cell's
  kernel as one [prenet + attention + H, 4H] matrix, the K = 1 conv1d layers (location attention's memory layer and
  location_dense, post_net_proj) and the location convolution [K, 1, F] at rank 3, the stop-token layer with its one
  logical unit. Only the encoder's cuDNN LSTM differs: one opaque buffer in TensorFlow, own names here."""
  import os
  import torch
  from openseq2seq_amd.encoders import Tacotron2Encoder
  from openseq2seq_amd.decoders import Tacotron2Decoder
  sys_path = os.path.dirname(os.path.abspath(__file__))
  import sys
  sys.path.insert(0, sys_path)
  import ref_exec_util as rx
  from test_tacotron_e2e_gpu import CONVS, POST
  d = np.load(os.path.join(sys_path, "golden", "ref_exec_tacotron_full.npz"))
  C = rx.gen.TACO_FULL
  V, E, Henc, H, NM, NG, PRE, U = [C[k] for k in ("V", "E", "Henc", "H", "NM", "NG", "pre", "U")]
  store = _cpu_store(monkeypatch)
  enc = Tacotron2Encoder({"cnn_dropout_prob": 0.0, "rnn_dropout_prob": 0.0, "src_emb_size": E, "conv_layers": CONVS,
                          "activation_fn": "relu", "num_rnn_layers": 1, "rnn_cell_dim": Henc, "use_cudnn_rnn": True,
                          "rnn_type": "CudnnLSTM", "rnn_unidirectional": False, "dtype": "mixed"}, None, mode="train")
  enc.build(store, src_vocab_size=V, num_style_features=NM)
  dec = Tacotron2Decoder({"attention_layer_size": U, "attention_type": "location", "attention_bias": True,
                          "decoder_cell_units": H, "decoder_cell_type": "LSTMCell", "decoder_layers": 2,
                          "dropout_prob": 0.0, "enable_prenet": True, "prenet_layers": 2, "prenet_units": PRE,
                          "enable_postnet": True, "postnet_keep_dropout_prob": 1.0, "postnet_conv_layers": POST,
                          "dtype": "mixed"}, None, mode="train")
  dec.build(store, memory_dim=enc.output_dim, num_audio_features={"mel": NM, "magnitude": NG}, exp_mag=True)
  store.finalize()

  class M(object):
    params = {"dtype": "float32"}
  M.store = store
  m = M()
  opaque = lambda n: "/tacotron2_encoder/weight_" in n or "/tacotron2_encoder/bias_" in n or "/cudnn_rnn/" in n  # noqa: E731
  want = {str(n): tuple(int(v) for v in d["shape/" + str(n)]) for n in d["var_names"] if not opaque(str(n))}
  arrays = ck.model_variables(m)
  got = {k: v.shape for k, v in arrays.items() if not opaque(k) and "/bn/moving_" not in k}
  assert got == want, (sorted(set(got) ^ set(want)), [(k, got[k], want[k]) for k in got if k in want and got[k] != want[k]])
  # row blocks of the attention cell: prenet output | attention context | h
  by = {p.name: p for p in store.params}
  k0 = arrays["ForwardPass/tacotron_2_decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/lstm_cell/kernel"]
  c0 = "ForwardPass/tacotron_2_decoder/attention_wrapper/cell_0/"
  np.testing.assert_array_equal(k0[:PRE].T, by[c0 + "kernel_inputs"].master[0].numpy())
  np.testing.assert_array_equal(k0[PRE:].T, by[c0 + "kernel_attention_state"].master[0].numpy())
  # file round trip
  from openseq2seq_amd.utils import tensor_bundle
  prefix = str(tmp_path / "model.ckpt-0")
  tensor_bundle.write_bundle(prefix, arrays)
  before = {p.name: p.master.clone() for p in store.params}
  for p in store.params:
    p.master = torch.zeros_like(p.master)
  assert ck.load(m, prefix, restore_optimizer=False) == []
  # the encoder's CudnnLSTM is written as TF's cuDNN Saveable writes it (canonical kernel / bias per direction):
  # the two cuDNN bias vectors are stored as their SUM and come back as two halves of it
  cud = ck.cudnn_groups(store.params)
  assert len(cud) == 2
  halves = {}
  for parts in cud.values():
    halves[parts["bias"].name] = parts["bias_h"].name
    halves[parts["bias_h"].name] = parts["bias"].name
    assert any(k.endswith("/cudnn_compatible_lstm_cell/kernel") for k in arrays)
  now = {p.name: p.master for p in store.params}
  for p in store.params:
    n = getattr(p, "logical_out", None)
    if p.name in halves:
      assert torch.equal(p.master + now[halves[p.name]], before[p.name] + before[halves[p.name]]), p.name
      assert torch.equal(p.master, now[halves[p.name]]), p.name
    elif n is None:
      assert torch.equal(p.master, before[p.name]), p.name
    elif p.master.dim() == 3:
      assert torch.equal(p.master[:, :n], before[p.name][:, :n]) and not p.master[:, n:].any(), p.name
    else:
      assert torch.equal(p.master[:n], before[p.name][:n]) and not p.master[n:].any(), p.name


def test_sample_norm_vectors_are_half_with_a_master_twin_in_a_mixed_graph():
  """tf.contrib.layers.layer_norm / instance_norm create gamma / beta in the dtype of their input
  (conv_ln_actv / conv_in_actv, parts/cnns/conv_blocks.py:234-309): DT_HALF + an fp32 twin in a mixed-precision graph
  of the reference, unlike the explicitly fp32 BatchNorm vectors."""
  import torch
  from openseq2seq_amd.utils import checkpoint as ck

  class P(object):
    def __init__(self, name, kind, arr):
      self.name, self.kind, self.shape, self.master = name, kind, arr.shape, torch.from_numpy(arr)

  g = np.linspace(0.5, 1.5, 8).astype(np.float32)
  names = ["ForwardPass/w2l_encoder/LayerNorm/gamma", "ForwardPass/w2l_encoder/LayerNorm_3/beta",
           "ForwardPass/w2l_encoder/InstanceNorm_1/gamma", "ForwardPass/w2l_encoder/conv11/bn/gamma",
           "ForwardPass/transformer_encoder/layer_normalization/layer_norm_scale"]

  class Store(object):
    params = [P(n, "vector", g) for n in names]
    state = {}

  class M(object):
    store = Store()
    params = {"dtype": "mixed"}

  out = ck.model_variables(M())
  for n in names[:3]:
    assert out[n].dtype == np.float16 and out[ck.MASTER_PREFIX + n].dtype == np.float32, n
  for n in names[3:]:
    assert out[n].dtype == np.float32 and ck.MASTER_PREFIX + n not in out, n


def test_optimizer_slots_follow_the_variables_when_the_creation_order_changes():
  """OS2S/opt/m1, m2 are flat buffers and t_v one value per variable; their total does not depend on
  the creation order, so a file written under another order used to load with every moment on the wrong variable.
  The file now records (name hash, offset, size) per variable: identical layout -> flat copy; a permuted one ->
  slot by slot; no record (files from before round 6) -> moments are NOT restored, with a warning."""
  import warnings
  import torch
  from openseq2seq_amd.utils import checkpoint as ck

  class P(object):
    def __init__(self, name, numel):
      self.name, self.numel, self.offset = name, numel, 0

  def store(order, chunk=8):
    class S(object):
      pass
    s = S()
    s.params = [P(n, k) for n, k in order]
    off = 0
    for p in s.params:
      p.offset = off
      off += -(-p.numel // chunk) * chunk
    s.m1 = torch.zeros(off)
    s.m2 = torch.zeros(off)
    s.t_v = torch.zeros(len(s.params))
    return s

  class Op(object):
    state = torch.zeros(4)

  a = [("dec/kv/kernel", 24), ("dec/layer_0/q", 10), ("dec/layer_0/ffn", 17), ("enc/emb", 5)]
  old = store(a)
  for i, p in enumerate(old.params):                 # slot value = 100 * variable index + element index
    old.m1[p.offset:p.offset + p.numel] = 100.0 * i + torch.arange(p.numel)
    old.m2[p.offset:p.offset + p.numel] = -(100.0 * i + torch.arange(p.numel))
    old.t_v[i] = 7.0 + i
  data = {"OS2S/opt/m1": old.m1.numpy(), "OS2S/opt/m2": old.m2.numpy(), "OS2S/opt/t_v": old.t_v.numpy(),
          "OS2S/opt/state": np.arange(4, dtype=np.float32), "OS2S/opt/layout": ck.slot_layout(old)}
  same = store(a)
  assert ck.restore_slots(same, Op(), data) == "flat"
  assert torch.equal(same.m1, old.m1) and torch.equal(same.t_v, old.t_v)
  b = [a[1], a[2], a[0], a[3]]                       # the round-4 order: per-layer variables first
  new = store(b)
  assert new.m1.numel() == old.m1.numel()            # the trap: same total, different layout
  assert ck.restore_slots(new, Op(), data) == "by_name"
  for i, p in enumerate(new.params):
    j = [n for n, _ in a].index(p.name)
    want = 100.0 * j + torch.arange(p.numel)
    assert torch.equal(new.m1[p.offset:p.offset + p.numel], want), p.name
    assert torch.equal(new.m2[p.offset:p.offset + p.numel], -want), p.name
    assert float(new.t_v[i]) == 7.0 + j
  legacy = dict(data)
  del legacy["OS2S/opt/layout"]
  cold = store(b)
  op = Op()
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    assert ck.restore_slots(cold, op, legacy, "old.ckpt") == "skipped"
  assert any("NOT restored" in str(x.message) for x in w)
  assert float(cold.m1.abs().sum()) == 0.0 and float(op.state[3]) == 3.0     # scalars restored, moments not


def _tf_cudnn_compatible_gru(x, tf, H):
  """tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell over a sequence (its call(): gates = sigmoid([x, h] . gates/kernel +
  gates/bias), r, u = split(gates); candidate = tanh(x . Wc + bc + r * (h . Rc + bRc)); h' = u h + (1 - u) candidate)."""
  sig = lambda a: 1.0 / (1.0 + np.exp(-a))
  B, T, _ = x.shape
  h = np.zeros((B, H))
  out = []
  for t in range(T):
    g = sig(np.concatenate([x[:, t], h], 1) @ tf["gates/kernel"] + tf["gates/bias"])
    r, u = g[:, :H], g[:, H:]
    c = np.tanh(x[:, t] @ tf["candidate/input_projection/kernel"] + tf["candidate/input_projection/bias"] +
                r * (h @ tf["candidate/hidden_projection/kernel"] + tf["candidate/hidden_projection/bias"]))
    h = u * h + (1 - u) * c
    out.append(h)
  return np.stack(out, 1)


def _tf_cudnn_compatible_lstm(x, tf, H):
  """CudnnCompatibleLSTMCell = LSTMBlockCell(forget_bias=0): i, ci, f, o = split([x, h] . kernel + bias);
  c' = sigmoid(f) c + sigmoid(i) tanh(ci); h' = sigmoid(o) tanh(c')."""
  sig = lambda a: 1.0 / (1.0 + np.exp(-a))
  B, T, _ = x.shape
  h, c = np.zeros((B, H)), np.zeros((B, H))
  out = []
  for t in range(T):
    z = np.concatenate([x[:, t], h], 1) @ tf["kernel"] + tf["bias"]
    i, ci, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    c = sig(f) * c + sig(i) * np.tanh(ci)
    h = sig(o) * np.tanh(c)
    out.append(h)
  return np.stack(out, 1)


def test_cudnn_layers_exchange_canonical_checkpoint_tensors():
  """DeepSpeech2's CudnnGRU stack and Tacotron 2's CudnnLSTM encoder are written as the tensors TF's cuDNN Saveable
  writes (CudnnCompatible{GRU,LSTM}Cell layout: utils/checkpoint.py, cuDNN block). The exported tensors, run through
  the CudnnCompatible cells' own equations, must reproduce what the device-layout parameters compute (oracle/rnn.py,
  cuDNN gate form) — and come back unchanged in function (gate biases return as two halves of their sum)."""
  import torch
  from collections import namedtuple
  from oracle import rnn as orn
  rng = np.random.RandomState(3)
  B, T, n_in, H = 3, 7, 10, 6
  x = rng.randn(B, T, n_in)
  for kind, G, tf_cell in (("gru", 3, _tf_cudnn_compatible_gru), ("lstm", 4, _tf_cudnn_compatible_lstm)):
    wx = rng.randn(1, G * H, n_in) * 0.4
    wh = rng.randn(1, G * H, H) * 0.4
    bx, bh = rng.randn(G * H) * 0.3, rng.randn(G * H) * 0.3
    tfv = ck.cudnn_to_canonical(wx, wh, bx, bh)
    if kind == "gru":
      assert tfv["gates/kernel"].shape == (n_in + H, 2 * H) and tfv["candidate/hidden_projection/kernel"].shape == (H, H)
    else:
      assert tfv["kernel"].shape == (n_in + H, 4 * H) and tfv["bias"].shape == (4 * H,)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    ref = orn.cudnn_rnn(kind, t(x), None, t(wx[0]), t(wh[0]), t(bx), t(bh)).numpy()
    np.testing.assert_allclose(tf_cell(x, tfv, H), ref, rtol=1e-9, atol=1e-10)
    back = ck.canonical_to_cudnn(lambda s: tfv.get(s), G, n_in, H)
    assert back is not None
    wx2, wh2, bx2, bh2 = back
    np.testing.assert_array_equal(wx2, wx)
    np.testing.assert_array_equal(wh2, wh)
    ref2 = orn.cudnn_rnn(kind, t(x), None, t(wx2[0]), t(wh2[0]), t(bx2), t(bh2)).numpy()
    np.testing.assert_allclose(ref2, ref, rtol=1e-9, atol=1e-10)
    assert ck.canonical_to_cudnn(lambda s: None, G, n_in, H) is None
  # names: the scopes TF's Saveable writes under, per direction and layer
  P = namedtuple("P", "name shape")
  names = []
  for l in range(2):
    for tag in ("fw", "bw"):
      base = "ForwardPass/ds2_encoder/cudnn_gru/layer_%d/%s/" % (l, tag)
      names += [P(base + "wx_0", (1, 18, 10)), P(base + "wh", (1, 18, 6)), P(base + "bias", (18,)), P(base + "bias_h", (18,))]
  names.append(P("ForwardPass/tacotron2_encoder/cudnn_rnn/layer_0/fw/wx_0", (1, 24, 10)))     # incomplete: not a group
  groups = ck.cudnn_groups(names)
  assert sorted(groups) == [("ForwardPass/ds2_encoder/cudnn_gru", l, tag) for l in range(2) for tag in ("bw", "fw")]
  assert ck.cudnn_canonical_prefix("ForwardPass/ds2_encoder/cudnn_gru", 1, "bw", True, 3) == \
      "ForwardPass/ds2_encoder/cudnn_gru/stack_bidirectional_rnn/cell_1/bidirectional_rnn/bw/cudnn_compatible_gru_cell"
  assert ck.cudnn_canonical_prefix("S/cudnn_rnn", 0, "fw", False, 4) == "S/cudnn_rnn/rnn/multi_rnn_cell/cell_0/cudnn_compatible_lstm_cell"
