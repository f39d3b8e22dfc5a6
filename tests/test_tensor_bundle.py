"""TensorFlow V2 checkpoint files (utils/tensor_bundle.py) — the container of the reference's
tf.train.Saver (open_seq2seq/utils/funcs.py:117-144) and of its name + shape restore
(utils/helpers.py:462-553). No TensorFlow here: the known answers are the published CRC-32C
vectors (RFC 3720 B.4), the LevelDB table layout assembled by hand in this file (independently of
the writer), and writer -> reader round trips."""
import os
import struct

import numpy as np
import pytest

from openseq2seq_amd.utils import checkpoint as ck
from openseq2seq_amd.utils import tensor_bundle as tb


def test_crc32c_known_answers():
  assert tb.crc32c(b"123456789") == 0xe3069283
  assert tb.crc32c(bytes(32)) == 0x8a9136aa
  assert tb.crc32c(b"\xff" * 32) == 0x62a8ab43
  assert tb.crc32c(bytes(range(32))) == 0x46dd794e
  assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
  blob = np.random.RandomState(0).bytes(10007)
  for cut in (0, 1, 7, 8, 4099, 10007):
    assert tb.crc32c(blob[cut:], tb.crc32c(blob[:cut])) == tb.crc32c(blob)
  arr = np.frombuffer(blob[:10000], np.float32)
  assert tb.crc32c(arr) == tb.crc32c(blob[:10000])
  for c in (0, 1, 0xe3069283, 0xffffffff):
    assert tb.unmask_crc(tb.mask_crc(c)) == c and tb.mask_crc(c) != c


def _vi(v):
  out = b""
  while v >= 0x80:
    out += bytes([(v & 0x7f) | 0x80])
    v >>= 7
  return out + bytes([v])


def _trailer(contents, ctype=0):
  return bytes([ctype]) + struct.pack("<I", tb.mask_crc(tb.crc32c(contents + bytes([ctype]))))


def _hand_index(path, ctype=0):
  """A two-data-block table written byte by byte: block 0 holds the header and 'a/kernel' +
  'a/kernel_2' (second key prefix-compressed: shared = 8), block 1 holds 'b' with its own restart."""
  f32 = np.arange(6, dtype="<f4").reshape(2, 3)
  i64 = np.asarray(1234567890123, "<i8")
  f16 = np.asarray([1.5, -2.0], "<f2")
  data = f32.tobytes() + i64.tobytes() + f16.tobytes()

  def entry(dtype, shape_bytes, offset, size, raw):
    e = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(shape_bytes)) + shape_bytes
    if offset:
      e += b"\x20" + _vi(offset)
    e += b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", tb.mask_crc(tb.crc32c(raw)))
    return e
  e_f32 = entry(1, b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03", 0, 24, f32.tobytes())
  e_i64 = entry(9, b"", 24, 8, i64.tobytes())
  e_f16 = entry(19, b"\x12\x02\x08\x02", 32, 4, f16.tobytes())
  header = b"\x08\x01\x1a\x02\x08\x01"
  b0 = (_vi(0) + _vi(0) + _vi(len(header)) + header
        + _vi(0) + _vi(8) + _vi(len(e_f32)) + b"a/kernel" + e_f32
        + _vi(8) + _vi(2) + _vi(len(e_i64)) + b"_2" + e_i64
        + struct.pack("<II", 0, 1))
  b1 = _vi(0) + _vi(1) + _vi(len(e_f16)) + b"b" + e_f16 + struct.pack("<II", 0, 1)
  out = b0 + _trailer(b0, ctype)
  h0 = _vi(0) + _vi(len(b0))
  off1 = len(out)
  out += b1 + _trailer(b1)
  h1 = _vi(off1) + _vi(len(b1))
  meta = struct.pack("<II", 0, 1)
  offm = len(out)
  out += meta + _trailer(meta)
  # index block: separator keys 'a/kernel_2' (last key of block 0) and 'c' (a short successor)
  idx = (_vi(0) + _vi(10) + _vi(len(h0)) + b"a/kernel_2" + h0
         + _vi(0) + _vi(1) + _vi(len(h1)) + b"c" + h1 + struct.pack("<II", 0, 1))
  offi = len(out)
  out += idx + _trailer(idx)
  footer = _vi(offm) + _vi(len(meta)) + _vi(offi) + _vi(len(idx))
  footer += bytes(40 - len(footer)) + struct.pack("<II", 0x8b80fb57, 0xdb477524)
  out += footer
  with open(path + ".index", "wb") as f:
    f.write(out)
  with open(path + ".data-00000-of-00001", "wb") as f:
    f.write(data)
  return {"a/kernel": f32, "a/kernel_2": i64, "b": f16}


def test_reads_hand_assembled_table(tmp_path):
  prefix = str(tmp_path / "model.ckpt-5")
  want = _hand_index(prefix)
  r = tb.BundleReader(prefix)
  assert sorted(r.keys()) == sorted(want)
  assert r.get_variable_to_shape_map() == {"a/kernel": [2, 3], "a/kernel_2": [], "b": [2]}
  assert r.get_variable_to_dtype_map() == {"a/kernel": "float32", "a/kernel_2": "int64", "b": "float16"}
  for k, v in want.items():
    got = r.get_tensor(k)
    assert got.dtype == v.dtype and got.shape == v.shape
    np.testing.assert_array_equal(got, v)
  assert r.has_tensor("b") and not r.has_tensor("c")
  with pytest.raises(KeyError):
    r.get_tensor("c")


def test_round_trip_many_blocks(tmp_path):
  rng = np.random.RandomState(1)
  arrays = {"global_step": np.asarray(77, np.int64), "flag": np.asarray([True, False, True])}
  for i in range(300):
    shape = tuple(rng.randint(1, 6, size=rng.randint(0, 4)))
    arrays["ForwardPass/w2l_encoder/conv%d%d/kernel" % (i // 10, i % 10)] = np.asarray(rng.randn(*shape), np.float32)
    arrays[ck.MASTER_PREFIX + "layer_%03d/bias" % i] = rng.randn(7).astype(np.float16)
  prefix = str(tmp_path / "sub" / "model.ckpt-77")
  tb.write_bundle(prefix, arrays)
  # small blocks: the index block gets many entries and restarts
  items = tb.read_table(prefix + ".index")
  tb.write_table(prefix + ".index", items, block_size=512)
  assert tb.read_table(prefix + ".index") == items
  assert [k for k, _ in items] == sorted(k for k, _ in items) and items[0][0] == b""
  r = tb.BundleReader(prefix)
  assert set(r.keys()) == set(arrays)
  for k, v in arrays.items():
    got = r[k]
    assert got.dtype == v.dtype and got.shape == v.shape
    np.testing.assert_array_equal(got, v)
  assert ck.read_step(prefix) == 77


def test_bfloat16_entries_widen_to_float32(tmp_path):
  prefix = str(tmp_path / "m")
  vals = np.asarray([1.0, -3.5, 0.15625], np.float32)
  raw = (vals.view(np.uint32) >> 16).astype("<u2")
  e = tb.BundleEntry()
  e.dtype, e.shape, e.offset, e.size = tb.DT_BFLOAT16, (3,), 0, 6
  e.crc32c = tb.mask_crc(tb.crc32c(raw))
  tb.write_table(prefix + ".index", [(b"", b"\x08\x01\x1a\x02\x08\x01"), (b"w", e.serialize())])
  open(prefix + ".data-00000-of-00001", "wb").write(raw.tobytes())
  np.testing.assert_array_equal(tb.BundleReader(prefix)["w"], vals)


def test_corruption_and_misuse_are_reported(tmp_path):
  prefix = str(tmp_path / "model.ckpt-1")
  tb.write_bundle(prefix, {"w": np.arange(100, dtype=np.float32)})
  # the reference's hint for V2 files addressed with an extension (helpers.py:529-537)
  with pytest.raises(ValueError, match="PREFIX"):
    tb.BundleReader(prefix + ".index")
  with pytest.raises(ValueError, match="not found"):
    tb.BundleReader(str(tmp_path / "nothing"))
  data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
  data[17] ^= 0x40
  open(prefix + ".data-00000-of-00001", "wb").write(data)
  with pytest.raises(ValueError, match="Data loss"):
    tb.BundleReader(prefix).get_tensor("w")
  assert tb.BundleReader(prefix, verify=False).get_tensor("w").shape == (100,)
  idx = bytearray(open(prefix + ".index", "rb").read())
  idx[3] ^= 0x01
  open(prefix + ".index", "wb").write(idx)
  with pytest.raises(ValueError, match="Data loss"):
    tb.BundleReader(prefix)
  idx[-1] ^= 0xff
  open(prefix + ".index", "wb").write(idx)
  with pytest.raises(ValueError, match="magic"):
    tb.BundleReader(prefix)
  # SNAPPY-compressed index blocks: the reference prints this diagnosis (helpers.py:525-528)
  p2 = str(tmp_path / "snappy")
  _hand_index(p2, ctype=1)
  with pytest.raises(ValueError, match="SNAPPY"):
    tb.BundleReader(p2)
  with pytest.raises(TypeError):
    tb.write_bundle(str(tmp_path / "bad"), {"s": np.asarray(["text"])})


def test_restore_by_name_and_shape_from_bundle(tmp_path):
  """helpers.py:462-553 on a reference-written checkpoint: conv kernel in TF layout under the
  FP32 master-copy name only, dense kernel with the logical output width."""
  rng = np.random.RandomState(2)
  conv_tf = rng.randn(11, 24, 40).astype(np.float32)               # [K, Cin, Cout]
  fc_tf = rng.randn(48, 29).astype(np.float32)                     # [H, V]
  conv_name = "ForwardPass/w2l_encoder/conv11/kernel"
  fc_name = "ForwardPass/fully_connected_ctc_decoder/fully_connected/kernel"
  prefix = str(tmp_path / "model.ckpt-9")
  tb.write_bundle(prefix, {ck.MASTER_PREFIX + conv_name: conv_tf, fc_name: fc_tf.astype(np.float16),
                           "global_step": np.asarray(9, np.int64)})
  data = ck.open_checkpoint(prefix)
  w = ck.import_param(conv_name, (11, 40, 24), "conv", data)
  np.testing.assert_array_equal(w, np.transpose(conv_tf, (0, 2, 1)))
  fc = ck.import_param(fc_name, (1, 32, 48), "conv", data, logical_out=29)
  assert fc.shape == (1, 32, 48) and fc.dtype == np.float32
  np.testing.assert_array_equal(fc[0, :29], fc_tf.astype(np.float16).astype(np.float32).T)
  assert not fc[0, 29:].any()
  assert ck.import_param("ForwardPass/missing/kernel", (1, 4, 4), "conv", data) is None
  # latest_checkpoint / read_step as run.py uses them
  open(os.path.join(str(tmp_path), "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-9"\n')
  assert ck.latest_checkpoint(str(tmp_path)) == prefix and ck.read_step(prefix) == 9
