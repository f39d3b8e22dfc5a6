"""Reader and writer for TensorFlow V2 checkpoints ("TensorBundle": `<prefix>.index` +
`<prefix>.data-NNNNN-of-MMMMM`), the files tf.train.Saver produces for the reference
(open_seq2seq/utils/funcs.py:117-144, utils/hooks.py:227-236) and that its restore path reads
through tf.train.NewCheckpointReader (utils/helpers.py:462-553). TensorFlow is not a dependency
here: the format is restated from its published definition (tensorflow/core/util/tensor_bundle,
tensorflow/core/lib/io/table — the LevelDB table format, tensorflow/core/protobuf/tensor_bundle.proto,
tensorflow/core/framework/{types,tensor_shape}.proto; TF 1.13, the version the reference pins).

  <prefix>.index     an immutable sorted string table:
      data blocks | metaindex block | index block | 48-byte footer
    block     = entries, restart offsets (uint32 each), number of restarts (uint32)
                then a 5-byte trailer: compression type (0 = none, 1 = snappy), masked CRC-32C of
                contents + type
    entry     = varint shared-key-bytes, varint unshared-key-bytes, varint value-bytes,
                unshared key bytes, value bytes        (keys are prefix-compressed inside a block)
    footer    = metaindex BlockHandle, index BlockHandle (varint offset, varint size), zero padding
                to 40 bytes, magic 0xdb4775248b80fb57 (little-endian)
    key ""    -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: VersionDef}
    key name  -> BundleEntryProto   {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size,
                                     6: masked crc32c (fixed32), 7: slices (partitioned variables)}
  <prefix>.data-*    the tensors' bytes, little-endian, row-major, at [offset, offset + size)

Partitioned variables (entries with slices), string / resource / variant tensors and
snappy-compressed index blocks are rejected with an explicit error — the reference's checkpoints
(dense float / half / int64 variables saved by one Saver) use none of them.

Parity: no TensorFlow and no .index fixture exists in this environment or under /root/reference,
so the byte format is "parity unpinned" against real TF output; tests/test_tensor_bundle.py pins
the pieces that have published known answers (CRC-32C vectors, the table magic, LevelDB's
documented block layout via a hand-assembled file) and the writer -> reader round trip.
"""
from __future__ import absolute_import, division, print_function

import ctypes
import os
import struct

import numpy as np

from .. import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_BYTES = 48
BLOCK_TRAILER_BYTES = 5
BLOCK_RESTART_INTERVAL = 16
BLOCK_SIZE = 262144          # table::Options::block_size in TF

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'),
           5: np.dtype('<i2'), 6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'),
           17: np.dtype('<u2'), 19: np.dtype('<f2'), 22: np.dtype('<u4'), 23: np.dtype('<u8')}
DT_BFLOAT16 = 14             # returned as float32 (numpy has no bfloat16)
_DTYPE_NAMES = {7: "string", 8: "complex64", 18: "complex128", 20: "resource", 21: "variant"}
_ENUM_OF = {v: k for k, v in _DTYPES.items()}


def crc32c(data, init=0):
  """CRC-32C of a bytes-like / contiguous ndarray (os2s_crc32c in libos2s_hip.so, host code)."""
  f = _lib.bind("os2s_crc32c", [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t], ctypes.c_uint32)
  if isinstance(data, np.ndarray):
    a = np.ascontiguousarray(data)
    return int(f(init, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
  b = bytes(data)
  return int(f(init, b, len(b)))


def mask_crc(crc):
  """crc32c::Mask: rotate right by 15 and add a constant (CRCs of data that embeds CRCs)."""
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def unmask_crc(masked):
  rot = (masked - 0xa282ead8) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / the few protobuf messages involved ---------------------------------------------------
def _put_varint(out, v):
  v &= (1 << 64) - 1
  while v >= 0x80:
    out.append((v & 0x7f) | 0x80)
    v >>= 7
  out.append(v)


def _varint(v):
  out = bytearray()
  _put_varint(out, v)
  return bytes(out)


def _get_varint(buf, pos):
  shift = result = 0
  while True:
    if pos >= len(buf):
      raise ValueError("truncated varint")
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7
    if shift > 63:
      raise ValueError("varint too long")


def _proto_fields(buf):
  """Yields (field number, wire type, value) of one serialized message; value is an int for
  varint / fixed fields and a memoryview slice for length-delimited ones."""
  pos, n = 0, len(buf)
  while pos < n:
    tag, pos = _get_varint(buf, pos)
    field, wt = tag >> 3, tag & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wt == 2:
      ln, pos = _get_varint(buf, pos)
      if pos + ln > n:
        raise ValueError("truncated length-delimited field")
      v = buf[pos:pos + ln]
      pos += ln
    elif wt == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError("unsupported protobuf wire type %d" % wt)
    yield field, wt, v


def _signed64(v):
  return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf):
  dims = []
  for field, _, v in _proto_fields(buf):
    if field == 2:                       # repeated Dim
      size = 0
      for f2, _, v2 in _proto_fields(v):
        if f2 == 1:
          size = _signed64(v2)
      dims.append(size)
    elif field == 3 and v:
      raise ValueError("tensor of unknown rank in checkpoint")
  return tuple(dims)


def _encode_shape(shape):
  out = bytearray()
  for d in shape:
    dim = bytearray()
    if d:
      dim.append(0x08)
      _put_varint(dim, int(d))
    out.append(0x12)
    _put_varint(out, len(dim))
    out += dim
  return bytes(out)


class BundleEntry(object):
  __slots__ = ("dtype", "shape", "shard_id", "offset", "size", "crc32c", "sliced")

  def __init__(self):
    self.dtype, self.shape, self.shard_id, self.offset, self.size = 0, (), 0, 0, 0
    self.crc32c, self.sliced = 0, False

  @classmethod
  def parse(cls, buf):
    e = cls()
    for field, _, v in _proto_fields(buf):
      if field == 1:
        e.dtype = v
      elif field == 2:
        e.shape = _parse_shape(v)
      elif field == 3:
        e.shard_id = v
      elif field == 4:
        e.offset = v
      elif field == 5:
        e.size = v
      elif field == 6:
        e.crc32c = v
      elif field == 7:
        e.sliced = True
    return e

  def serialize(self):
    out = bytearray()
    if self.dtype:
      out.append(0x08)
      _put_varint(out, self.dtype)
    shp = _encode_shape(self.shape)        # BundleWriter always sets the shape message
    out.append(0x12)
    _put_varint(out, len(shp))
    out += shp
    if self.shard_id:
      out.append(0x18)
      _put_varint(out, self.shard_id)
    if self.offset:
      out.append(0x20)
      _put_varint(out, self.offset)
    if self.size:
      out.append(0x28)
      _put_varint(out, self.size)
    if self.crc32c:
      out.append(0x35)
      out += struct.pack('<I', self.crc32c)
    return bytes(out)


# ---- the table file -------------------------------------------------------------------------------
def _block_entries(block):
  """(key, value) pairs of one block's contents (restart array included, trailer excluded)."""
  if len(block) < 4:
    raise ValueError("table block too small")
  num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * num_restarts
  if limit < 0:
    raise ValueError("bad restart count in table block")
  pos, key = 0, b""
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise ValueError("corrupt entry in table block")
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    yield key, block[pos + 0:pos + vlen]
    pos += vlen


def _read_block(data, offset, size, verify):
  end = offset + size + BLOCK_TRAILER_BYTES
  if end > len(data):
    raise ValueError("table block handle points past the end of the file")
  ctype = data[offset + size]
  if verify:
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(bytes(data[offset:offset + size + 1])):
      raise ValueError("Data loss: block checksum mismatch in checkpoint index")
  if ctype == 1:
    raise ValueError("corrupted compressed block contents: the checkpoint index is SNAPPY-compressed, "
                     "which this reader does not decode")
  if ctype != 0:
    raise ValueError("unknown table block compression type %d" % ctype)
  return data[offset:offset + size]


def read_table(path, verify=True):
  """All (key bytes, value bytes) of a table file, in key order."""
  with open(path, "rb") as f:
    data = memoryview(f.read())
  if len(data) < FOOTER_BYTES:
    raise ValueError("Data loss: %s is too short to be a checkpoint index" % path)
  footer = data[len(data) - FOOTER_BYTES:]
  if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
    raise ValueError("Data loss: %s is not a table file (bad magic number); a V2 checkpoint is "
                     "addressed by its filename PREFIX, without .index / .data-*" % path)
  pos = 0
  _, pos = _get_varint(footer, pos)           # metaindex handle (unused by TensorBundle)
  _, pos = _get_varint(footer, pos)
  ioff, pos = _get_varint(footer, pos)
  isize, pos = _get_varint(footer, pos)
  out = []
  for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
    boff, p2 = _get_varint(handle, 0)
    bsize, _ = _get_varint(handle, p2)
    for k, v in _block_entries(_read_block(data, boff, bsize, verify)):
      out.append((k, bytes(v)))
  return out


class _BlockBuilder(object):
  def __init__(self):
    self.buf = bytearray()
    self.restarts = [0]
    self.counter = 0
    self.last_key = b""

  def add(self, key, value):
    shared = 0
    if self.counter < BLOCK_RESTART_INTERVAL:
      m = min(len(key), len(self.last_key))
      while shared < m and key[shared] == self.last_key[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.counter = 0
    _put_varint(self.buf, shared)
    _put_varint(self.buf, len(key) - shared)
    _put_varint(self.buf, len(value))
    self.buf += key[shared:]
    self.buf += value
    self.last_key = key
    self.counter += 1

  def size_estimate(self):
    return len(self.buf) + 4 * len(self.restarts) + 4

  def finish(self):
    return bytes(self.buf) + b"".join(struct.pack('<I', r) for r in self.restarts) + \
        struct.pack('<I', len(self.restarts))


def write_table(path, items, block_size=BLOCK_SIZE):
  """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
  out = bytearray()
  index = _BlockBuilder()

  def emit(contents):
    off = len(out)
    out.extend(contents)
    out.append(0)                                               # kNoCompression
    out.extend(struct.pack('<I', mask_crc(crc32c(contents + b"\x00"))))
    return _varint(off) + _varint(len(contents))

  cur = _BlockBuilder()
  last = None
  for key, value in items:
    if last is not None and key <= last:
      raise ValueError("table keys must be strictly increasing")
    cur.add(key, value)
    last = key
    if cur.size_estimate() >= block_size:
      index.add(cur.last_key, emit(cur.finish()))               # separator = the block's last key
      cur = _BlockBuilder()
  if cur.counter or not index.counter:
    index.add(cur.last_key, emit(cur.finish()))
  meta_handle = emit(_BlockBuilder().finish())
  index_handle = emit(index.finish())
  footer = meta_handle + index_handle
  footer += b"\x00" * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out.extend(footer)
  with open(path, "wb") as f:
    f.write(out)


# ---- the bundle -----------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
  return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def is_bundle(prefix):
  return os.path.exists(prefix + ".index")


class BundleReader(object):
  """tf.train.NewCheckpointReader for a V2 checkpoint prefix: has_tensor, get_tensor,
  get_variable_to_shape_map, get_variable_to_dtype_map."""

  def __init__(self, prefix, verify=True):
    if not os.path.exists(prefix + ".index"):
      hint = ""
      for ext in (".index", ".meta"):
        if prefix.endswith(ext):
          hint = " (pass the filename PREFIX: %s)" % prefix[:-len(ext)]
      raise ValueError("Error in loading checkpoint: %s.index not found%s" % (prefix, hint))
    self.prefix = prefix
    self.verify = verify
    self.entries = {}
    self.num_shards = 1
    header_seen = False
    for key, value in read_table(prefix + ".index", verify):
      if key == b"":
        header_seen = True
        for field, _, v in _proto_fields(memoryview(value)):
          if field == 1:
            self.num_shards = v
          elif field == 2 and v != 0:
            raise ValueError("big-endian checkpoint")
      else:
        self.entries[key.decode("utf-8")] = BundleEntry.parse(memoryview(value))
    if not header_seen:
      raise ValueError("Data loss: checkpoint index without a bundle header")
    self._maps = {}

  def has_tensor(self, name):
    return name in self.entries

  def get_variable_to_shape_map(self):
    return {k: list(e.shape) for k, e in self.entries.items()}

  def get_variable_to_dtype_map(self):
    out = {}
    for k, e in self.entries.items():
      out[k] = "bfloat16" if e.dtype == DT_BFLOAT16 else \
          (_DTYPES[e.dtype].name if e.dtype in _DTYPES else _DTYPE_NAMES.get(e.dtype, "dtype_%d" % e.dtype))
    return out

  def _shard(self, shard_id):
    if shard_id not in self._maps:
      path = _data_path(self.prefix, shard_id, self.num_shards)
      if not os.path.exists(path):
        raise ValueError("Error in loading checkpoint: data file %s not found" % path)
      self._maps[shard_id] = np.memmap(path, dtype=np.uint8, mode="r")
    return self._maps[shard_id]

  def get_tensor(self, name):
    if name not in self.entries:
      raise KeyError("Key %s not found in checkpoint" % name)
    e = self.entries[name]
    if e.sliced:
      raise NotImplementedError("%s is a partitioned variable (tensor slices)" % name)
    if e.dtype != DT_BFLOAT16 and e.dtype not in _DTYPES:
      raise NotImplementedError("%s has dtype %s" % (name, _DTYPE_NAMES.get(e.dtype, e.dtype)))
    dt = np.dtype('<u2') if e.dtype == DT_BFLOAT16 else _DTYPES[e.dtype]
    count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
    if count * dt.itemsize != e.size:
      raise ValueError("Data loss: %s: %d bytes stored for shape %s %s" % (name, e.size, e.shape, dt))
    raw = self._shard(e.shard_id)
    if e.offset + e.size > raw.shape[0]:
      raise ValueError("Data loss: %s extends past the end of its data file" % name)
    chunk = np.array(raw[e.offset:e.offset + e.size])            # own copy
    if self.verify and unmask_crc(e.crc32c) != crc32c(chunk):
      raise ValueError("Data loss: checksum mismatch for tensor %s" % name)
    arr = chunk.view(dt).reshape(e.shape)
    if e.dtype == DT_BFLOAT16:
      arr = (arr.astype(np.uint32) << 16).view(np.float32)
    return arr

  # dict-like access for utils/checkpoint.load
  def __contains__(self, name):
    return name in self.entries

  def __getitem__(self, name):
    return self.get_tensor(name)

  def keys(self):
    return self.entries.keys()


def write_bundle(prefix, arrays):
  """Writes {name: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001 (one shard), the
  layout of tf.train.Saver(...).save (BundleWriter: tensors in name order, no alignment)."""
  d = os.path.dirname(prefix)
  if d:
    os.makedirs(d, exist_ok=True)
  items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]     # num_shards = 1, little-endian, version.producer = 1
  offset = 0
  tmp = _data_path(prefix, 0, 1) + ".tmp"
  with open(tmp, "wb") as f:
    for name in sorted(arrays, key=lambda s: s.encode("utf-8")):
      a = np.asarray(arrays[name])
      if a.dtype.newbyteorder('<') not in _ENUM_OF and a.dtype not in _ENUM_OF:
        raise TypeError("dtype %s of %s has no checkpoint encoding" % (a.dtype, name))
      shape = tuple(a.shape)             # ascontiguousarray turns a scalar into shape (1,)
      a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False))
      e = BundleEntry()
      e.dtype = _ENUM_OF[a.dtype]
      e.shape = shape
      e.offset, e.size = offset, a.nbytes
      e.crc32c = mask_crc(crc32c(a))
      f.write(a.tobytes())
      offset += a.nbytes
      items.append((name.encode("utf-8"), e.serialize()))
  os.replace(tmp, _data_path(prefix, 0, 1))
  write_table(prefix + ".index", items)
