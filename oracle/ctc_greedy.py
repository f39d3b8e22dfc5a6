"""ORACLE (test infrastructure only): CTC greedy decode on the CPU.

Three independent CPU implementations used to cross-pin each other:
  * `greedy_numpy`  — NumPy restatement of tf.nn.ctc_greedy_decoder as called in
    open_seq2seq/decoders/fc_decoders.py:244-251 (argmax -> merge repeats ->
    drop blank V-1), cf. open_seq2seq/utils/ctc_decoder.py:5-40;
  * `greedy_c`      — the plain-C restatement oracle/ctc_greedy.c (gcc);
  * `greedy_reference_cpp` — the REFERENCE's own decoders/ctc_greedy_decoder.cpp
    compiled where it lies into oracle/_ref/ (probabilities in, ids out).
"""
import ctypes
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


def greedy_numpy(logits, seq_len, blank=None, merge_repeated=True):
  """logits [T,B,V] float32, seq_len [B] -> (ids [B,T] int32 (-1 pad), lens [B], neg_sum [B])."""
  logits = np.asarray(logits, dtype=np.float32)
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  ids = np.full((B, T), -1, dtype=np.int32)
  lens = np.zeros((B,), dtype=np.int32)
  neg = np.zeros((B,), dtype=np.float32)
  for b in range(B):
    n = int(min(max(int(seq_len[b]), 0), T))
    if n == 0:
      continue
    am = np.argmax(logits[:n, b, :], axis=-1)  # first maximum wins
    score = np.float32(0)
    for t in range(n):
      score = np.float32(score + logits[t, b, am[t]])
    neg[b] = -score
    keep = am != blank
    if merge_repeated:
      keep[1:] &= am[1:] != am[:-1]
    out = am[keep]
    ids[b, :len(out)] = out
    lens[b] = len(out)
  return ids, lens, neg


def _load(path):
  if not os.path.exists(path):
    raise FileNotFoundError(
        "%s missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
  return ctypes.CDLL(path)


def greedy_c(logits, seq_len, blank=None, merge_repeated=True):
  lib = _load(os.path.join(_DIR, "liboracle.so"))
  logits = np.ascontiguousarray(logits, dtype=np.float32)
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  sl = np.ascontiguousarray(seq_len, dtype=np.int32)
  ids = np.empty((B, T), dtype=np.int32)
  lens = np.empty((B,), dtype=np.int32)
  neg = np.empty((B,), dtype=np.float32)
  fp = ctypes.POINTER(ctypes.c_float)
  ip = ctypes.POINTER(ctypes.c_int32)
  lib.oracle_ctc_greedy_decode.restype = ctypes.c_int
  rc = lib.oracle_ctc_greedy_decode(
      logits.ctypes.data_as(fp), sl.ctypes.data_as(ip), T, B, V, int(blank),
      int(bool(merge_repeated)), ids.ctypes.data_as(ip), lens.ctypes.data_as(ip),
      neg.ctypes.data_as(fp))
  assert rc == 0
  return ids, lens, neg


def reference_cpp_available():
  return os.path.exists(os.path.join(_DIR, "_ref", "libref_ctc_greedy.so"))


def greedy_reference_cpp(probs_tv):
  """probs_tv [T,V] (softmax probabilities, blank last) -> ids list, through the
  reference's compiled C++ (always merges repeats, blank = V-1)."""
  lib = _load(os.path.join(_DIR, "_ref", "libref_ctc_greedy.so"))
  p = np.ascontiguousarray(probs_tv, dtype=np.float64)
  T, V = p.shape
  out = np.empty((max(T, 1),), dtype=np.int32)
  lib.ref_ctc_greedy_decode.restype = ctypes.c_int
  n = lib.ref_ctc_greedy_decode(
      p.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), T, V,
      out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
  return out[:n].copy()
