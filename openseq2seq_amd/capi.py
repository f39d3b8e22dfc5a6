"""Thin Python wrappers: one function per C-ABI entry point of include/os2s.h.

Arguments are torch CUDA tensors (used as plain device buffers); every wrapper
validates dtype / contiguity / device, passes raw pointers + the current HIP
stream, and raises Os2sError on a non-zero status.
"""
import functools

import torch

from . import _lib
from ._lib import c_void_p, c_int, c_int64, c_size_t, c_float, c_uint64


def _stream():
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, allow_none=False):
  if t is None:
    if allow_none:
      return c_void_p(0)
    raise ValueError("tensor required")
  if not t.is_cuda:
    raise _lib.Os2sError("os2s kernels need CUDA(HIP) tensors; got %s" % t.device)
  if dtype is not None and t.dtype != dtype:
    raise TypeError("expected %s, got %s" % (dtype, t.dtype))
  if not t.is_contiguous():
    raise ValueError("tensor must be contiguous")
  return c_void_p(t.data_ptr())


@functools.lru_cache(maxsize=None)
def _fn(name, argtypes, restype=c_int):
  return _lib.bind(name, list(argtypes), restype)


def abi_version():
  return _lib.lib().os2s_abi_version()


# --------------------------------------------------------------------------
# CTC greedy decode
# --------------------------------------------------------------------------
def ctc_greedy_decode(logits, seq_len, blank=None, merge_repeated=True):
  """logits [T,B,V] fp32 (time-major), seq_len [B] int32.

  Returns (ids [B,T] int32 padded with -1, lens [B] int32, neg_sum_logits [B] fp32).
  Mirrors tf.nn.ctc_greedy_decoder as used in fc_decoders.py:244-251.
  """
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  dev = logits.device
  ids = torch.empty((B, T), dtype=torch.int32, device=dev)
  lens = torch.empty((B,), dtype=torch.int32, device=dev)
  neg = torch.empty((B,), dtype=torch.float32, device=dev)
  wsf = _fn("os2s_ctc_greedy_decode_workspace_bytes", (c_int, c_int), c_size_t)
  nbytes = int(wsf(T, B))
  ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
  f = _fn("os2s_ctc_greedy_decode",
          (c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t))
  _lib.check(f(_stream(), _ptr(logits, torch.float32), _ptr(seq_len, torch.int32),
               T, B, V, int(blank), int(bool(merge_repeated)), _ptr(ids),
               _ptr(lens), _ptr(neg), _ptr(ws), nbytes),
             "os2s_ctc_greedy_decode")
  return ids, lens, neg
