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


# --------------------------------------------------------------------------
# conv1d (implicit GEMM)
# --------------------------------------------------------------------------
c_ll = _lib.ctypes.c_longlong


def same_padding(tin, k, stride, dil):
  """TF 'SAME' padding (asymmetric when stride > 1): returns (tout, pad_left)."""
  tout = (tin + stride - 1) // stride
  total = max((tout - 1) * stride + (k - 1) * dil + 1 - tin, 0)
  return tout, total // 2


def valid_padding(tin, k, stride, dil):
  return (tin - (k - 1) * dil - 1) // stride + 1, 0


def conv1d_num_mtiles(B, tout):
  return int(_fn("os2s_conv1d_num_mtiles", (c_int, c_int))(B, tout))


def conv1d_fwd(x, w, *, stride=1, dil=1, pad_left=None, tout=None, in_len=None,
               bias=None, stats=None, out=None, out_f32=False, accumulate=False,
               time_major=False):
  """x [B,Tin,Cin] bf16, w [K,Cout,Cin] bf16 -> y [B,Tout,Cout] (or [Tout,B,Cout]
  when time_major). pad_left/tout default to TF 'SAME'."""
  B, Tin, Cin = x.shape
  K, Cout, Cin2 = w.shape
  assert Cin == Cin2
  if pad_left is None or tout is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
  dt = torch.float32 if out_f32 else torch.bfloat16
  if out is None:
    shape = (tout, B, Cout) if time_major else (B, tout, Cout)
    out = torch.empty(shape, dtype=dt, device=x.device)
  if time_major:
    ysb, yst = Cout, B * Cout
  else:
    ysb, yst = tout * Cout, Cout
  f = _fn("os2s_conv1d_fwd",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
           c_ll, c_ll, c_int, c_int))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(w, torch.bfloat16),
               _ptr(out, dt), _ptr(in_len, torch.int32, True),
               _ptr(bias, torch.float32, True), _ptr(stats, torch.float32, True),
               B, Tin, Cin, Cout, K, stride, dil, pad_left, tout, ysb, yst,
               int(out_f32), int(accumulate)), "os2s_conv1d_fwd")
  return out


def conv1d_wgrad(x, dy, K, *, stride=1, dil=1, pad_left=None, in_len=None,
                 out=None, accumulate=False):
  """x [B,Tin,Cin] bf16, dy [B,Tout,Cout] bf16 -> dW [K,Cout,Cin] fp32."""
  B, Tin, Cin = x.shape
  B2, Tout, Cout = dy.shape
  assert B == B2
  if pad_left is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
    assert tout == Tout
  if out is None:
    assert not accumulate
    out = torch.empty((K, Cout, Cin), dtype=torch.float32, device=x.device)
  f = _fn("os2s_conv1d_wgrad",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
           c_int, c_int, c_int, c_int, c_int, c_int, c_int))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(dy, torch.bfloat16),
               _ptr(out, torch.float32), _ptr(in_len, torch.int32, True), B, Tin,
               Cin, Cout, K, stride, dil, pad_left, Tout, int(accumulate)),
             "os2s_conv1d_wgrad")
  return out
