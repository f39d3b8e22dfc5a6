"""hipBLASLt comparison back end for A/B runs (tools only — the product binding knows nothing about it).

`matmul_lt` calls tools/lt/libos2s_lt.so (build with tools/lt/build.sh); `install()` monkey-patches
`openseq2seq_amd.capi.gemm`, `gemm_nt` and `gemm_wgrad` so that the bare matmuls of a model run on the
vendor library (matmul + one elementwise epilogue pass where the in-tree kernel fuses the epilogue).

  python -c "import sys; sys.path.insert(0, 'tools/lt'); import lt_backend; lt_backend.install(); \
             import runpy; sys.argv = ['bench.py', '--only-transformer']; runpy.run_path('bench.py', run_name='__main__')"
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong as c_ll, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lt_lib = None
MIN_ROWS = 256


def _lt():
  global _lt_lib
  if _lt_lib is None:
    path = os.path.join(_HERE, "libos2s_lt.so")
    if not os.path.exists(path):
      raise RuntimeError("comparison library missing: run tools/lt/build.sh")
    _lt_lib = ctypes.CDLL(path)
    _lt_lib.os2s_lt_matmul.restype = c_int
    _lt_lib.os2s_lt_matmul.argtypes = [c_void_p, c_void_p, c_int, c_ll, c_void_p, c_int, c_ll, c_void_p, c_int,
                                       c_ll, c_int, c_int, c_int, c_float]
  return _lt_lib


def matmul_lt(a, b, a_is_t=False, b_is_t=False, out=None, out_f32=False, beta=0.0):
  """out[M,N] = op(a) @ op(b) (+ beta * out) via hipBLASLt; a, b bf16 2-D (row stride free)."""
  M, K = (a.shape[1], a.shape[0]) if a_is_t else (a.shape[0], a.shape[1])
  K2, N = (b.shape[1], b.shape[0]) if b_is_t else (b.shape[0], b.shape[1])
  assert K == K2 and a.stride(1) == 1 and b.stride(1) == 1
  if out is None:
    assert beta == 0.0
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
  assert out.stride(1) == 1 and tuple(out.shape) == (M, N)
  st = c_void_p(torch.cuda.current_stream().cuda_stream)
  rc = _lt().os2s_lt_matmul(st, c_void_p(a.data_ptr()), int(a_is_t), a.stride(0), c_void_p(b.data_ptr()),
                            int(b_is_t), b.stride(0), c_void_p(out.data_ptr()), int(out.dtype == torch.float32),
                            out.stride(0), M, N, K, float(beta))
  if rc != 0:
    raise RuntimeError("os2s_lt_matmul failed: %d" % rc)
  return out


def install():
  """Route the bare matmuls of openseq2seq_amd through hipBLASLt (A/B runs)."""
  from openseq2seq_amd import capi
  gemm0, gemm_nt0, wgrad0 = capi.gemm, capi.gemm_nt, capi.gemm_wgrad

  def gemm(x2d, w, **kw):
    plain = (all(kw.get(k) is None for k in ("residual", "stats", "in_len", "out_len", "bias"))
             and not kw.get("act", 0) and not kw.get("out_f32", False) and not kw.get("time_major", False)
             and kw.get("keep_prob", 1.0) >= 1.0)
    if plain and x2d.shape[0] >= MIN_ROWS and x2d.stride(1) == 1 and w.stride(1) == 1:
      try:
        return matmul_lt(x2d, w, b_is_t=True, out=kw.get("out"), beta=1.0 if kw.get("accumulate", False) else 0.0)
      except RuntimeError:
        pass
    return gemm0(x2d, w, **kw)

  def gemm_nt(a, w, out=None, bias=None, act=0, keep_prob=1.0, seed=0, residual=None, accumulate=False,
              out_f32=False):
    if accumulate or out_f32 or a.shape[0] < MIN_ROWS:
      return gemm_nt0(a, w, out=out, bias=bias, act=act, keep_prob=keep_prob, seed=seed, residual=residual,
                      accumulate=accumulate, out_f32=out_f32)
    y = matmul_lt(a, w, b_is_t=True, out=out)
    if not (act == 0 and keep_prob >= 1.0 and residual is None and bias is None):
      capi.dense_epilogue(y, bias=bias, act=act, keep_prob=keep_prob, seed=seed, residual=residual)
    return y

  def gemm_wgrad(x2d, dy2d, out, accumulate=True):
    if x2d.shape[0] >= MIN_ROWS and x2d.stride(1) == 1 and dy2d.stride(1) == 1 and out.stride(1) == 1:
      try:
        matmul_lt(dy2d, x2d, a_is_t=True, out=out, beta=1.0 if accumulate else 0.0)
        return
      except RuntimeError:
        pass
    wgrad0(x2d, dy2d, out, accumulate=accumulate)

  capi.gemm, capi.gemm_nt, capi.gemm_wgrad = gemm, gemm_nt, gemm_wgrad
