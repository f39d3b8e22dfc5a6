"""The Dense GEMMs of a Transformer-big train step at the bench's token count (8300 packed tokens per
stack): forward / data gradient (os2s_gemm_nt vs hipBLASLt) and weight gradient (lockstep K = 1
kernel with fp32 atomics vs conv1d_wgrad1x1_pp_kernel vs hipBLASLt), plus the K = 1 weight
gradients of Jasper's residual branches (B = 32 ragged). Prints one line per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lt"))
import lt_backend  # noqa: E402  (hipBLASLt comparison harness, tools only)
from openseq2seq_amd import capi, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
M = int(os.environ.get("M", 8300))


def timeit(fn, n=8):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
      fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n)
  return best


def nan_on_error(fn):
  try:
    return timeit(fn)
  except Exception as e:  # noqa
    return float("nan")


print("== forward / data gradient: C[M,N] = A[M,K] W[N,K]^T, M = %d" % M, flush=True)
for N, K in [(1024, 1024), (3072, 1024), (1024, 3072), (2048, 1024), (1024, 2048), (4096, 1024),
             (1024, 4096), (32768, 1024), (1024, 32768)]:
  a = torch.randn(M, K, device=dev).to(torch.bfloat16)
  w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
  y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  t1 = timeit(lambda: capi.gemm_nt(a, w, out=y))
  t2 = nan_on_error(lambda: lt_backend.matmul_lt(a, w, b_is_t=True, out=y))
  fl = 2.0 * M * N * K
  print("N %5d K %5d: gemm_nt %.3f ms %5.0f TF/s | hipBLASLt %.3f ms %5.0f TF/s" % (
      N, K, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9), flush=True)

print("== weight gradient: dW[Cout,Cin] += dY[M,Cout]^T X[M,Cin], M = %d" % M, flush=True)
for Cin, Cout in [(1024, 1024), (1024, 3072), (1024, 2048), (1024, 4096), (4096, 1024), (1024, 32768)]:
  x = torch.randn(1, M, Cin, device=dev).to(torch.bfloat16)
  dy = torch.randn(1, M, Cout, device=dev).to(torch.bfloat16)
  dw = torch.zeros(1, Cout, Cin, device=dev)
  res = {}
  for name, var in (("lockstep", 0), ("pp1x1", 2)):
    _lib.set_option("conv1d_wgrad.variant", var); _lib.set_option("conv1d_wgrad.split", -1)
    try:
      res[name] = timeit(lambda: capi.conv1d_wgrad(x, dy, 1, pad_left=0, out=dw, accumulate=True))
    finally:
      _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  res["lt"] = nan_on_error(lambda: lt_backend.matmul_lt(dy[0], x[0], a_is_t=True, out=dw[0], beta=1.0))
  fl = 2.0 * M * Cin * Cout
  print("Cin %5d Cout %5d: " % (Cin, Cout) + " | ".join(
      "%s %.3f ms %5.0f TF/s" % (k, v, fl / v / 1e9) for k, v in res.items()), flush=True)

print("== Jasper residual 1x1 weight gradients: B = 32, T = 840, ragged", flush=True)
g = torch.Generator().manual_seed(3)
B, T = 32, 840
lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32).to(dev)
for Cin, Cout in [(256, 256), (256, 384), (384, 512), (512, 640), (640, 768), (768, 768), (1024, 1024)]:
  x = torch.randn(B, T, Cin, device=dev).to(torch.bfloat16)
  dy = torch.randn(B, T, Cout, device=dev).to(torch.bfloat16)
  dw = torch.zeros(1, Cout, Cin, device=dev)
  res = {}
  for name, var in (("lockstep", 0), ("pp1x1", 2)):
    _lib.set_option("conv1d_wgrad.variant", var); _lib.set_option("conv1d_wgrad.split", -1)
    try:
      res[name] = timeit(lambda: capi.conv1d_wgrad(x, dy, 1, pad_left=0, in_len=lens, out=dw, accumulate=True))
    finally:
      _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  fl = 2.0 * float(lens.sum()) * Cin * Cout
  print("Cin %5d Cout %5d: " % (Cin, Cout) + " | ".join(
      "%s %.3f ms %5.0f TF/s" % (k, v, fl / v / 1e9) for k, v in res.items()), flush=True)
