"""Hand-written NT GEMM (os2s_gemm_nt) vs hipBLASLt (os2s_matmul_lt) at the Transformer-big shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lt"))
import lt_backend  # noqa: E402  (hipBLASLt comparison harness, tools only)
from openseq2seq_amd import capi
dev = torch.device("cuda:0")
def timeit(fn, n=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n)
  return best
for M, N, K in [(16384, 1024, 1024), (16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096),
                (8192, 32768, 1024), (8192, 1024, 32768), (27000, 768, 768)]:
  a = torch.randn(M, K, device=dev).to(torch.bfloat16)
  w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
  y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  t1 = timeit(lambda: capi.gemm_nt(a, w, out=y))
  try:
    t2 = timeit(lambda: lt_backend.matmul_lt(a, w, b_is_t=True, out=y))
  except Exception as e:
    t2 = float("nan")
  fl = 2.0 * M * N * K
  print("M %5d N %5d K %5d: gemm_nt %.3f ms %5.0f TF/s | hipBLASLt %.3f ms %5.0f TF/s" % (
      M, N, K, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9), flush=True)
