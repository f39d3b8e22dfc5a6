"""Which in-tree kernel should a bare matmul C[M,N] = A[M,K] W[N,K]^T take? os2s_gemm_nt (256 x 256
ping-pong tile) vs the K = 1 case of os2s_conv1d_fwd (lockstep tiles, 2-3 workgroups per CU) vs
hipBLASLt, at the shapes of the NMT / DS2 / Tacotron2 configs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lt"))
import lt_backend  # noqa: E402  (hipBLASLt comparison harness, tools only)
from openseq2seq_amd import capi

dev = torch.device("cuda:0")


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


for M, N, K in [(6400, 2048, 512), (6400, 2048, 1024), (6400, 2048, 1536), (6400, 512, 1024), (6400, 512, 2048),
                (6400, 1024, 2048), (6400, 32768, 512), (6400, 32768, 1024), (6400, 512, 32768),
                (12800, 2400, 1600), (12800, 1600, 2400), (12800, 2400, 1312), (3200, 4096, 1536),
                (1600, 1024, 512), (800, 2048, 512)]:
  a = torch.randn(M, K, device=dev).to(torch.bfloat16)
  w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
  y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  res = {}
  if K % 64 == 0:
    res["gemm_nt"] = timeit(lambda: capi.gemm_nt(a, w, out=y))
  res["conv_k1"] = timeit(lambda: capi.conv1d_fwd(a.view(1, M, K), w.view(1, N, K), pad_left=0, tout=M, out=y.view(1, M, N)))
  try:
    res["lt"] = timeit(lambda: lt_backend.matmul_lt(a, w, b_is_t=True, out=y))
  except Exception:
    pass
  fl = 2.0 * M * N * K
  print("M %5d N %5d K %5d: " % (M, N, K) + " | ".join(
      "%s %.3f ms %5.0f TF/s" % (k, v, fl / v / 1e9) for k, v in res.items()), flush=True)
