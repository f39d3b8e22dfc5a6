import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
from openseq2seq_amd import capi
from bench_decode_kernels import timeit
dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
N = 256
for (n, k) in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (32768, 1024)]:
  for pad in (0, 8, 64, 128):
    w = bf(n, k + pad)[:, :k]
    x = bf(N, k + pad)[:, :k]
    t = timeit(lambda: capi.gemm_skinny(x, w))
    print("N=%d K=%d pad=%d: %.1f us" % (n, k, pad, t), flush=True)
