"""Library GEMM (torch.matmul -> hipBLASLt/rocBLAS) vs the in-tree MFMA GEMM (conv1d K=1) on the
Transformer-big training shapes: M tokens x N x K, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lt"))
import lt_backend  # noqa: E402  (hipBLASLt comparison harness, tools only)
from openseq2seq_amd import capi
dev = torch.device("cuda:0")
def t(fn, n=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for M in (8192, 16384):
  for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (32768, 1024)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16) * 0.02
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    f = 2.0 * M * N * K / 1e9
    a = t(lambda: torch.matmul(x, w.t()))
    b = t(lambda: capi.gemm(x, w))
    dw = torch.zeros(N, K, device=dev)
    c = t(lambda: torch.matmul(dy.t(), x))          # wgrad shape (bf16 out)
    d = t(lambda: capi.gemm_wgrad(x, dy, dw, accumulate=True))
    a2 = t(lambda: lt_backend.matmul_lt(x, w, b_is_t=True))
    c2 = t(lambda: lt_backend.matmul_lt(dy, x, a_is_t=True, out=dw, beta=1.0))
    print("   os2s_matmul_lt: fwd %.3f ms %4.0f TF/s | wgrad(fp32,+=) %.3f ms %4.0f TF/s" % (a2, f / a2, c2, f / c2))
    print("M %5d N %5d K %4d: lib fwd %.3f ms %4.0f TF/s | ours %.3f ms %4.0f TF/s || lib wgrad %.3f ms %4.0f | ours %.3f ms %4.0f"
          % (M, N, K, a, f / a, b, f / b, c, f / c, d, f / d), flush=True)
