#!/usr/bin/env python
"""Micro-benchmarks of the kernels of one Transformer-big beam-search step (N = 256 beam rows):
skinny GEMMs per shape, the logits GEMM, decode attention, LayerNorm, the beam kernels."""
import os
import sys

import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lt"))
import lt_backend  # noqa: E402  (hipBLASLt comparison harness, tools only)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openseq2seq_amd import capi  # noqa: E402


def timeit(fn, reps=20, warm=3):
  """GPU time per call: `reps` calls captured in one hipGraph (no host launch cost)."""
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(reps):
      fn()
  g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
  dev = torch.device("cuda:0")
  N, D, H, F, V = 256, 1024, 16, 4096, 32768
  bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
  x, xf = bf(N, D), bf(N, F)
  for name, (n, k, inp) in {"qkv 3072x1024": (3 * D, D, x), "proj 1024x1024": (D, D, x),
                            "ffn-in 4096x1024": (F, D, x), "ffn-out 1024x4096": (D, F, xf),
                            "logits 32768x1024": (V, D, x)}.items():
    w = bf(n, k)
    os.environ["OS2S_SKINNY_VARIANT"] = "reg"
    t = timeit(lambda: capi.gemm_skinny(inp, w))
    os.environ["OS2S_SKINNY_VARIANT"] = "l64"
    t1 = timeit(lambda: capi.gemm_skinny(inp, w))
    os.environ["OS2S_SKINNY_VARIANT"] = "l32"
    t3 = timeit(lambda: capi.gemm_skinny(inp, w))
    os.environ["OS2S_SKINNY_VARIANT"] = "wide"
    t4 = timeit(lambda: capi.gemm_skinny(inp, w))
    del os.environ["OS2S_SKINNY_VARIANT"]
    t5 = timeit(lambda: capi.gemm_skinny(inp, w))
    t2 = timeit(lambda: lt_backend.matmul_lt(inp, w, b_is_t=True))
    print("gemm %-18s reg %6.1f  lds64 %6.1f  lds32 %6.1f  wide %6.1f  auto %6.1f  hipBLASLt %6.1f us  (W %.1f MB)"
          % (name, t, t1, t3, t4, t5, t2, n * k * 2 / 1e6), flush=True)
  Tmax, step = 106, 50
  kc, vc = bf(N, Tmax, D), bf(N, Tmax, D)
  anc = torch.randint(0, N, (N, Tmax), device=dev, dtype=torch.int32)
  qkv = bf(N, 3 * D)
  t = timeit(lambda: capi.decode_self_attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], kc, vc, anc, H, step, 0.125))
  print("decode self-attention step %d: %.1f us (ideal %.1f us @8TB/s)" % (step, t, N * (step + 1) * D * 4 / 8e6))
  B, beam, S = 64, 4, 40
  cu = torch.arange(B + 1, device=dev, dtype=torch.int32) * S
  kv = bf(B * S, 2 * D)
  t = timeit(lambda: capi.decode_cross_attention(x, kv[:, :D], kv[:, D:], cu, beam, H, S, 0.125))
  print("decode cross-attention S=%d: %.1f us" % (S, t))
  g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
  t = timeit(lambda: capi.layernorm_fwd(x, g, b, save=False))
  print("layernorm [256,1024]: %.1f us" % t)
  st = capi.BeamState(torch.zeros(B, dtype=torch.int32, device=dev), beam, V, 200, 0.6, 1)
  logits = bf(N, V)

  st2 = capi.BeamState(torch.zeros(B, dtype=torch.int32, device=dev), beam, V, 100000, 0.6, 1)
  t = timeit(lambda: st2.step(logits), reps=20)
  print("beam step (row top-k + select): %.1f us" % t)


if __name__ == "__main__":
  main()
