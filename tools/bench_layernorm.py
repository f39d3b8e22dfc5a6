"""LayerNorm forward / backward on their own at the Transformer-big shape ([tokens, 1024] bf16,
tokens = 8.3k per side): microseconds per launch against the HBM time of the bytes moved.
Round 3: fwd 13 us, bwd + its gamma/beta finalize 25.6 us back to back; a backward that requests all
three inputs of a row together, prefetches the next row and reduces with DPP measured the same
(25.2-25.5 us at 16 / 32 rows per workgroup, 29.9 at 8) and was not kept — the 45 us the kernel
averages inside a training step is CU sharing with the weight-gradient stream, not its own latency.

  python tools/bench_layernorm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  from openseq2seq_amd import capi
  dev = torch.device("cuda:0")
  N, D, iters = 8310, 1024, 50
  x = torch.randn(N, D, device=dev).to(torch.bfloat16)
  dy = torch.randn(N, D, device=dev).to(torch.bfloat16)
  dres = torch.randn(N, D, device=dev).to(torch.bfloat16)
  gam, bet = torch.rand(D, device=dev) + 0.5, torch.zeros(D, device=dev)
  dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
  y, mean, rstd = capi.layernorm_fwd(x, gam, bet)

  def timed(fn):
    for _ in range(5):
      fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

  tf = timed(lambda: capi.layernorm_fwd(x, gam, bet))
  tb = timed(lambda: capi.layernorm_bwd(dy, x, gam, mean, rstd, dres, dgam, dbet))
  print("fwd %.1f us (HBM time %.1f us)  bwd + finalize %.1f us (HBM time %.1f us)" %
        (tf, 2 * N * D * 2 / 8e6, tb, 4 * N * D * 2 / 8e6))


if __name__ == "__main__":
  main()
