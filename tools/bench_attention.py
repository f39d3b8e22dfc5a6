"""Attention kernels on their own at the Transformer-big training shape (B=256 sequences, lengths
U[8,56], 16 heads of 64, QKV packed [tokens, 3*1024], attention dropout 0.1): microseconds per launch
for self-attention (encoder), causal self-attention (decoder) and cross attention, forward and
backward, against the HBM time of the bytes each launch has to move.

  python tools/bench_attention.py [--batch 256] [--iters 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=256)
  ap.add_argument("--iters", type=int, default=50)
  ap.add_argument("--keep", type=float, default=0.9)
  args = ap.parse_args()
  from openseq2seq_amd import capi
  dev = torch.device("cuda:0")
  rng = np.random.RandomState(0)
  H, D = 16, 1024
  lq = rng.randint(8, 57, size=args.batch)
  lk = rng.randint(8, 57, size=args.batch)

  def cu(l):
    return torch.tensor([0] + list(np.cumsum(l)), dtype=torch.int32, device=dev)

  def timed(fn):
    for _ in range(5):
      fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.iters

  for name, causal, cross in (("self", False, False), ("causal", True, False), ("cross", False, True)):
    nq, nk = int(lq.sum()), int((lk if cross else lq).sum())
    cq, ck = cu(lq), cu(lk if cross else lq)
    if cross:
      qb = torch.randn(nq, D, device=dev).to(torch.bfloat16)
      kvb = torch.randn(nk, 2 * D, device=dev).to(torch.bfloat16)
      q, k, v = qb, kvb[:, :D], kvb[:, D:]
    else:
      qkv = torch.randn(nq, 3 * D, device=dev).to(torch.bfloat16)
      q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    do = torch.randn(nq, D, device=dev).to(torch.bfloat16)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    o, lse = capi.attention_fwd(q, k, v, cq, ck, H, 64, causal, 0.125, args.keep, 1)
    tf = timed(lambda: capi.attention_fwd(q, k, v, cq, ck, H, 64, causal, 0.125, args.keep, 1))
    tb = timed(lambda: capi.attention_bwd(q, k, v, do, lse, dq, dk, dv, cq, ck, H, 64, causal, 0.125,
                                          args.keep, 1))
    fb = (2 * nq + 2 * nk) * D * 2            # q, o + k, v
    bb = (3 * nq + 4 * nk) * D * 2            # q, do, dq + k, v, dk, dv
    print("%-6s tokens q=%d k=%d  fwd %.1f us (HBM time %.1f us)  bwd %.1f us (HBM time %.1f us)" %
          (name, nq, nk, tf, fb / 8e6, tb, bb / 8e6))


if __name__ == "__main__":
  main()
