"""Sweep of the BatchNorm kernels' tiling (os2s_set_option bn.<kernel>.groups / .rows) at Jasper shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from openseq2seq_amd import capi, _lib

dev = torch.device("cuda:0")
B = 32
rng = np.random.RandomState(0)
lens = (rng.uniform(2.0, 16.7, B) * 50).astype(np.int32) + 1
T = int(-(-lens.max() // 16) * 16)
out_len = torch.from_numpy(lens).to(dev)
_BN_KERNELS = ("act_fwd", "act_bwd_reduce", "bwd_apply")


def set_tiling(which, groups, rows):
  _lib.set_option("bn.%s.groups" % _BN_KERNELS[which], groups)
  _lib.set_option("bn.%s.rows" % _BN_KERNELS[which], rows)


def timeit(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(n):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / n * 1e3


# keep several distinct buffers so the 256 MB MALL does not hold the working set
NBUF = 6
for C in (256, 512, 768, 1024):
  J = 1
  bufs = []
  for i in range(NBUF):
    bufs.append(dict(y=torch.randn(B, T, C, device=dev).bfloat16(), out=torch.empty(B, T, C, device=dev, dtype=torch.bfloat16),
                     dout=torch.randn(B, T, C, device=dev).bfloat16(), dz=torch.empty(B, T, C, device=dev, dtype=torch.bfloat16),
                     dy=torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)))
  sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
  mean, rstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
  c1, c2 = torch.randn(C, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01
  partial = torch.empty(B * T // 8 + 8, 1 + J, C, device=dev)
  it = [0]

  def nxt():
    it[0] = (it[0] + 1) % NBUF
    return bufs[it[0]]
  print("C", C)
  for G in (256, 128, 64, 48, 32, 16):
    if G < 256 and (C // 8) % G and G != 32:
      continue
    row = []
    for R in (16, 32, 64, 128):
      for w in range(3):
        set_tiling(w, G, R)
      def f_fwd():
        b = nxt(); capi.bn_act_fwd([b["y"]], [sc], [sh], b["out"], out_len, 1, 0.8, 7)
      def f_red():
        b = nxt(); capi.bn_act_bwd_reduce(b["dout"], b["out"], [b["y"]], [mean], [rstd], b["dz"], partial, out_len, 1, 0.8, 7)
      def f_app():
        b = nxt(); capi.bn_bwd_apply(b["dz"], b["y"], sc, mean, rstd, c1, c2, b["dy"], out_len=out_len, margin=24)
      row.append("R%-3d f %5.1f r %5.1f a %5.1f" % (R, timeit(f_fwd), timeit(f_red), timeit(f_app)))
    print("  G %3d | " % G + " | ".join(row))
