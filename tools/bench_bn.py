"""Times the BatchNorm/activation kernels at Jasper 10x5 shapes (B=32, ragged lengths U[2,16.7] s):
effective GB/s = algorithmic bytes (live rows read, all rows written) / time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi

dev = torch.device("cuda:0")
B = 32
rng = np.random.RandomState(0)
dur = rng.uniform(2.0, 16.7, B)
lens = (dur * 100 / 2).astype(np.int32) + 1
T = int(-(-lens.max() // 16) * 16)
out_len = torch.from_numpy(lens).to(dev)
live = float(lens.sum()) / (B * T)
print("B %d T %d live %.3f" % (B, T, live))


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


for C in (256, 384, 512, 640, 768, 1024):
  for J in (1, 4):
    ys = [torch.randn(B, T, C, device=dev).bfloat16() for _ in range(J)]
    sc = [torch.rand(C, device=dev) + 0.5 for _ in range(J)]
    sh = [torch.randn(C, device=dev) for _ in range(J)]
    mean = [torch.randn(C, device=dev) * 0.1 for _ in range(J)]
    rstd = [torch.rand(C, device=dev) + 0.5 for _ in range(J)]
    out = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(B, T, C, device=dev).bfloat16()
    dz = torch.empty_like(out)
    dy = torch.empty_like(out)
    nparts = capi.bn_act_bwd_num_parts(B * T)
    partial = torch.empty(nparts, 1 + J, C, device=dev)
    c1 = torch.randn(C, device=dev) * 0.01
    c2 = torch.randn(C, device=dev) * 0.01
    plane = B * T * C * 2 / 1e9
    t_f = timeit(lambda: capi.bn_act_fwd(ys, sc, sh, out, out_len, 1, 0.8, 7))
    t_r = timeit(lambda: capi.bn_act_bwd_reduce(dout, out, ys, mean, rstd, dz, partial, out_len, 1, 0.8, 7))
    t_a = timeit(lambda: capi.bn_bwd_apply(dz, ys[0], sc[0], mean[0], rstd[0], c1, c2, dy, out_len=out_len, margin=24))
    gb_f = plane * (J * live + 1)
    gb_r = plane * ((2 + J) * live + 1)
    gb_a = plane * (2 * live + 1)
    print("C %4d J %d  fwd %6.1f us %5.2f TB/s | reduce %6.1f us %5.2f TB/s | apply %6.1f us %5.2f TB/s (ragged-ideal bytes)"
          % (C, J, t_f, gb_f / t_f * 1e3, t_r, gb_r / t_r * 1e3, t_a, gb_a / t_a * 1e3))
