"""Ping-pong conv kernel: time vs forced tail-split factor at the wide Jasper shapes, dense and
ragged (fits the cost model of the device-side split decision)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
B, T = 32, 840
shapes = [(512, 512, 17), (640, 640, 21), (768, 768, 25), (768, 896, 29)]
rng = np.random.RandomState(1234)
def lens_for(lo, hi):
  dur = rng.uniform(lo, hi, size=B)
  return np.minimum((1 + (dur * 16000).astype(np.int64) // 160 + 1) // 2, T).astype(np.int32)
batches = {"dense": None, "rag2-16.7": lens_for(2.0, 16.7), "rag8-16.7": lens_for(8.0, 16.7), "rag12-16.7": lens_for(12.0, 16.7)}
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
L = _lib.lib()
for cin, cout, K in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  stats = torch.empty(capi.conv1d_num_mtiles(B, T), 2, cout, device=dev)
  dil = 2 if K == 29 else 1
  for name, ln in batches.items():
    lens = None if ln is None else torch.from_numpy(ln).to(dev)
    pl = (K - 1) * dil // 2
    nl = B * 7 if ln is None else int(sum(min(7, (int(v) + pl + 127) // 128) for v in ln))
    U = ((nl + 1) // 2) * ((cout + 255) // 256)
    out = []
    _lib.set_option("conv1d.variant", 3)
    out.append("tile128 %.3f" % timeit(lambda: capi.conv1d_fwd(x, w, out=y, dil=dil, stats=stats, in_len=lens)))
    _lib.set_option("conv1d.variant", 10)
    for f in (-1, 1, 2, 3, 4, 6, 8):
      _lib.set_option("conv1d.split", f)
      out.append("f%d %.3f" % (f, timeit(lambda: capi.conv1d_fwd(x, w, out=y, dil=dil, stats=stats, in_len=lens))))
    _lib.set_option("conv1d.split", -1); _lib.set_option("conv1d.variant", -1)
    print("C %4d->%4d K %2d %-10s live windows %3d units %3d (r %3d): %s" % (cin, cout, K, name, nl, U, U % 256, "  ".join(out)), flush=True)
