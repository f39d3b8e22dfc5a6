"""conv1d wgrad at the Jasper 10x5 block shapes (B=32, T=840 dense, or ragged with --ragged):
ms and TF/s per shape for forced batch-split factors. Usage: bench_wgrad_shapes.py [--ragged] [nsplit...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi
dev = torch.device("cuda:0")
args = sys.argv[1:]
ragged = "--ragged" in args
splits = [a for a in args if a != "--ragged"] or ["auto"]
shapes = [(32, 840, 256, 256, 11), (32, 840, 384, 384, 13), (32, 840, 512, 512, 17),
          (32, 840, 640, 640, 21), (32, 840, 768, 768, 25), (32, 840, 768, 896, 29), (32, 840, 896, 1024, 1)]
rng = np.random.RandomState(0)
for B, T, cin, cout, K in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  dy = torch.randn(B, T, cout, device=dev).to(torch.bfloat16)
  lens = torch.full((B,), T, dtype=torch.int32)
  if ragged:
    lens = torch.from_numpy(rng.randint(100, T + 1, size=B).astype(np.int32)); lens[0] = T
  frac = float(lens.sum()) / (B * T)
  dw = torch.zeros(K, cout, cin, device=dev)
  dil = 2 if K == 29 else 1
  out = []
  for sp in splits:
    if sp == "auto": os.environ.pop("OS2S_WGRAD_NSPLIT", None)
    else: os.environ["OS2S_WGRAD_NSPLIT"] = sp
    f = lambda: capi.conv1d_wgrad(x, dy, K, dil=dil, in_len=lens.to(dev), out=dw, accumulate=True)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    out.append("split %s %.3f ms %5.0f TF/s" % (sp, ms, 2.0 * B * T * frac * cin * cout * K / ms / 1e9))
  print("C %4d->%4d K %2d (live %.2f): " % (cin, cout, K, frac) + "  ".join(out), flush=True)
