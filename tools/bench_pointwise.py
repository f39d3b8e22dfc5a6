"""1x1 (pointwise) conv forward at QuartzNet 15x5 shapes: us per launch by tile variant, dense and ragged."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
B, T = 32, 836
shapes = [(256, 256), (256, 512), (512, 512), (512, 1024), (1024, 1024)]
rng = np.random.RandomState(1234)
dur = rng.uniform(2.0, 16.7, size=B)
lens_np = np.minimum((1 + (dur * 16000).astype(np.int64) // 160 + 1) // 2, T).astype(np.int32)
live = float(lens_np.sum()) / (B * T)
print("live fraction %.3f" % live)
def timeit(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n)
  return best * 1e3
cfgs = [("auto", -1, 0), ("v0", 0, 0), ("v5", 5, 0), ("v10", 10, 0), ("v10+1x1pp", 10, 2), ("v12", 12, 0), ("v14", 14, 0)]
for cin, cout in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(1, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  nm = capi.conv1d_num_mtiles(B, T)
  stats = torch.empty(nm, 2, cout, device=dev)
  rag = torch.from_numpy(lens_np).to(dev)
  byts = B * T * (cin + cout) * 2
  fl = 2.0 * B * T * cin * cout
  out = []
  for name, v, v1 in cfgs:
    _lib.set_option("conv1d.variant", v); _lib.set_option("conv1x1.variant", v1)
    try:
      us_d = timeit(lambda: capi.conv1d_fwd(x, w, out=y, stats=stats))
      us_r = timeit(lambda: capi.conv1d_fwd(x, w, out=y, stats=stats, in_len=rag))
      out.append("%s %.1f/%.1f" % (name, us_d, us_r))
    except Exception as e:
      out.append("%s ERR" % name)
  _lib.set_option("conv1d.variant", -1); _lib.set_option("conv1x1.variant", 0)
  print("C %4d->%4d: ideal dense %.1f us (hbm 4.5TB/s) %.1f us (mfma 1.2PF) | " % (cin, cout, byts / 4.5e6, fl / 1.2e9) + "  ".join(out), flush=True)
