"""conv1d wgrad at the Jasper block shapes (B=32, T'=840), lockstep kernel vs ping-pong kernel with
forced split factors, dense and ragged: ms per launch and executed TF/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
B, T = 32, 840
shapes = [(256, 256, 11), (384, 384, 13), (512, 512, 17), (640, 640, 21), (768, 768, 25), (768, 896, 29)]
rng = np.random.RandomState(1234)
dur = rng.uniform(2.0, 16.7, size=B)
lens_np = np.minimum((1 + (dur * 16000).astype(np.int64) // 160 + 1) // 2, T).astype(np.int32)
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
  dy = torch.randn(B, T, cout, device=dev).to(torch.bfloat16)
  dw = torch.zeros(K, cout, cin, device=dev)
  dil = 2 if K == 29 else 1
  _, pl = capi.same_padding(T, K, 1, dil)
  for name, ln in (("dense", None), ("ragged", lens_np)):
    lens = None if ln is None else torch.from_numpy(ln).to(dev)
    live = 1.0 if ln is None else float(ln.sum()) / (B * T)
    fl = 2.0 * B * T * cin * cout * K * live
    out = []
    _lib.set_option("conv1d_wgrad.variant", 0); _lib.set_option("conv1d_wgrad.split", -1)
    t = timeit(lambda: capi.conv1d_wgrad(x, dy, K, dil=dil, pad_left=pl, in_len=lens, out=dw, accumulate=True))
    out.append("lockstep %.3f ms %4.0f TF" % (t, fl / t / 1e9))
    for variant, label in ((1, "pp"), (3, "sw")):      # ping-pong (2 waves / SIMD) vs one wave per SIMD (round 6)
      for f in ((-1, 1, 2, 3, 4, 6, 8, 12, 16) if os.environ.get("OS2S_BENCH_SPLITS") else (-1, 1)):
        _lib.set_option("conv1d_wgrad.variant", variant); _lib.set_option("conv1d_wgrad.split", f)
        t = timeit(lambda: capi.conv1d_wgrad(x, dy, K, dil=dil, pad_left=pl, in_len=lens, out=dw, accumulate=True))
        out.append("%s f%d %.3f" % (label, f, t) + (" %4.0f TF" % (fl / t / 1e9)))
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
    units = ((cout + 127) // 128) * ((cin + 127) // 128) * ((K + 3) // 4)
    print("C %4d->%4d K %2d %-6s units %3d: %s" % (cin, cout, K, name, units, "  ".join(out)), flush=True)
