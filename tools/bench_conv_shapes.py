"""conv1d fwd at the Jasper 10x5 block shapes (B=32, T=840 after the stride-2 layer) for the tile
variants, dense and with the ragged lengths of the bench batch: ms and executed TF/s per
(shape, variant). Usage: python tools/bench_conv_shapes.py [variants...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
_lib.lib().os2s_set_option.argtypes = [_lib.ctypes.c_char_p, _lib.ctypes.c_double]
_lib.lib().os2s_set_option(b"conv1d.pp_prio", float(os.environ.get("OS2S_PP_PRIO", "0")))
B, T = 32, 840
shapes = [(256, 256, 11), (256, 384, 13), (384, 384, 13), (384, 512, 17), (512, 512, 17), (512, 640, 21),
          (640, 640, 21), (640, 768, 25), (768, 768, 25), (768, 896, 29)]
variants = [int(v) for v in sys.argv[1:]] or [14, 12, 13, 10]
rng = np.random.RandomState(1234)
dur = rng.uniform(2.0, 16.7, size=B)
lens_np = np.minimum((1 + (dur * 16000).astype(np.int64) // 160 + 1) // 2, T).astype(np.int32)
live = float(lens_np.sum()) / (B * T)
print("ragged batch: live frame fraction %.3f, live 128-row windows %d" % (live, int(((lens_np + 127) // 128).sum())))
res = {}
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
for cin, cout, K in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  nm = capi.conv1d_num_mtiles(B, T)
  stats = torch.empty(nm, 2, cout, device=dev)
  rag = torch.from_numpy(lens_np).to(dev)
  dil = 2 if K == 29 else 1
  for v in variants:
    _lib.set_option("conv1d.variant", v)
    ms_d = timeit(lambda: capi.conv1d_fwd(x, w, out=y, dil=dil, stats=stats))
    ms_r = timeit(lambda: capi.conv1d_fwd(x, w, out=y, dil=dil, stats=stats, in_len=rag))
    fl = 2.0 * B * T * cin * cout * K
    res[(v, cin, cout, K)] = (ms_d, fl / ms_d / 1e9, ms_r, fl * live / ms_r / 1e9)
  _lib.set_option("conv1d.variant", -1)
  print("C %4d->%4d K %2d: " % (cin, cout, K) + "  ".join(
      "v%d %.3f ms %4.0f TF | rag %.3f ms %4.0f TF(live)" % ((v,) + res[(v, cin, cout, K)]) for v in variants), flush=True)
