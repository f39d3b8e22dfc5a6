"""conv1d fwd / dgrad / wgrad at Jasper shapes with the ragged lengths of the bench batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openseq2seq_amd import capi
dev = torch.device("cuda:0")
B, T = 32, 808
rng = np.random.RandomState(1234)
dur = rng.uniform(2.0, 16.7, size=B)
lens_np = np.minimum((1 + (dur * 16000).astype(np.int64) // 160 + 1) // 2, T).astype(np.int32)
print("live frame fraction %.3f" % (lens_np.sum() / (B * T)))
def timeit(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for cin, cout, K, d in [(256, 256, 11, 1), (512, 512, 17, 1), (768, 768, 25, 1), (768, 768, 1, 1)]:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  dy = torch.randn(B, T, cout, device=dev).to(torch.bfloat16)
  full = torch.full((B,), T, dtype=torch.int32, device=dev)
  rag = torch.from_numpy(lens_np).to(dev)
  nm = capi.conv1d_num_mtiles(B, T)
  stats = torch.empty(nm, 2, cout, device=dev)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  dw = torch.zeros(K, cout, cin, device=dev)
  _, pl = capi.same_padding(T, K, 1, d)
  fl = 2.0 * B * T * cin * cout * K
  res = []
  for name, ln in (("full", full), ("ragged", rag)):
    tf = timeit(lambda: capi.conv1d_fwd(x, w, dil=d, in_len=ln, stats=stats, out=y))
    td = timeit(lambda: capi.conv1d_fwd(dy, w, dil=d, out=y, out_len=ln))
    tw = timeit(lambda: capi.conv1d_wgrad(x, dy, K, dil=d, pad_left=pl, in_len=ln, out=dw, accumulate=True))
    res.append((name, tf, td, tw))
  print("Cin %d Cout %d K %d:" % (cin, cout, K), " ".join("%s fwd %.3f dgrad %.3f wgrad %.3f ms |" % r for r in res),
        "dense TF/s fwd %.0f" % (fl / res[0][1] / 1e9))
