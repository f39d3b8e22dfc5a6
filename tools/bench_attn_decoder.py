"""Micro-benchmark of the attention-decoder loop at Tacotron2 / NMT shapes."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "tacotron"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if which == "tacotron":
  B, S, L, H, M, U, mode, K, F = 32, 200, 2, 1024, 1024, 128, 2, 32, 32   # tacotron_gst.py: encoder 512 + style 512
else:
  B, S, L, H, M, U, mode, K, F = 128, 50, 1, 512, 1024, 512, 1, 0, 0
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
bf = lambda t: t.to(torch.bfloat16).to(dev)
kc = [M + H, 2 * H]
wcat = [bf(rn(4 * H, kc[l], sc=1 / math.sqrt(kc[l]))) for l in range(L)]
dec = capi.AttnDecoder(B, T, S, L, H, M, U, mode, dev, use_bias=True, loc_k=K, loc_f=F, out_keep=0.9,
                       out_seeds=(1, 2))
wq = bf(rn(U, H, sc=0.03))
wqT = wq.t().contiguous()
dec.set_params(wcat, wq, rn(U).to(dev), bias=[None] + [rn(4 * H, sc=0.1).to(dev)] * (L - 1),
               g=torch.ones(1, device=dev), b=rn(U, sc=0.1).to(dev),
               conv_w=rn(K, F, sc=0.3).to(dev) if mode == 2 else None,
               conv_b=rn(F, sc=0.1).to(dev) if mode == 2 else None,
               dense_w=rn(F, U, sc=0.3).to(dev) if mode == 2 else None)
lens = torch.randint(S // 2, S + 1, (B,), generator=g, dtype=torch.int32).to(dev)
dec.set_inputs(bf(rn(B, T, 4 * H)), bf(rn(B, S, U)), bf(rn(B, S, M)), lens)
wcatT = [w.t().contiguous() for w in wcat]
dy, dc = bf(rn(B, T, H)), bf(rn(B, T, M))
z = lambda *s: torch.zeros(*s, device=dev)
def run():
  dec.forward()
  torch.cuda.synchronize(); t1 = time.perf_counter()
  dec.backward(wcatT, wqT, dy_top=dy, dctx_ext=dc, dv=z(U), dg=z(1), dconv_w=z(K, F) if mode == 2 else None,
               dconv_b=z(F) if mode == 2 else None, ddense_w=z(F, U) if mode == 2 else None)
  torch.cuda.synchronize()
  return t1
run()
torch.cuda.synchronize(); t0 = time.perf_counter()
t1 = run()
t2 = time.perf_counter()
print("%s T=%d: fwd %.1f us/step, bwd %.1f us/step" % (which, T, (t1 - t0) / T * 1e6, (t2 - t1) / T * 1e6))

import ctypes
from openseq2seq_amd import _lib
arr = (ctypes.c_longlong * 32)()
try:
  fn = _lib.lib().os2s_debug_attn_phases
  fn(arr)
  for k, name in ((0, "fwd"), (1, "bwd")):
    v = [arr[k * 16 + i] for i in range(8)]
    print(name, "phase cycles:", [v[i + 1] - v[i] for i in range(6)], "total", v[6] - v[0])
except AttributeError:
  pass
