"""Micro-benchmark of os2s_conv1d_fwd at Jasper 10x5 layer shapes (B=32, T'=840)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi

dev = torch.device("cuda:0")
from openseq2seq_amd import _lib
variant = int(sys.argv[1]) if len(sys.argv) > 1 else -1
_lib.set_option("conv1d.variant", variant)
print("variant", variant)
B, T = 32, 840
shapes = [(256, 256, 11, 1), (384, 384, 13, 1), (512, 512, 17, 1), (640, 640, 21, 1),
          (768, 768, 25, 1), (768, 896, 29, 2), (896, 1024, 1, 1), (256, 768, 1, 1)]
for cin, cout, K, d in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  lens = torch.full((B,), T, dtype=torch.int32, device=dev)
  nm = capi.conv1d_num_mtiles(B, T)
  stats = torch.empty(nm, 2, cout, device=dev)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  for _ in range(3):
    capi.conv1d_fwd(x, w, dil=d, in_len=lens, stats=stats, out=y)
  torch.cuda.synchronize()
  n = 20
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    capi.conv1d_fwd(x, w, dil=d, in_len=lens, stats=stats, out=y)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / n
  fl = 2.0 * B * T * cin * cout * K
  print("Cin %4d Cout %4d K %2d d %d : %.3f ms  %.1f TF/s" % (cin, cout, K, d, ms, fl / ms / 1e9))

print("wgrad")
for cin, cout, K, d in shapes:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  dy = torch.randn(B, T, cout, device=dev).to(torch.bfloat16)
  lens = torch.full((B,), T, dtype=torch.int32, device=dev)
  dw = torch.zeros(K, cout, cin, device=dev)
  _, pl = capi.same_padding(T, K, 1, d)
  for _ in range(3):
    capi.conv1d_wgrad(x, dy, K, dil=d, pad_left=pl, in_len=lens, out=dw, accumulate=True)
  torch.cuda.synchronize()
  n = 20
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    capi.conv1d_wgrad(x, dy, K, dil=d, pad_left=pl, in_len=lens, out=dw, accumulate=True)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / n
  fl = 2.0 * B * T * cin * cout * K
  print("Cin %4d Cout %4d K %2d d %d : %.3f ms  %.1f TF/s" % (cin, cout, K, d, ms, fl / ms / 1e9))
