"""conv1d fwd at the Jasper 10x5 block shapes (B=32, T=840 after the stride-2 layer) for the tile
variants: ms and TF/s per (shape, variant). Usage: python tools/bench_conv_shapes.py [variants...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
shapes = [(32, 840, 256, 256, 11), (32, 840, 384, 384, 13), (32, 840, 512, 512, 17),
          (32, 840, 640, 640, 21), (32, 840, 768, 768, 25), (32, 840, 768, 896, 29), (32, 840, 896, 1024, 1)]
variants = [int(v) for v in sys.argv[1:]] or [3, 5]
res = {}
for v in variants:
  _lib.lib().os2s_conv1d_set_variant(v)
  for B, T, cin, cout, K in shapes:
    x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
    y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
    dil = 2 if K == 29 else 1
    for _ in range(3): capi.conv1d_fwd(x, w, out=y, dil=dil)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): capi.conv1d_fwd(x, w, out=y, dil=dil)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res[(v, cin, cout, K)] = (ms, 2.0 * B * T * cin * cout * K / ms / 1e9)
for B, T, cin, cout, K in shapes:
  print("C %4d->%4d K %2d: " % (cin, cout, K) + "  ".join(
      "v%d %.3f ms %5.0f TF/s" % (v, res[(v, cin, cout, K)][0], res[(v, cin, cout, K)][1]) for v in variants))
