"""conv1d fwd at chosen (B, T, Cin, Cout, K) for a given tile variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
shapes = [(10, 2176, 768, 768, 25), (32, 1024, 512, 512, 17), (16, 2048, 1024, 1024, 11), (8, 2176, 640, 640, 21)]
for v in (3, 5):
  _lib.lib().os2s_conv1d_set_variant(v)
  for B, T, cin, cout, K in shapes:
    x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
    y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(3): capi.conv1d_fwd(x, w, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): capi.conv1d_fwd(x, w, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nb5 = -(-(B * -(-T // 128)) // 2) * -(-cout // 256)
    print("variant %d B %d T %d C %d->%d K %d: %.3f ms %.0f TF/s (256^2 blocks: %d)" % (
        v, B, T, cin, cout, K, ms, 2.0 * B * T * cin * cout * K / ms / 1e9, nb5))
