"""Ablations of the one-wave-per-SIMD weight-gradient stream (conv1d_wgrad_sw.hpp): the same launch with one
ingredient of the loop removed (results are wrong then; only the time is read) and the shader clock during each run.
Needs a library built with OS2S_EXTRA_HIPFLAGS=-DOS2S_SW_ABLATE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi, _lib
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, T, C, K, d = 32, 840, 768, 25, 1
x = torch.randn(B, T, C, generator=g).to(torch.bfloat16).to(cuda)
dy = torch.randn(B, T, C, generator=g).to(torch.bfloat16).to(cuda)
dw = torch.zeros(K, C, C, device=cuda)
steps = B * 14
names = {0: "full stream", 1: "no LDS-DMA in the loop", 2: "no barrier", 3: "no DMA, no barrier", 4: "no transpose reads",
         5: "no DMA, no reads", 8: "no MFMAs"}
def run(label):
  for _ in range(5): capi.conv1d_wgrad(x, dy, K, dil=d, out=dw, accumulate=True)
  torch.cuda.synchronize()
  probe = capi.clock_probe_start(30e-3 * 1.5e9)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  n = 100
  for _ in range(n): capi.conv1d_wgrad(x, dy, K, dil=d, out=dw, accumulate=True)
  e1.record(); torch.cuda.synchronize()
  mhz = capi.clock_probe_read(probe)
  ms = e0.elapsed_time(e1) / n
  print("%-28s %.3f ms/launch, clock %4.0f MHz, %4.0f cycles per 64-row step" % (label, ms, mhz, ms * 1e-3 * mhz * 1e6 / steps), flush=True)
_lib.set_option("conv1d_wgrad.split", 1)
_lib.set_option("conv1d_wgrad.variant", 1)
run("ping-pong kernel")
_lib.set_option("conv1d_wgrad.variant", 3)
for a in (0, 1, 2, 3, 4, 5, 8):
  _lib.set_option("conv1d_wgrad.sw_ablate", a)
  run("sw: " + names[a])
