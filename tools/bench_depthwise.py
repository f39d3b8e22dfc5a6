"""Depthwise forward at QuartzNet shapes: register-window kernel (depthwise.variant 1) vs the matrix-core kernel."""
import sys, torch
sys.path.insert(0, ".")
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T = 32, 830
lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32)
lens[0] = T
def setopt(name, v):
  _lib.lib().os2s_set_option(name.encode(), __import__("ctypes").c_double(v))
for C, K in ((256, 33), (256, 39), (512, 51), (512, 63), (512, 75), (1024, 75)):
  x = (torch.randn(B, T, C, generator=g) * (torch.arange(T)[None, :, None] < lens[:, None, None])).to(torch.bfloat16).to(dev)
  w = (torch.randn(K, C, generator=g) * 0.2).to(dev)
  ld = lens.to(dev)
  res = {}
  for variant in (1, -1):
    setopt("depthwise.variant", variant)
    y = capi.depthwise_conv1d_fwd(x, w, in_len=ld)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      y = capi.depthwise_conv1d_fwd(x, w, in_len=ld)
    e1.record(); torch.cuda.synchronize()
    res[variant] = (e0.elapsed_time(e1) / 20 * 1e3, y.float())
  abl = {}
  for m in (1, 2, 4, 3, 5, 6, 7):
    setopt("depthwise.ablate", m)
    y = capi.depthwise_conv1d_fwd(x, w, in_len=ld); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      y = capi.depthwise_conv1d_fwd(x, w, in_len=ld)
    e1.record(); torch.cuda.synchronize()
    abl[m] = e0.elapsed_time(e1) / 20 * 1e3
  setopt("depthwise.ablate", 0)
  print("   ablations (1 = no compute, 2 = no store, 4 = no loads): " + ", ".join("%d: %.1f" % (m, t) for m, t in abl.items()))
  live = float(lens.sum()) * C
  err = float((res[1][1] - res[-1][1]).abs().max())
  print("C %4d K %2d: window kernel %6.1f us, mfma kernel %6.1f us (%.2f TB/s of live r+w), max |diff| %.3e"
        % (C, K, res[1][0], res[-1][0], live * 4 / (res[-1][0] * 1e-6) / 1e12, err))

print("---- weight gradient ----")
for C, K in ((256, 33), (256, 39), (512, 51), (512, 63), (512, 75), (1024, 75)):
  x = (torch.randn(B, T, C, generator=g) * (torch.arange(T)[None, :, None] < lens[:, None, None])).to(torch.bfloat16).to(dev)
  dy = (torch.randn(B, T, C, generator=g) * (torch.arange(T)[None, :, None] < lens[:, None, None])).to(torch.bfloat16).to(dev)
  ld = lens.to(dev)
  res = {}
  for variant in (1, -1):
    setopt("depthwise.variant", variant)
    dw = torch.zeros(K, C, device=dev)
    capi.depthwise_conv1d_wgrad(x, dy, dw, in_len=ld)
    torch.cuda.synchronize()
    first = dw.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      capi.depthwise_conv1d_wgrad(x, dy, dw, in_len=ld)
    e1.record(); torch.cuda.synchronize()
    res[variant] = (e0.elapsed_time(e1) / 20 * 1e3, first)
  abl = {}
  for m in (1, 2, 4, 3, 6, 7):
    setopt("depthwise.ablate", m)
    capi.depthwise_conv1d_wgrad(x, dy, dw, in_len=ld); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      capi.depthwise_conv1d_wgrad(x, dy, dw, in_len=ld)
    e1.record(); torch.cuda.synchronize()
    abl[m] = e0.elapsed_time(e1) / 20 * 1e3
  setopt("depthwise.ablate", 0)
  print("   wgrad ablations (1 = no MFMA phase, 2 = no diagonal sums / atomics, 4 = no loads): " + ", ".join("%d: %.1f" % (m, t) for m, t in abl.items()))
  err = float((res[1][1] - res[-1][1]).abs().max() / res[1][1].abs().max())
  print("C %4d K %2d: window kernel %6.1f us, mfma kernel %6.1f us, max rel diff %.2e" % (C, K, res[1][0], res[-1][0], err))
