import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi, _lib
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(5 + 768 + 25)
B, T, C, K, d = 32, 840, 768, 25, 1
x = torch.randn(B, T, C, generator=g).to(torch.bfloat16).to(cuda)
dy = torch.randn(B, T, C, generator=g).to(torch.bfloat16).to(cuda)
lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32).to(cuda)
def run(v, f, ln=lens):
  _lib.set_option("conv1d_wgrad.variant", v); _lib.set_option("conv1d_wgrad.split", f)
  o = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=ln)
  torch.cuda.synchronize()
  return o
ref = run(0, -1)
pp = run(1, 1)
print("pp==lock", torch.equal(pp, ref))
for it in range(30):
  a = run(3, 1)
  if it >= 3 and torch.equal(a, ref): continue
  diff = (a - ref).abs()
  nz = (diff > 0)
  print("sw vs lock: differing", int(nz.sum()), "of", nz.numel(), "max", float(diff.max()), "ref max", float(ref.abs().max()))
  if int(nz.sum()):
    idx = nz.nonzero()
    print(" taps", torch.unique(idx[:, 0]).tolist()[:30])
    print(" co blocks(32)", torch.unique(idx[:, 1] // 32).tolist()[:40])
    print(" ci blocks(32)", torch.unique(idx[:, 2] // 32).tolist()[:40])
# clock under each kernel
dw = torch.zeros(K, C, C, device=cuda)
for v, name, xo in ((1, "pp", 0), (1, "pp xcd", 1), (3, "sw", 0), (3, "sw xcd", 1)):
  _lib.set_option("conv1d_wgrad.variant", v); _lib.set_option("conv1d_wgrad.split", 1)
  _lib.set_option("conv1d_wgrad.xcd_order", xo)
  for ln, lname in ((None, "dense"), (lens, "ragged")):
    for _ in range(5): capi.conv1d_wgrad(x, dy, K, dil=d, in_len=ln, out=dw, accumulate=True)
    torch.cuda.synchronize()
    probe = capi.clock_probe_start(30e-3 * 1.5e9)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 100
    for _ in range(n): capi.conv1d_wgrad(x, dy, K, dil=d, in_len=ln, out=dw, accumulate=True)
    e1.record(); torch.cuda.synchronize()
    mhz = capi.clock_probe_read(probe)
    ms = e0.elapsed_time(e1) / n
    live = 1.0 if ln is None else float((torch.minimum((ln + 12 + 63) // 64, torch.tensor(14, device=cuda))).sum()) / (B * 14)
    steps = B * 14 * live
    print("%s %s: %.3f ms/launch, clock %.0f MHz, %.0f cycles per 64-row step (%.0f steps/unit)" % (name, lname, ms, mhz, ms * 1e-3 * mhz * 1e6 / steps, steps))
_lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
