"""Slot timeline of the ping-pong conv kernel (s_memtime stamps of waves 0 and 4 of four
workgroups): where a step's cycles go — LOAD, barrier waits, MFMA issue, the vmcnt(0) drain.
Usage: python tools/pp_timeline.py [fixed_w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
B, T = 32, 840
for cin, cout, K, fixed in [(768, 768, 25, 0), (768, 768, 25, 1), (512, 512, 17, 0)]:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  st = torch.zeros(4 * 2 * 48 * 9, dtype=torch.int64, device=dev)
  _lib.set_option("conv1d.variant", 10); _lib.set_option("conv1d.split", 1)
  for _ in range(3): capi.conv1d_fwd(x, w, out=y)
  torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d", st.data_ptr(), fixed)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  capi.conv1d_fwd(x, w, out=y)
  e1.record(); torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d", 0, 0)
  _lib.set_option("conv1d.variant", -1); _lib.set_option("conv1d.split", -1)
  t = st.cpu().numpy().reshape(4, 2, 48, 9).astype(np.float64)
  print("C %d->%d K %d fixed_w %d: launch %.3f ms" % (cin, cout, K, fixed, e0.elapsed_time(e1)))
  names = ["LOADe(reads)", "bar", "COMPe", "bar", "LOADo(reads)", "vmcnt", "bar", "COMPo", "bar"]
  for wg in range(4):
    for g in range(2):
      a = t[wg, g, 4:44]                      # skip the pipeline fill
      prev_end = np.concatenate([t[wg, g, 3:43, 8:9], a], axis=1)   # [end of previous step, 9 stamps]
      d = np.diff(prev_end, axis=1)           # 9 durations per step
      step = prev_end[:, 9] - prev_end[:, 0]
      print("  wg %d group %s: step %6.0f cyc (min %5.0f max %5.0f) | " % (wg, "AB"[g], step.mean(), step.min(), step.max()) +
            " ".join("%s %4.0f" % (n, v) for n, v in zip(names, d.mean(0))))

# ---- weight-gradient kernel -----------------------------------------------------------------
for cin, cout, K, mode in [(768, 768, 25, 0), (768, 768, 25, 1), (768, 768, 25, 2), (768, 768, 25, 3), (768, 768, 25, 4), (768, 768, 25, 7)]:
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  dy = torch.randn(B, T, cout, device=dev).to(torch.bfloat16)
  st = torch.zeros(4 * 2 * 48 * 10, dtype=torch.int64, device=dev)
  _lib.set_option("conv1d_wgrad.variant", 1); _lib.set_option("conv1d_wgrad.split", 1)
  for _ in range(3): capi.conv1d_wgrad(x, dy, K)
  torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d_wgrad", st.data_ptr(), mode)
  capi.conv1d_wgrad(x, dy, K)
  torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d_wgrad", 0, 0)
  _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  t = st.cpu().numpy().reshape(4, 2, 48, 10).astype(np.float64)
  print("wgrad C %d->%d K %d dbg_mode %d (1 no dY DMA, 2 no X DMA, 4 frozen cursor)" % (cin, cout, K, mode))
  names = ["LOADe", "bar", "COMPe", "bar", "LOADo(rd+vm)", "dma", "lgkm", "bar", "COMPo", "bar"]
  for wg in range(1):
    for g in range(2):
      a = t[wg, g, 4:44]
      prev_end = np.concatenate([t[wg, g, 3:43, 9:10], a], axis=1)
      d = np.diff(prev_end, axis=1)
      step = prev_end[:, 10] - prev_end[:, 0]
      print("  wg %d group %s: step %6.0f cyc | " % (wg, "AB"[g], step.mean()) +
            " ".join("%s %4.0f" % (n, v) for n, v in zip(names, d.mean(0))))
