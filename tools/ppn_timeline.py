"""Slot timeline of the NARROW ping-pong conv tiles (conv1d_ppn_kernel<DBG>: s_memtime stamps of waves 0
and 4 of four workgroups, steps 8..31): where a step's cycles go.
Usage: python tools/ppn_timeline.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.os2s_set_option.argtypes = [_lib.ctypes.c_char_p, _lib.ctypes.c_double]
B, T, NS = 32, 840, 24
# mode: 0 normal, 2 no DMA issue in the loop, 4 no fragment reads in the loop, 6 neither (barriers + MFMAs only)
for cin, cout, K, v, prio, mode in [(384, 384, 13, 12, 0, 0), (384, 384, 13, 12, 1, 0), (384, 384, 13, 12, 0, 2),
                                    (384, 384, 13, 12, 0, 4), (384, 384, 13, 12, 0, 6),
                                    (512, 512, 17, 13, 0, 0), (512, 512, 17, 13, 0, 2), (512, 512, 17, 13, 0, 4),
                                    (512, 512, 17, 13, 0, 6)]:
  L.os2s_set_option(b"conv1d.pp_prio", float(prio))
  x = torch.randn(B, T, cin, device=dev).to(torch.bfloat16)
  w = (torch.randn(K, cout, cin, device=dev) * 0.02).to(torch.bfloat16)
  y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
  st = torch.zeros(4 * 2 * NS * 9, dtype=torch.int64, device=dev)
  _lib.set_option("conv1d.variant", v)
  for _ in range(3): capi.conv1d_fwd(x, w, out=y)
  torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d", st.data_ptr(), mode)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  capi.conv1d_fwd(x, w, out=y)
  e1.record(); torch.cuda.synchronize()
  _lib.set_debug_stamps("conv1d", 0, 0)
  _lib.set_option("conv1d.variant", -1)
  t = st.cpu().numpy().reshape(4, 2, NS, 9).astype(np.float64)
  print("C %d->%d K %d variant %d prio %d mode %d: launch %.3f ms" % (cin, cout, K, v, prio, mode, e0.elapsed_time(e1)))
  names = ["rd1", "vmcnt", "dma", "rd2", "valu", "lgkm", "bar", "MFMA", "bar"]
  for wg in range(1):
    for g in range(2):
      a = t[wg, g, 1:NS]
      prev_end = np.concatenate([t[wg, g, 0:NS - 1, 8:9], a], axis=1)
      d = np.diff(prev_end, axis=1)
      step = prev_end[:, 9] - prev_end[:, 0]
      print("  wg %d group %s: step %6.0f cyc (min %5.0f max %5.0f) | " % (wg, "AB"[g], step.mean(), step.min(), step.max()) +
            " ".join("%s %4.0f" % (n, v_) for n, v_ in zip(names, d.mean(0))))
