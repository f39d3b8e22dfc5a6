"""Phase breakdown of conv1x1_pp_kernel work units (stamps via os2s_set_debug_stamps) for a Jasper
block-end grouped launch: entry -> decoded -> pipeline filled -> main loop -> epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseq2seq_amd import capi, _lib

dev = torch.device("cuda:0")
B = int(os.environ.get("PH_B", "32"))
NG = int(os.environ.get("PH_GROUPS", "10"))
rng = np.random.RandomState(0)
lens_np = (rng.uniform(2.0, 16.7, B) * 50).astype(np.int32) + 1
T = int(-(-lens_np.max() // 16) * 16)
lens = torch.from_numpy(lens_np).to(dev)
L = _lib.lib()
nm = capi.conv1d_num_mtiles(B, T)
cins = [256, 256, 256, 384, 384, 512, 512, 640, 640, 768][10 - NG:]
cout = 768
for variant in (1, 2):
  items = []
  for cin in cins:
    items.append(dict(x=torch.randn(B, T, cin, device=dev).bfloat16(), w=(torch.randn(1, cout, cin, device=dev) * 0.05).bfloat16(),
                      y=torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16), stats=torch.empty(nm, 2, cout, device=dev)))
  _lib.set_option("conv1x1.variant", variant)
  for _ in range(3):
    capi.conv1x1_fwd_grouped(items, in_len=lens)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
    capi.conv1x1_fwd_grouped(items, in_len=lens)
  e1.record(); torch.cuda.synchronize()
  print("variant %d: %.1f us / launch" % (variant, e0.elapsed_time(e1) / 5 * 1e3))
  if variant == 2:
    STRIDE = int(os.environ.get("PH_STRIDE", "8"))     # 24 with a -DOS2S_EPI_STAMPS build
    st = torch.zeros(2048 * STRIDE, dtype=torch.int64, device=dev)
    _lib.set_debug_stamps("conv1d", st.data_ptr(), 0)
    e0.record()
    capi.conv1x1_fwd_grouped(items, in_len=lens)
    e1.record(); torch.cuda.synchronize()
    _lib.set_debug_stamps("conv1d", 0, 0)
    us = e0.elapsed_time(e1) * 1e3
    full = st.cpu().numpy().reshape(2048, STRIDE).astype(np.float64)
    t = full[:, :8]
    ok = t[:, 4] > 0
    t = t[ok]
    if STRIDE > 8:
      e = full[ok][:, 8:24]
      n = int((e[0] > 0).sum())
      base = t[ok][:, 3:4]
      print("  epilogue stamps (cycles after main loop end), mean over units:", " ".join("%6.0f" % v for v in (e[:, :n] - base).mean(0)))
    print("stamped units %d; launch %.1f us (counters are per XCD: only differences inside a unit are used)" % (ok.sum(), us))
    for steps in sorted(set(t[:, 5].astype(int))):
      m = t[t[:, 5] == steps]
      d = np.diff(m[:, :5], axis=1)
      print("  steps %2d (%4d units) ticks: decode %7.0f fill %7.0f loop %8.0f (%6.0f/step) epilogue %7.0f | total %8.0f"
            % (steps, len(m), d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 2].mean() / steps, d[:, 3].mean(), d.sum(1).mean()))
    # units per CU: same XCD counter -> order the units of XCD x by start tick; gaps between the
    # end of a unit and the start of the next one on the same CU cannot be told apart from other CUs,
    # so report the XCD-level busy span instead
    for x in range(2):
      m = t[(np.arange(len(ok))[ok] % 8) == x]
      print("  XCD %d: %d units, span %.0f ticks, sum of unit totals / 32 CUs = %.0f ticks"
            % (x, len(m), m[:, 4].max() - m[:, 0].min(), (m[:, 4] - m[:, 0]).sum() / 32))
_lib.set_option("conv1x1.variant", 0)
