"""GPU idle time inside a train step from a rocprofv3 --kernel-trace CSV: steps are delimited by
the optimizer's apply kernel (one per step); per step: wall span, union of kernel intervals over
all queues (busy), idle = span - busy, and the kernels in front of which the longest idle gaps
sit. Usage: python tools/trace_gaps.py <kernel_trace.csv> [marker substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "mt_apply_kernel"
rows = []
with open(path) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 3:
  sys.exit("fewer than 3 step markers (%s) in the trace" % marker)
gaps = defaultdict(lambda: [0, 0.0])
spans, busys = [], []
for a, b in zip(marks[1:-1], marks[2:]):       # steps between consecutive markers (skip the first)
  seg = rows[a + 1:b + 1]
  t0, t1 = rows[a][1], seg[-1][1]
  busy, cur_s, cur_e = 0, None, None
  for s, e, name, q in seg:
    s = max(s, t0)
    if cur_e is None:
      if s > t0:
        gaps[name][0] += 1; gaps[name][1] += s - t0
      cur_s, cur_e = s, e
    elif s > cur_e:
      busy += cur_e - cur_s
      gaps[name][0] += 1; gaps[name][1] += s - cur_e
      cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  busy += cur_e - cur_s
  spans.append(t1 - t0); busys.append(busy)
n = len(spans)
print("%d steps: span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%)" % (
    n, sum(spans) / n / 1e6, sum(busys) / n / 1e6, (sum(spans) - sum(busys)) / n / 1e6,
    100.0 * (sum(spans) - sum(busys)) / sum(spans)))
print("idle gaps by the kernel that ends them (per step):")
for name, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
  print("  %8.1f us  %5.1f x  %s" % (t / n / 1e3, c / n, name[:100]))
