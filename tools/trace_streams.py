"""Stream view of a train step from a rocprofv3 --kernel-trace CSV: steps are cut at the optimizer's
apply kernel; per step: wall span, busy time per queue (stream), and per kernel family the summed
durations and launch counts per queue. Usage: python tools/trace_streams.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "mt_apply_kernel" in r[2]]


def fam(n):
  m = re.search(r"os2s::([A-Za-z0-9_]+)", n)
  return m.group(1) if m else n.split("(")[0][-40:]


def union(iv):
  iv = sorted(iv)
  tot, cs, ce = 0, None, None
  for s, e in iv:
    if ce is None or s > ce:
      if ce is not None:
        tot += ce - cs
      cs, ce = s, e
    else:
      ce = max(ce, e)
  return tot + (ce - cs if ce is not None else 0)


n = 0
span = 0
qb = defaultdict(float)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for a, b in zip(marks[1:-1], marks[2:]):
  seg = rows[a + 1:b + 1]
  span += seg[-1][1] - rows[a][1]
  byq = defaultdict(list)
  for s, e, k, q in seg:
    byq[q].append((s, e))
    acc[fam(k)][q][0] += e - s
    acc[fam(k)][q][1] += 1
  for q, iv in byq.items():
    qb[q] += union(iv)
  qb["any"] += union([(r[0], r[1]) for r in seg])
  n += 1
print("%d steps: span %.3f ms; busy: %s" % (n, span / n / 1e6, ", ".join("%s %.2f" % (q, v / n / 1e6) for q, v in sorted(qb.items()))))
tot = {k: sum(v[0] for v in d.values()) for k, d in acc.items()}
for k in sorted(tot, key=lambda k: -tot[k])[:28]:
  print("  %8.3f ms  %-34s %s" % (tot[k] / n / 1e6, k[:34], "  ".join(
      "q%s: %.3f ms / %.1f launches" % (q, v[0] / n / 1e6, v[1] / float(n)) for q, v in sorted(acc[k].items()))))
