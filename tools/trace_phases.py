"""Phase view of a train step from a rocprofv3 --kernel-trace CSV (Jasper bench): steps are cut at
the optimizer's apply kernel, a step into forward (up to the CTC kernels) / backward / optimizer;
per phase: wall span, busy time per queue (stream) and the kernel families' summed durations.
Usage: python tools/trace_phases.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "mt_apply_kernel" in r[2]]


def fam(n):
  for k in ("conv1d_pp_kernel", "conv1d_wgrad_pp_kernel", "conv1d_wgrad_grouped", "conv1d_wgrad_kernel",
            "conv1d_igemm_grouped", "conv1d_igemm_kernel", "bn_act_fwd", "bn_act_bwd_reduce", "bn_bwd_apply",
            "bn_finalize", "bn_bwd_finalize", "ctc", "mt_", "conv_weight_dgrad_copy"):
    if k in n:
      return k
  return n.split("(")[0][-40:]


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


acc = defaultdict(lambda: defaultdict(float))
span = defaultdict(float)
qbusy = defaultdict(lambda: defaultdict(float))
n = 0
for a, b in zip(marks[1:-1], marks[2:]):
  seg = rows[a + 1:b + 1]
  t0 = rows[a][1]
  ctc = [i for i, r in enumerate(seg) if "ctc" in r[2].lower()]
  first_opt = next(i for i, r in enumerate(seg) if "mt_" in r[2])
  cuts = [(0, ctc[0], "forward"), (ctc[0], first_opt, "backward (incl. CTC)"), (first_opt, len(seg), "optimizer")]
  prev_end = t0
  for i0, i1, name in cuts:
    part = seg[i0:i1]
    if not part:
      continue
    end = max(r[1] for r in part)
    span[name] += end - prev_end
    prev_end = end
    byq = defaultdict(list)
    for s, e, k, q in part:
      acc[name][fam(k)] += e - s
      byq[q].append((s, e))
    for q, iv in byq.items():
      qbusy[name][q] += union(iv)
    qbusy[name]["any"] += union([(r[0], r[1]) for r in part])
  n += 1
for name in ("forward", "backward (incl. CTC)", "optimizer"):
  print("%-22s span %7.3f ms | busy(any queue) %7.3f | per queue: %s" % (
      name, span[name] / n / 1e6, qbusy[name]["any"] / n / 1e6,
      ", ".join("%s %.2f" % (q, v / n / 1e6) for q, v in sorted(qbusy[name].items()) if q != "any")))
  for k, v in sorted(acc[name].items(), key=lambda kv: -kv[1])[:10]:
    print("      %8.3f ms  %s" % (v / n / 1e6, k))
