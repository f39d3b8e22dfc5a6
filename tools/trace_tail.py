"""Print the kernels around the optimizer's first launch of a step from a rocprofv3 --kernel-trace CSV: start / end
(us, relative), queue, name — to see what the GPU does in the 0.3 ms before opt_latch_scale_kernel.
Usage: python tools/trace_tail.py <kernel_trace.csv> [n_before] [n_after]"""
import csv
import sys

path = sys.argv[1]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 14
na = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = []
with open(path) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "opt_latch_scale_kernel" in r[2]]
for m in marks[2:4]:
  t0 = rows[m][0]
  print("---- step ending at row %d" % m)
  for s, e, name, q in rows[max(0, m - nb):m + na]:
    print("  %9.1f .. %9.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, name[:90]))
  late = [(s, e, name, q) for s, e, name, q in rows[:m] if e > t0 - 700_000 and s < t0 - 1_200_000]
  for s, e, name, q in late:
    print("  (started earlier) %9.1f .. %9.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, name[:90]))
  qs = sorted(set(r[3] for r in rows))
  print("  queues in the trace:", qs)

# gap after every launch of a named kernel to the next launch on the SAME queue and on ANY queue
if len(sys.argv) > 4:
  pat = sys.argv[4]
  for i, (s, e, name, q) in enumerate(rows):
    if pat in name:
      nxt_same = next((r for r in rows[i + 1:] if r[3] == q), None)
      nxt_any = next((r for r in rows[i + 1:] if r[0] >= e), None)
      running = [r for r in rows[:i] + rows[i + 1:] if r[0] < e + 1000 and r[1] > e]
      print("%s q%s dur %.1f us: next on its queue +%.1f us (%s), next anywhere +%.1f us, %d kernels running at its end"
            % (pat, q, (e - s) / 1e3, (nxt_same[0] - e) / 1e3 if nxt_same else -1, nxt_same[2][:40] if nxt_same else "",
               (nxt_any[0] - e) / 1e3 if nxt_any else -1, len(running)))
