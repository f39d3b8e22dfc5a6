import csv, sys, re
rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^.*?os2s::", "", r["Kernel_Name"])[:40], r["Queue_Id"]))
rows.sort()
ap = [r for r in rows if r[2].startswith("mt_apply")]
print("apply launches", len(ap))
# last 16 apply launches: what overlaps them
for a in ap[-16:]:
  ov = [(r[2], r[3], min(a[1], r[1]) - max(a[0], r[0])) for r in rows if r is not a and r[0] < a[1] and r[1] > a[0]]
  print("apply q%s %.0f us: overlaps %s" % (a[3], (a[1] - a[0]) / 1e3, [(n, q, "%.0f" % (d / 1e3)) for n, q, d in ov][:6]))
