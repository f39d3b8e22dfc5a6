#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe40; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tr -o a -- python bench.py --only-transformer --steps 5 --warmup 3 > $O/prof_tr.log 2>&1
OS2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_js -o a -- python bench.py --no-transformer --no-other-configs --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 3 > $O/prof_js.log 2>&1
python - <<PY
import csv,glob
for m,n in (("tr",8),("js",8)):
  fs=glob.glob("$O/prof_%s/**/*kernel_stats.csv"%m, recursive=True)
  print("==",m,"(per step: total/%d)"%n)
  tot=0
  rows=list(csv.DictReader(open(fs[0])))
  for r in rows: tot+=int(r["TotalDurationNs"])
  print("sum of kernel durations per step: %.2f ms"%(tot/n/1e6))
  for r in rows[:22]: print("%-86s %5d %8.3f ms/step %8.1f us avg %5s%%"%(r["Name"][:86], int(r["Calls"])//n, int(r["TotalDurationNs"])/n/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -1 $O/prof_tr.log | cut -c1-300
