#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe43; mkdir -p $O
OS2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -o a -- python bench.py --only-quartznet --steps 5 --warmup 3 > $O/prof_q.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("$O/prof_q/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0]))); n=8
print("quartznet serial; sum of kernel durations per step: %.2f ms"%(sum(int(r["TotalDurationNs"]) for r in rows)/n/1e6))
for r in rows[:18]: print("%-90s %5d %8.3f ms/step %8.1f us avg"%(r["Name"][:90], int(r["Calls"])//n, int(r["TotalDurationNs"])/n/1e6, float(r["AverageNs"])/1e3))
PY
tail -1 $O/prof_q.log | cut -c1-400
