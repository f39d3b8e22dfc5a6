#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe44; mkdir -p $O
timeout 600 python -m pytest tests/test_sepconv_gpu.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
timeout 300 python bench.py --only-quartznet --steps 10 --warmup 3 > $O/q.json 2> $O/q.err; python -c "
import json;d=json.load(open('$O/q.json'));print('quartznet:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/q.err
done
OS2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -o a -- python bench.py --only-quartznet --steps 5 --warmup 3 > $O/prof_q.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("$O/prof_q/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0]))); n=8
print("quartznet serial; sum of kernel durations per step: %.2f ms"%(sum(int(r["TotalDurationNs"]) for r in rows)/n/1e6))
for r in rows[:8]: print("%-90s %5d %8.3f ms/step %8.1f us avg"%(r["Name"][:90], int(r["Calls"])//n, int(r["TotalDurationNs"])/n/1e6, float(r["AverageNs"])/1e3))
PY
