#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe41; mkdir -p $O
timeout 900 python -m pytest tests/test_transformer_kernels_gpu.py tests/test_transformer_e2e_gpu.py tests/test_rnn_gpu.py tests/test_boundary.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
  timeout 300 python bench.py --only-transformer --steps 40 --warmup 5 > $O/tr.json 2> $O/tr.err
  python -c "
import json;d=json.load(open('$O/tr.json'));print('transformer:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/tr.err
done
timeout 300 python bench.py --only-nmt --steps 10 --warmup 3 > $O/nmt.json 2> $O/nmt.err; python -c "
import json;d=json.load(open('$O/nmt.json'));print('nmt:', round(d['ms_per_step'],3), 'ms/step')"
OS2S_DENSE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tr -o a -- python bench.py --only-transformer --steps 5 --warmup 3 > $O/prof_tr.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("$O/prof_tr/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0]))); n=8
print("serial streams; sum of kernel durations per step: %.2f ms"%(sum(int(r["TotalDurationNs"]) for r in rows)/n/1e6))
for r in rows[:14]: print("%-80s %5d %8.3f ms/step %8.1f us avg"%(r["Name"][:80], int(r["Calls"])//n, int(r["TotalDurationNs"])/n/1e6, float(r["AverageNs"])/1e3))
PY
