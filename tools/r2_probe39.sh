#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe39; mkdir -p $O
timeout 900 python -m pytest tests/test_rnn_gpu.py tests/test_ds2_gpu.py tests/test_nmt_e2e_gpu.py tests/test_tacotron_e2e_gpu.py -x -q -m gpu 2>&1 | tail -4
for m in ds2 nmt tacotron; do
  timeout 300 python bench.py --only-$m --steps 5 --warmup 2 > $O/$m.json 2> $O/$m.err
  python -c "
import json;d=json.load(open('$O/$m.json'));print('$m:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/$m.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ds2 -o a -- python bench.py --only-ds2 --steps 3 --warmup 2 > $O/prof_ds2.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("$O/prof_ds2/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(fs[0])))[:8]: print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
