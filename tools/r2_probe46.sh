#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe46; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_decoder_gpu.py tests/test_tacotron_e2e_gpu.py tests/test_nmt_e2e_gpu.py tests/test_fp8_weights_gpu.py -x -q -m gpu 2>&1 | tail -3
for m in tacotron nmt; do
timeout 300 python bench.py --only-$m --steps 5 --warmup 2 > $O/$m.json 2> $O/$m.err; python -c "
import json;d=json.load(open('$O/$m.json'));print('$m:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/$m.err
done
OS2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o a -- python bench.py --only-tacotron --steps 3 --warmup 2 > $O/prof.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0]))); n=5
for r in rows[:7]: print("%-70s %5d %8.3f ms/step %8.1f us avg"%(r["Name"][:70], int(r["Calls"])//n, int(r["TotalDurationNs"])/n/1e6, float(r["AverageNs"])/1e3))
PY
