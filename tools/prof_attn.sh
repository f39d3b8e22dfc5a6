#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_attn_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python tools/bench_attn_decoder.py $1 ${2:-64} > $OUT/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]: print(r["Name"][:50], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $OUT/log
