#!/bin/bash
# rocprofv3 kernel-trace summary of the Transformer-big train step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_transformer_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python bench.py --only-transformer --steps 5 --warmup 3 > $OUT/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:24]: print(r["Name"][:84], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $OUT/log | cut -c1-200
