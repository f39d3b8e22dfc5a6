#!/bin/bash
# rocprofv3 kernel-trace summary of the Jasper train step (headline bench, no secondary configs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_jasper_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python bench.py --no-other-configs --no-transformer --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 3 > $OUT/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]: print(r["Name"][:84], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $OUT/log | cut -c1-200
