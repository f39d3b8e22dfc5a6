#!/bin/bash
# rocprofv3 kernel-trace summary of one of the secondary configs: tools/prof_simple.sh <tag> --only-ds2|--only-nmt|--only-tacotron|--only-quartznet
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_simple_$1
mkdir -p $OUT
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python bench.py "$@" --steps 4 --warmup 2 > $OUT/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $OUT/log | cut -c1-160
