#!/bin/bash
# rocprofv3 kernel-trace summary of Transformer-big beam-search inference (eager loop so
# every kernel is attributed; pass extra bench args after the tag)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_tinfer_$1
mkdir -p $OUT
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python tools/bench_transformer_infer.py --reps 1 "$@" > $OUT/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep rep $OUT/log
