#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe13
mkdir -p $OUT
OS2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o jasper -- python bench.py --no-other-configs --no-transformer --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 3 > $OUT/prof.log 2>&1
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
head -22 "$F" | cut -c1-150
