#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe19
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ds2 -o ds2 -- python bench.py --only-ds2 --steps 3 --warmup 2 > $OUT/ds2.log 2>&1
F=$(find $OUT/ds2 -name '*kernel_stats.csv' | head -1)
head -12 "$F" | cut -c1-160
tail -1 $OUT/ds2.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/taco -o taco -- python bench.py --only-tacotron --steps 2 --warmup 1 > $OUT/taco.log 2>&1
F=$(find $OUT/taco -name '*kernel_stats.csv' | head -1)
head -12 "$F" | cut -c1-160
