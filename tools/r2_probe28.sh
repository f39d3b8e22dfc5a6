#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe28; mkdir -p $O
export TMPDIR=/tmp
OS2S_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o jasper -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing > $O/prof.log 2>&1
ls $O/prof | head -3
