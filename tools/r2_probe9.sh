#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe9
mkdir -p $OUT
timeout 300 python tools/pp_timeline.py > $OUT/timeline.log 2>&1
grep -A2 "^wgrad" $OUT/timeline.log
