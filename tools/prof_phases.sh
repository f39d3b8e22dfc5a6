#!/bin/bash
# kernel trace of the Jasper step with the bench's streams + phase view (tools/trace_phases.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_phases_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o a -- python bench.py --no-other-configs --no-transformer --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 3 > $OUT/log 2>&1
python tools/trace_phases.py $(ls $OUT/*kernel_trace.csv | head -1) | tee $OUT/phases.txt
rm -f $OUT/*kernel_trace.csv
