#!/bin/bash
# kernel trace of a bench configuration with the bench's streams + stream view (tools/trace_streams.py)
# usage: prof_streams.sh <tag> <bench.py flags...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof_streams_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o a -- python bench.py "$@" --steps 5 --warmup 3 > $OUT/log 2>&1
python tools/trace_streams.py $(ls $OUT/*kernel_trace.csv | head -1) | tee $OUT/streams.txt
rm -f $OUT/*kernel_trace.csv
