#!/bin/bash
# round-2 probe 1: ping-pong conv kernel — parity tests, per-shape timings, SQ counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv1d_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python tools/bench_conv_shapes.py 3 5 10 > $OUT/shapes.log 2>&1
cat $OUT/shapes.log
