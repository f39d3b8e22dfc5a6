#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe7
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv1d_gpu.py -x -q -k wgrad > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 600 python tools/bench_wgrad_shapes.py > $OUT/wgrad.log 2>&1
cat $OUT/wgrad.log
