#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe11
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_jasper_full_size_gpu.py tests/test_optimizer_gpu.py -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^$" $OUT/pytest.log | tail -25
