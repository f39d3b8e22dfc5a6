#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe15
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 600 python tools/bench_gemm.py > $OUT/gemm.log 2>&1
cat $OUT/gemm.log
