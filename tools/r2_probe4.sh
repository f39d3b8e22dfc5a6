#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe4
mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv1d_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 300 python tools/pp_timeline.py > $OUT/timeline.log 2>&1
grep -v "wg [123]" $OUT/timeline.log
timeout 600 python tools/bench_conv_split.py > $OUT/split.log 2>&1
cat $OUT/split.log
