#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe10
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --no-transformer --no-other-configs --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-900
OS2S_WGRAD_STREAM=0 timeout 600 python bench.py --no-transformer --no-other-configs --no-cpu-baseline > $OUT/bench_nostream.log 2>&1
tail -1 $OUT/bench_nostream.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o jasper -- python bench.py --no-other-configs --no-transformer --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 3 > $OUT/prof.log 2>&1
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
head -24 "$F" | cut -c1-200
