#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe12
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_conv1d_gpu.py tests/test_jasper_e2e_gpu.py tests/test_jasper_full_size_gpu.py tests/test_sepconv_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
OS2S_BENCH_CONV_TABLE=1 OS2S_BENCH_CONV_EVERY=1 timeout 600 python bench.py --no-transformer --no-other-configs --no-cpu-baseline --steps 5 --warmup 3 > $OUT/bench_table.log 2>&1
grep "^conv" $OUT/bench_table.log | head -12
timeout 600 python bench.py --no-transformer --no-other-configs --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-420
