#!/bin/bash
# in-situ: tests touching BN + bench + kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_probe22; mkdir -p $O
timeout 900 python -m pytest tests/test_batchnorm_gpu.py tests/test_jasper_e2e_gpu.py tests/test_jasper_full_size_gpu.py -x -q > $O/tests.log 2>&1
tail -5 $O/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs > $O/bench.json 2> $O/bench.err
cat $O/bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o jasper -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing > $O/prof.log 2>&1
ls $O/prof | head
