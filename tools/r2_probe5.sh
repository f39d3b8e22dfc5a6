#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe5
mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv1d_gpu.py -x -q -k pingpong > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python tools/bench_conv_split.py > $OUT/split.log 2>&1
cat $OUT/split.log
echo "==== PRIO=1 build"
OS2S_EXTRA_HIPFLAGS=-DOS2S_PP_PRIO=1 python -c "from openseq2seq_amd import build; build.build_hip(force=True, only=['conv1d_igemm.hip'])" > $OUT/build_prio.log 2>&1
timeout 600 python tools/bench_conv_split.py > $OUT/split_prio.log 2>&1
cat $OUT/split_prio.log
