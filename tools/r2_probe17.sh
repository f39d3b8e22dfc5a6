#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe17
mkdir -p $OUT
timeout 900 python -m pytest tests/test_fp8_weights_gpu.py tests/test_attn_decoder_gpu.py tests/test_tacotron_e2e_gpu.py -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "rel-L2|passed|failed|Error|assert" $OUT/pytest.log | tail -12
