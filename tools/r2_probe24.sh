#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe24; mkdir -p $O
timeout 900 python -m pytest tests/test_conv1d_gpu.py tests/test_gemm_gpu.py -x -q > $O/tests.log 2>&1
tail -25 $O/tests.log
