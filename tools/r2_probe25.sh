#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe25; mkdir -p $O
PH_STRIDE=24 PH_B=32 PH_GROUPS=10 timeout 300 python tools/conv1x1_phases.py 2>&1 | grep -v amdgpu.ids > $O/phases4.log
cat $O/phases4.log
timeout 900 python -m pytest tests/test_conv1d_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -5
