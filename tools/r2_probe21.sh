#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_probe21
timeout 600 python tools/bench_bn_sweep.py > gpurun_out/r2_probe21/sweep.log 2>&1
cat gpurun_out/r2_probe21/sweep.log
