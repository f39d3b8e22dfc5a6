#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe25; mkdir -p $O
timeout 300 python tools/conv1x1_phases.py > $O/phases.log 2>&1
cat $O/phases.log
