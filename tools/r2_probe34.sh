#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe34; mkdir -p $O
timeout 300 python tools/bench_gemm_routes.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids | head -60
