#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe42; mkdir -p $O
timeout 600 python -m pytest tests/test_conv1d_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
timeout 200 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids | grep -A3 "^wgrad" | head -8
timeout 300 python tools/bench_wgrad_shapes.py 2>&1 | grep -v amdgpu.ids | tail -14
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing"
for rep in 1 2; do timeout 600 $B > $O/b.json 2> $O/b.err; python -c "
import json;d=json.load(open('$O/b.json'));print('jasper:', round(d['ms_per_step'],3))"; done
