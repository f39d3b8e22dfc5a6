#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe47; mkdir -p $O
timeout 900 python -m pytest tests/test_conv1d_gpu.py tests/test_jasper_e2e_gpu.py tests/test_sepconv_gpu.py tests/test_boundary.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/bench_dense_shapes.py 2>&1 | grep -A12 "Jasper residual" | grep -v amdgpu
python - <<PY
import ctypes, json, subprocess, os
PY
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing"
for at in 0 1 0 1; do
  OS2S_WGRAD_ATOMICS=$at timeout 600 $B > $O/b.json 2> $O/b.err; python -c "
import json;d=json.load(open('$O/b.json'));print('jasper atomics=$at:', round(d['ms_per_step'],3))"
done
for at in 0 1; do
  OS2S_WGRAD_ATOMICS=$at timeout 300 python bench.py --only-quartznet --steps 10 --warmup 3 > $O/q.json 2> $O/q.err; python -c "
import json;d=json.load(open('$O/q.json'));print('quartznet atomics=$at:', round(d['ms_per_step'],3))"
  OS2S_WGRAD_ATOMICS=$at timeout 300 python bench.py --only-transformer --steps 20 --warmup 5 > $O/t.json 2> $O/t.err; python -c "
import json;d=json.load(open('$O/t.json'));print('transformer atomics=$at:', round(d['ms_per_step'],3))"
done
