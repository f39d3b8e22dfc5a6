#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe35; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_transformer_kernels_gpu.py tests/test_boundary.py tests/test_transformer_e2e_gpu.py tests/test_nmt_e2e_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/bench_gemm_routes.py 2>&1 | grep -v amdgpu.ids | head -12
for m in transformer nmt; do
  timeout 300 python bench.py --only-$m --steps 20 --warmup 5 > $O/$m.json 2> $O/$m.err
  python -c "
import json;d=json.load(open('$O/$m.json'));print('$m:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/$m.err
done
