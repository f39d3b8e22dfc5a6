#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe45; mkdir -p $O
timeout 600 python -m pytest tests/test_logmel_gpu.py tests/test_sepconv_gpu.py tests/test_speech_data_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, json
sys.path.insert(0, ".")
import torch, bench
print(json.dumps(bench.bench_frontend(torch.device("cuda:0"), 32)))
PY
timeout 300 python bench.py --only-quartznet --steps 10 --warmup 3 > $O/q.json 2> $O/q.err; python -c "
import json;d=json.load(open('$O/q.json'));print('quartznet:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/q.err
