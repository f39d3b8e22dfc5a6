#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profile_r02; mkdir -p $O
timeout 900 python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err
python -c "
import json;d=json.load(open('$O/r02_bench_default.json'));print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4)); print(json.dumps(d['roofline']['by_kernel'], indent=1))"
