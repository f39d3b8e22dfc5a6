#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe23; mkdir -p $O
OS2S_BENCH_CONV_TABLE=1 OS2S_BENCH_CONV_EVERY=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs > $O/bench.json 2> $O/bench.err
grep "^conv" $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"
