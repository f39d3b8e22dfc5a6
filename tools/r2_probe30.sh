#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe30; mkdir -p $O
timeout 600 python tools/bench_conv_shapes.py 3 10 > $O/shapes.log 2>&1
cat $O/shapes.log
timeout 600 python tools/bench_wgrad_shapes.py > $O/wgrad.log 2>&1
tail -12 $O/wgrad.log | cut -c1-110
timeout 900 python -m pytest tests/test_conv1d_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"
done
