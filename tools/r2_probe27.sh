#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe27; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -8 $O/pytest.log
OS2S_BENCH_CONV_TABLE=1 OS2S_BENCH_CONV_EVERY=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs > $O/table.json 2> $O/table.err
grep "^conv" $O/table.err | head -12
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"
