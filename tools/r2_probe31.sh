#!/bin/bash
# A/B on ONE box: ping-pong thresholds (conv Cout >= 320 vs 448; wgrad units >= 8 vs 40)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe31; mkdir -p $O
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing"
for rep in 1 2; do
for cfg in "448 40" "320 40" "448 8" "320 8"; do
  set -- $cfg
  OS2S_PP_MIN_COUT=$1 OS2S_WGRAD_PP_MIN_UNITS=$2 timeout 600 $B > $O/b.json 2> $O/b.err
  python -c "
import json;d=json.load(open('$O/b.json'));print('conv>=$1 wgrad>=$2 rep $rep:', round(d['ms_per_step'],3))"
done; done
