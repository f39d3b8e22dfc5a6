#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe18
mkdir -p $OUT
for f in "" "--no-fp8"; do
  timeout 600 python bench.py --only-tacotron $f --steps 4 --warmup 2 > $OUT/taco$f.log 2>&1
  echo "fp8 flag [$f]: $(tail -1 $OUT/taco$f.log | cut -c1-330)"
done
