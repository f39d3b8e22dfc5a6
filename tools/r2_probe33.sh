#!/bin/bash
# in-tree GEMMs as the default back end: full GPU suite, Dense shapes, Transformer-big sustained
# (300 steps) with hipBLASLt / in-tree / in-tree + dW side stream, other configs with both back ends
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe33; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/bench_dense_shapes.py > $O/dense_shapes.log 2>&1; grep -v amdgpu.ids $O/dense_shapes.log | head -18
for cfg in "lt 0 20" "pp 0 20" "pp 1 20" "lt 0 300" "pp 0 300" "pp 1 300"; do
  set -- $cfg
  OS2S_GEMM=$1 OS2S_DENSE_WGRAD_STREAM=$2 timeout 300 python bench.py --only-transformer --steps $3 --warmup 5 > $O/tr_$1_$2_$3.json 2> $O/tr_$1_$2_$3.err
  python -c "
import json;d=json.load(open('$O/tr_$1_$2_$3.json'));print('transformer gemm=$1 wgrad_stream=$2 steps=$3:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/tr_$1_$2_$3.err
done
for m in nmt ds2 tacotron; do
  for be in lt pp; do
    OS2S_GEMM=$be timeout 300 python bench.py --only-$m --steps 5 --warmup 2 > $O/${m}_$be.json 2> $O/${m}_$be.err
    python -c "
import json;d=json.load(open('$O/${m}_$be.json'));print('$m gemm=$be:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/${m}_$be.err
  done
done
