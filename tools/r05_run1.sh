#!/bin/bash
# round-5 GPU call 1: parity of the narrow ping-pong tiles + per-shape timing + Jasper A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
python -m pytest tests/test_conv1d_gpu.py -x -q -k "pingpong or narrow or tile_variants or dgrad" 2>&1 | tail -15 > gpurun_out/r05a/pytest_conv.log
python -m pytest tests/test_jasper_layerwise_gpu.py -x -q -k "fused" 2>&1 | tail -8 >> gpurun_out/r05a/pytest_conv.log
cat gpurun_out/r05a/pytest_conv.log
python tools/bench_conv_shapes.py 14 12 13 10 > gpurun_out/r05a/conv_shapes.log 2>&1
cat gpurun_out/r05a/conv_shapes.log
J="python bench.py --no-other-configs --no-transformer --no-cpu-baseline --steps 12 --warmup 4"
for rep in 1 2; do
  for c in "1.18,1e6,1e6" "1.18,0.62,0.80" "1.18,0.55,0.70"; do
    $J --pp-cost $c 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('pp-cost $c', 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])" | tee -a gpurun_out/r05a/ab.log
  done
done
