#!/bin/bash
# Final-tree evidence of a round in ONE short GPU call (tools/profile_round.sh is the long form with the PMC passes):
#   <tag>_bench_final.json(.err)             python bench.py — the driver's command
#   <tag>_jasper_kernel_stats_final.csv      rocprofv3 --kernel-trace --stats of the Jasper step, bench streams
#   <tag>_jasper_kernel_stats_serial_final.csv   the same with OS2S_WGRAD_STREAM=0 (every kernel alone on the GPU)
#   <tag>_transformer_kernel_stats_serial_final.csv
# Usage on the GPU box: bash tools/profile_final.sh r05
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench_final.json 2> $OUT/${TAG}_bench_final.err
J="python bench.py --no-transformer --no-other-configs --no-cpu-baseline --no-kernel-timing"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o jasper -- $J --steps 8 --warmup 3 > $OUT/ks.log 2>&1
cp $OUT/ks/jasper_kernel_stats.csv $OUT/${TAG}_jasper_kernel_stats_final.csv
OS2S_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kss -o jasper -- $J --steps 8 --warmup 3 > $OUT/kss.log 2>&1
cp $OUT/kss/jasper_kernel_stats.csv $OUT/${TAG}_jasper_kernel_stats_serial_final.csv
OS2S_DENSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tks -o tr -- python bench.py --only-transformer --steps 5 --warmup 3 > $OUT/tks.log 2>&1
cp $OUT/tks/tr_kernel_stats.csv $OUT/${TAG}_transformer_kernel_stats_serial_final.csv
rm -rf $OUT/ks $OUT/kss $OUT/tks
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_final.json"))
print("jasper ms/step", d["ms_per_step"], "frames/s", d["value"], "whole_step_frac", d["roofline"]["whole_step_frac"], "frac", d["roofline"]["frac"])
print("transformer", d["secondary"].get("ms_per_step"), d["secondary"].get("value"))
print({k: (v.get("ms_per_step") or v.get("us_per_step") or v.get("error")) for k, v in d.get("other_configs", {}).items()})
print("cpu", d["cpu_baseline"])
PY
