#!/bin/bash
# K = 1 ping-pong weight gradient: tests, Dense GEMM shapes, Transformer-big with the in-tree GEMMs
# (OS2S_GEMM=pp) vs hipBLASLt, kernel stats of the in-tree run, DS2 / Tacotron kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe32; mkdir -p $O
timeout 600 python -m pytest tests/test_conv1d_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5
timeout 300 python tools/bench_dense_shapes.py > $O/dense_shapes.log 2>&1; cat $O/dense_shapes.log
for cfg in "lt 0" "pp 0" "pp 1"; do
  set -- $cfg
  OS2S_GEMM=$1 OS2S_DENSE_WGRAD_STREAM=$2 timeout 300 python bench.py --only-transformer --steps 20 --warmup 5 > $O/tr_$1_$2.json 2> $O/tr_$1_$2.err
  python -c "
import json;d=json.load(open('$O/tr_$1_$2.json'));print('transformer gemm=$1 wgrad_stream=$2:', round(d['ms_per_step'],3), 'ms/step')" || tail -3 $O/tr_$1_$2.err
done
OS2S_GEMM=pp rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tr -o a -- python bench.py --only-transformer --steps 5 --warmup 3 > $O/prof_tr.log 2>&1
for m in ds2 tacotron; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o a -- python bench.py --only-$m --steps 3 --warmup 2 > $O/prof_$m.log 2>&1
done
python - <<PY
import csv,glob
for m in ("tr","ds2","tacotron"):
  fs=glob.glob("$O/prof_%s/**/*kernel_stats.csv"%m, recursive=True)
  if not fs: print("no stats for",m); continue
  print("==",m)
  for r in list(csv.DictReader(open(fs[0])))[:16]: print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
