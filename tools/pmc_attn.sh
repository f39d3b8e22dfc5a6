#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
python tools/bench_attn_decoder.py tacotron 64
python tools/bench_attn_decoder.py nmt 50
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$T -o c -- python tools/bench_attn_decoder.py tacotron 16 > $OUT/$T.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "os2s::ad_" not in r["Kernel_Name"]: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k, {c: round(v[0] / v[1]) for c, v in d.items()})
PY
