#!/bin/bash
# PMC counters for the conv micro-benchmark (separate passes; no trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_conv_$1
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -o c -- python tools/bench_conv.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o c -- python tools/bench_conv.py > $OUT/p2.log 2>&1
ls $OUT/p1 $OUT/p2
python - <<PY
import csv, collections, glob
for p in ("p1","p2"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print("no counters", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40] + " grid=" + r.get("Grid_Size","")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(k, {a: "%.3g" % b for a, b in v.items()})
PY
