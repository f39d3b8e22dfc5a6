#!/bin/bash
# SQ stall breakdown of the wgrad / fwd conv kernels on the Jasper shapes (own PMC pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wgrad
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p -- python tools/bench_wgrad_shapes.py auto > $OUT/log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:46]+" grid"+r["Grid_Size"]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for k,v in agg.items():
    if "wgrad" not in k: continue
    wc=v["SQ_WAVE_CYCLES"] or 1
    print(k, "n=%d"%n[k], " wait_any %.2f  wait_inst %.2f  active %.2f  wait_lds %.2f | mfma_busy/wavecyc*4 %.3f  lds_conf/lds_active %.3f" % (
      v["SQ_WAIT_ANY"]/wc, v["SQ_WAIT_INST_ANY"]/wc, v["SQ_ACTIVE_INST_ANY"]/wc, v["SQ_WAIT_INST_LDS"]/wc,
      v["SQ_VALU_MFMA_BUSY_CYCLES"]/(wc*4), v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1)))
PY
