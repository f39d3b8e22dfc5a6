#!/bin/bash
# Instruction counters of the Transformer attention kernels at the bench shape (tools/bench_attention.py):
# what the 19 / 41 us per launch are made of. Usage on the GPU box: bash tools/pmc_attention.sh r03
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_attention
mkdir -p $OUT
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$T -o c -- python tools/bench_attention.py --iters 3 > $OUT/$T.log 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_fwd_kernel" not in n and "attn_bwd_kernel" not in n: continue
        k = ("attn_fwd_kernel" if "attn_fwd" in n else "attn_bwd_kernel") + ("<dropout>" if "Lb1" in n or "<true>" in n else "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"command": "python tools/bench_attention.py --iters 3 (B=256 sequences, lengths U[8,56], 16 heads, keep 0.9)",
       "note": "per-launch means over the self / causal / cross launches; SQ_* summed over all SIMDs", "per_kernel": {}}
for k, d in agg.items():
    e = {c: v[0] / v[1] for c, v in d.items()}
    if e.get("SQ_WAVES"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA"):
            if c in e: e[c + "_per_wave"] = e[c] / e["SQ_WAVES"]
    out["per_kernel"][k] = e
json.dump(out, open("$OUT/${TAG}_attention_pmc.json", "w"), indent=1)
print(json.dumps({k: {c: round(v) for c, v in e.items() if c.endswith("_per_wave")} for k, e in out["per_kernel"].items()}))
PY
