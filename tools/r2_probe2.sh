#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe2
mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv1d_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python tools/bench_conv_split.py > $OUT/split.log 2>&1
cat $OUT/split.log
# SQ counters of the ping-pong kernel (768->768 K25 dense, no split)
cat > /tmp/pmc_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
B, T, C, K = 32, 840, 768, 25
x = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
w = (torch.randn(K, C, C, device=dev) * 0.02).to(torch.bfloat16)
y = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
_lib.lib().os2s_conv1d_set_variant(10); _lib.lib().os2s_conv1d_set_split(1)
for _ in range(5): capi.conv1d_fwd(x, w, out=y)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o c -- python /tmp/pmc_one.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p2 -o c -- python /tmp/pmc_one.py > $OUT/p2.log 2>&1
python - <<PY
import csv, collections, glob
for p in ("p1","p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f: print("no counters", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "conv1d" in k: print(k, {a: "%.4g" % (b / n[(k, a)]) for a, b in v.items()})
PY
