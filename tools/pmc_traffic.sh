#!/bin/bash
# HBM traffic counters (separate passes) for the conv micro-benchmark.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic_$1
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o c -- python tools/bench_conv.py > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o c -- python tools/bench_conv.py > $OUT/w.log 2>&1
python - <<PY
import csv, collections, glob
res = collections.defaultdict(dict)
for p, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print("no counters", p); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name: continue
        if "os2s" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"][:34] + " grid=" + r.get("Grid_Size", "")
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, v in agg.items():
        res[k][name] = v[0] / v[1]
for k, v in res.items():
    print(k, {a: "%.1f KB/launch" % b for a, b in v.items()})
PY
