#!/bin/bash
# HBM traffic of the dominant kernel (conv1d_igemm, all tile variants) under the bench workload:
# FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only (MI355X_MICROARCH.md, HBM /
# rocprofv3 PMC slots). Writes gpurun_out/pmc_bench_traffic_<tag>/traffic.json; copy it to
# profiles/<round>_pmc_bench_traffic.json — bench.py reports it as roofline.traffic.
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench_traffic_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-transformer --no-other-configs --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o c -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o c -- $CMD > $OUT/w.log 2>&1
python - <<PY
import csv, glob, json
tot = {}
for p, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    s, n = 0.0, 0
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "conv1d_igemm_kernel" in r["Kernel_Name"]:
                s += float(r["Counter_Value"]); n += 1
    tot[name] = (s, n)
f, nf = tot["FETCH_SIZE"]; w, nw = tot["WRITE_SIZE"]
out = {
  "command": "$CMD",
  "kernel": "conv1d_igemm_kernel (all tile variants, fwd + dgrad launches, incl. autotune launches)",
  "launches": nf,
  "FETCH_SIZE_KB_per_launch_raw": f / max(nf, 1),
  "WRITE_SIZE_KB_per_launch_raw": w / max(nw, 1),
  "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE as reported",
  "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
