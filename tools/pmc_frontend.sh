#!/bin/bash
# HBM-side traffic of the log-mel front end at the bench batch (separate --pmc passes; gfx950 correction as in
# tools/profile_round.sh): writes gpurun_out/pmc_frontend/<tag>_pmc_frontend_traffic.json
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_frontend
mkdir -p $OUT
export OS2S_FRONTEND_NO_SATURATED=1
FP="python bench.py --only-frontend"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ff -o c -- $FP > $OUT/ff.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/fw -o c -- $FP > $OUT/fw.log 2>&1
python - <<PY
import csv, glob, json, collections
def collect(sub, match):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((m for m in match if m in r["Kernel_Name"]), None)
            if k is None: continue
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
corr = "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE as reported (MI355X_MICROARCH.md); both in KB"
names = ["logmel_frames_kernel", "logmel_normalize_kernel", "absmax_kernel", "logmel_stats_kernel"]
f, fn = collect("ff", names); w, wn = collect("fw", names)
calls = fn[names[0]]
per = {k: {"launches": fn[k], "FETCH_SIZE_KB_total": f[k], "WRITE_SIZE_KB_total": w[k],
           "hbm_bytes_per_batch": (2.0 * f[k] + w[k]) * 1024.0 / max(calls, 1)} for k in names if fn[k]}
out = {"command": "OS2S_FRONTEND_NO_SATURATED=1 $FP", "kernels": per, "correction": corr,
       "note": "B = 32 bench batch, %d calls of the stage; hbm_bytes_per_launch = all kernels of ONE call" % calls,
       "hbm_bytes_per_launch": sum((2.0 * f[k] + w[k]) for k in names) * 1024.0 / max(calls, 1)}
json.dump(out, open("$OUT/${TAG}_pmc_frontend_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
