#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe14
mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/time.log
tail -3 $OUT/time.log
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["achieved"])
print("frontend", d.get("frontend"))
print("cpu", d.get("cpu_baseline"))
print("secondary", {k: d["secondary"].get(k) for k in ("value", "ms_per_step", "roofline")})
print("others", {k: (v.get("ms_per_step"), v.get("value")) for k, v in d.get("other_configs", {}).items()})
PY
