#!/bin/bash
# rocprofv3 kernel-trace + stats of bench.py (run on the GPU box through gpurun).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o jasper -- python bench.py "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name '*stats*' | head
F=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -40 "$F"
tail -1 "$OUT/bench.log" | cut -c1-400
