#!/bin/bash
# A/B: grouped 1x1 on the ping-pong tile (default for wide groups) vs lockstep tile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe26; mkdir -p $O
cat > /tmp/ab.py <<'PY'
import sys, runpy, ctypes
from openseq2seq_amd import _lib
v = int(sys.argv[1]); sys.argv = ["bench.py"] + sys.argv[2:]
_lib.lib().os2s_conv1x1_set_variant(v)
runpy.run_path("bench.py", run_name="__main__")
PY
for rep in 1 2; do
for v in 1 0; do
  PYTHONPATH=$GRAFT_REPO_ROOT timeout 600 python /tmp/ab.py $v --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing > $O/v$v.$rep.json 2> $O/v$v.$rep.err
  python -c "
import json;d=json.load(open('$O/v$v.$rep.json'));print('variant $v rep $rep', d['ms_per_step'])"
done; done
