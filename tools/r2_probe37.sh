#!/bin/bash
# s_setprio(1) around the MFMA runs of the ping-pong conv kernel (-DOS2S_PP_PRIO=1): timeline + bench A/B on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe37; mkdir -p $O
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing"
run() { timeout 600 $B > $O/b.json 2> $O/b.err; python -c "
import json;d=json.load(open('$O/b.json'));print('$1:', round(d['ms_per_step'],3))"; }
run base; run base
timeout 200 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids | sed -n 1,3p
touch openseq2seq_amd/csrc/conv1d_igemm.hip
OS2S_EXTRA_HIPFLAGS=-DOS2S_PP_PRIO=1 python -c "
import sys; sys.path.insert(0,'.')
from openseq2seq_amd import build; build.build_hip(verbose=False)" 2>&1 | tail -2
run prio; run prio
timeout 200 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids | sed -n 1,3p
