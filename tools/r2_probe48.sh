#!/bin/bash
# phase timers of the attention-decoder kernels (rebuild attn_decoder.hip with -DOS2S_ATTN_PHASE_TIMERS on the box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
touch openseq2seq_amd/csrc/attn_decoder.hip
OS2S_EXTRA_HIPFLAGS=-DOS2S_ATTN_PHASE_TIMERS python -c "
import sys; sys.path.insert(0,'.')
from openseq2seq_amd import build; build.build_hip(verbose=False)" 2>&1 | tail -2
timeout 200 python tools/bench_attn_decoder.py tacotron 64 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/bench_attn_decoder.py nmt 50 2>&1 | grep -v amdgpu.ids
