#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_probe36; mkdir -p $O
timeout 300 python tools/host_profile_jasper.py 2>&1 | grep -v amdgpu.ids | head -40
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o a -- python bench.py --no-transformer --no-other-configs --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 3 > $O/tr.log 2>&1
python tools/trace_gaps.py $(ls $O/tr/*kernel_trace.csv | head -1)
rocprofv3 --kernel-trace --output-format csv -d $O/trt -o a -- python bench.py --only-transformer --steps 6 --warmup 3 > $O/trt.log 2>&1
python tools/trace_gaps.py $(ls $O/trt/*kernel_trace.csv | head -1)
rm -rf $O/tr $O/trt
