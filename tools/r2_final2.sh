#!/bin/bash
# end of round: whole GPU suite + smoke, then the driver's bench command and the QuartzNet stats with the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profile_r02; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err
tail -c 300 $O/r02_bench_default.json; echo
OS2S_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/o_quartznet -o k -- python bench.py --only-quartznet --steps 3 --warmup 2 > $O/o_quartznet.log 2>&1
cp $O/o_quartznet/k_kernel_stats.csv $O/r02_quartznet_kernel_stats.csv
