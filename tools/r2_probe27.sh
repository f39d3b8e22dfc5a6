#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_probe27; mkdir -p $O
timeout 2400 python -m pytest tests/test_conv1d_gpu.py tests/test_gemm_gpu.py tests/test_jasper_e2e_gpu.py tests/test_jasper_full_size_gpu.py tests/test_speech_data_gpu.py tests/test_transformer_gpu.py -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-transformer --no-other-configs > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"
done
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o jasper -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-transformer --no-other-configs --no-kernel-timing > $O/prof.log 2>&1
ls $O/prof | head -3
