#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_probe16
mkdir -p $OUT
OS2S_GEMM=pp timeout 900 python -m pytest tests/test_transformer_e2e_gpu.py tests/test_transformer_kernels_gpu.py tests/test_beam_search_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for be in lt pp; do
  OS2S_GEMM=$be timeout 600 python bench.py --only-transformer --steps 30 --warmup 10 > $OUT/tr_$be.log 2>&1
  echo "$be: $(tail -1 $OUT/tr_$be.log | cut -c1-260)"
done
