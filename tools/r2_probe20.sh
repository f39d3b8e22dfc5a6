#!/bin/bash
# psf spectrogram front end: parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_probe20
timeout 600 python -m pytest tests/test_psf_spectrogram_gpu.py tests/test_speech_data_gpu.py -x -q > gpurun_out/r2_probe20/tests.log 2>&1
tail -30 gpurun_out/r2_probe20/tests.log
