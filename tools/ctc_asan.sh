#!/bin/bash
# AddressSanitizer + UBSan + LeakSanitizer run of the two host CTC decoders (no GPU needed):
#   bash tools/ctc_asan.sh          (from the repository root; silent except for the harness prints)
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Itools \
    tools/ctc_asan_harness.cpp openseq2seq_amd/csrc/ctc_beam_search.cpp -o /tmp/ctc_asan_harness -lpthread
/tmp/ctc_asan_harness
