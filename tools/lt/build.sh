#!/bin/bash
# builds the hipBLASLt comparison library (A/B runs only; see gemm_lt.hip)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared gemm_lt.hip -o libos2s_lt.so \
  -Wno-unused-value -Wno-unused-result -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib "$@"
