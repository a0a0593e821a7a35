#!/bin/bash
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $F "$@" -c hyperbo_amd/csrc/chol.hip -o /tmp/cb_chol.o
/opt/rocm/bin/hipcc $F "$@" -c tools/chol_bench.hip -o /tmp/cb_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/cb_main.o /tmp/cb_chol.o -o tools/chol_bench
