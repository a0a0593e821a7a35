#!/bin/bash
# usage: tools/build_gemm_bench.sh [extra hipcc flags for gemm.hip, e.g. -DVARIANT=1]
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -munsafe-fp-atomics -DHBO_GEMM_DEBUG"
/opt/rocm/bin/hipcc $F "$@" -c hyperbo_amd/csrc/gemm.hip -o /tmp/gb_gemm.o
/opt/rocm/bin/hipcc $F -c tools/gemm_bench.hip -o /tmp/gb_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/gb_main.o /tmp/gb_gemm.o -o tools/gemm_bench
