#!/bin/bash
# kernel-trace timeline of the fp32 factorisation's panel chain (cfg-3 shape): bash tools/trace_factor32.sh [N] ["opt=v ..."]
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/tr32
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr32 -- python $ROOT/tools/prof_factor32.py ${1:-16384} plain "${2:-}" > /tmp/tr32.log 2>&1
f=$(find /tmp/tr32 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_potrf.py $f | tee $ROOT/gpurun_out/trace_factor32.txt
[ -n "${HBO_WINDOW:-}" ] && python $ROOT/tools/trace_window.py $f $HBO_WINDOW | tee $ROOT/gpurun_out/trace_window32.txt
