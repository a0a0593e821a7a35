#!/bin/bash
# kernel-trace timeline of the panel chain of one cfg-2 evaluation: bash tools/trace_nll.sh [N] [all] [opt=v ...]
export TMPDIR=/tmp
ROOT=$PWD
N=${1:-8192}; ALL=${2:-}; shift; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/trnll
HBO_PROF_LEVEL=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/trnll -- python $ROOT/tools/prof_nll.py $N "$@" > /tmp/trnll.log 2>&1
f=$(find /tmp/trnll -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_potrf.py $f $ALL | tee $ROOT/gpurun_out/trace_nll.txt
[ -n "${HBO_WINDOW:-}" ] && python $ROOT/tools/trace_window.py $f $HBO_WINDOW | tee $ROOT/gpurun_out/trace_window.txt
