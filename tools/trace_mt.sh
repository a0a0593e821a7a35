#!/bin/bash
# kernel-trace listing of one cfg-4 evaluation: bash tools/trace_mt.sh [T|shard8] [opt=v ...]
export TMPDIR=/tmp
ROOT=$PWD
T=${1:-64}; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/trmt
rocprofv3 --kernel-trace --output-format csv -d /tmp/trmt -- python $ROOT/tools/prof_multitask.py $T "$@" > /tmp/trmt.log 2>&1
f=$(find /tmp/trmt -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_multitask.py $f > $ROOT/gpurun_out/trace_mt.txt
wc -l $ROOT/gpurun_out/trace_mt.txt
