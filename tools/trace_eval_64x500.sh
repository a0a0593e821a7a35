export TMPDIR=/tmp
ROOT=$PWD
python tools/eval_64x500.py | tail -1
cd /tmp && rm -rf /tmp/ev64
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev64 -- python $ROOT/tools/eval_64x500.py > /tmp/ev64.log 2>&1
f=$(find /tmp/ev64 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_multitask.py $f | cut -c1-110
