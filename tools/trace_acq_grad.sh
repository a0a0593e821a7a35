export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rm -rf /tmp/ag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ag -- python $ROOT/tools/bo_latency.py 8000 64 > /tmp/ag.log 2>&1
f=$(find /tmp/ag -name "*kernel_stats.csv" | head -1)
grep -E "tri_mat|acq_grad|gram_kernel|Name" $f | cut -c1-200
