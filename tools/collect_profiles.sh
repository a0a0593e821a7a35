#!/bin/bash
# Round evidence on the GPU box: bench line, rocprofv3 kernel statistics of the bench / cfg 3 / cfg 4 legs, HBM counters.
# usage (from the repo root, through gpurun): bash tools/collect_profiles.sh r04     -> gpurun_out/<tag>_*
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
cd /tmp
stats() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $OUT/${TAG}_$name.log 2>&1
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv
}
stats bench python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-multitask --no-extra
stats cfg3 python $ROOT/tools/prof_cfg3.py
stats cfg4 python $ROOT/tools/prof_multitask.py 64
stats cfg4_shard8 python $ROOT/tools/prof_multitask.py shard8
stats small_tasks python $ROOT/tools/train_small.py 24 100 60
KERNEL=matern52_mlp MEAN=linear_mlp stats small_tasks_mlp python $ROOT/tools/train_small.py 24 100 60
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-multitask --no-extra > $OUT/${TAG}_pmc_$ctr.log 2>&1
done
python $ROOT/tools/pmc_to_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_hbm.json 2> $OUT/${TAG}_pmc.err
ls -la $OUT | grep $TAG
