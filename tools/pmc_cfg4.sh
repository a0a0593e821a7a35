set -u
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/s3j; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc4_$ctr
  HBO_PROF_LEVEL=0 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc4_$ctr -- python $ROOT/tools/prof_multitask.py 64 > $OUT/pmc4_$ctr.log 2>&1
done
python $ROOT/tools/pmc_to_json.py $(find /tmp/pmc4_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc4_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $OUT/cfg4_pmc.json 2> $OUT/pmc4.err
rm -rf /tmp/st4; HBO_PROF_LEVEL=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st4 -- python $ROOT/tools/prof_multitask.py 64 > $OUT/st4.log 2>&1
cp $(find /tmp/st4 -name "*kernel_stats.csv" | head -1) $OUT/cfg4_kernel_stats.csv
