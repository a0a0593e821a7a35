export TMPDIR=/tmp
ROOT=$PWD
python tools/train_speed.py 24 500 100 60 2>&1 | tail -2
python tools/train_speed.py 24 500 50 60 2>&1 | tail -2
cd /tmp && rm -rf /tmp/trs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trs -- python $ROOT/tools/train_speed.py 24 500 100 60 > /tmp/trs.log 2>&1
f=$(find /tmp/trs -name "*kernel_stats.csv" | head -1)
cut -c1-140 $f | head -30
