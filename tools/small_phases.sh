#!/bin/bash
# Phase times of small_eval_kernel from early-return builds (tools/tmp_libs/libhbo_stop{1,2,4,5}.so: small.hip with -DHBO_SMALL_STOP=n):
# kernel duration up to the end of the Gram build / the factorisation + inverse / the NLL / K^-1, and of the whole kernel.
cd /tmp; export TMPDIR=/tmp
for v in 1 2 4 5 full; do
  L=$GRAFT_REPO_ROOT/tools/tmp_libs/libhbo_stop$v.so; [ $v = full ] && L=
  rm -rf /tmp/pp; HBO_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/tools/train_small.py ${1:-24} ${2:-100} 30 > /dev/null 2>&1
  echo "stop=$v $(grep small_eval $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | cut -d, -f2-4)"
done
