#!/bin/bash
# max / avg duration of the panel kernels in one rocprofv3 kernel trace of the headline evaluation with context options
# usage: tools/stall_probe.sh "opt=val opt=val"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o sp -- python /root/repo/tools/prof_nll.py 8192 $1 > /tmp/sp.log 2>&1
grep "level 0" /tmp/sp.log
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/sp/sp_kernel_stats.csv')):
    n = r['Name']
    if 'potf2_kernel' in n or 'trsm_kernel<double, false>' in n or 'gemm_kernel<double, true, true, 64>' in n or 'gemm_kernel<double, true, false, 128>' in n:
        print(f"  {n[28:70]:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  max {float(r['MaxNs'])/1e3:8.1f} us")
PY
