export TMPDIR=/tmp
ROOT=$PWD
for n in 64 256 512 1000; do python tools/small_latency.py $n 2>&1 | tail -1; done
cd /tmp && rm -rf /tmp/lat256
rocprofv3 --kernel-trace --output-format csv -d /tmp/lat256 -- python $ROOT/tools/small_latency.py 256 > /tmp/lat256.log 2>&1
f=$(find /tmp/lat256 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last evaluation: from the last aug_rows kernel
idx=[i for i,r in enumerate(rows) if 'aug_rows' in r[2]]
ev=rows[idx[-1]:]
t0=ev[0][0]
busy=sum(e-s for s,e,_ in ev)
print('kernels in one evaluation:', len(ev), 'span us %.1f'%((ev[-1][1]-t0)/1e3), 'sum of kernel time us %.1f'%(busy/1e3))
for s,e,n in ev: print('  %7.1f %6.1f %s'%((s-t0)/1e3,(e-s)/1e3,n[:70]))
P
