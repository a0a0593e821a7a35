export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rm -rf /tmp/trbo
rocprofv3 --kernel-trace --output-format csv -d /tmp/trbo -- python $ROOT/tools/prof_bo_step.py 8100 > /tmp/trbo.log 2>&1
f=$(find /tmp/trbo -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),int(r['Grid_Size_Y'])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx=[i for i,r in enumerate(rows) if 'append_row' in r[2]]
ev=rows[idx[-3]-8:idx[-2]-8]
t0=ev[0][0]
for s,e,n,gx,gy in ev: print('  %7.1f %6.1f %4dx%3d %s'%((s-t0)/1e3,(e-s)/1e3,gx,gy,n[:80]))
P
