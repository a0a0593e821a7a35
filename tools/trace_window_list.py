import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Queue_Id',0) or 0), int(r.get('Grid_Size_X', r.get('Grid_Size',0)) or 0)))
rows.sort()
starts=[i for i,r in enumerate(rows) if 'gram_kernel' in r[2]]
ev=rows[starts[-1]:]
t0=ev[0][0]
def short(n):
    for k in ('potf2','trsm_kernel','gemm_kernel','gram','nll_reduce','wtz','grad','aug','dmu'):
        if k in n: return k+('64' if 'Li64E' in n else ('128' if 'Li128E' in n else ''))
    return n[:20]
lo=float(sys.argv[2]) if len(sys.argv)>2 else 2.0
for r in ev:
    t=(r[0]-t0)/1e3
    if lo*1e3 <= t <= lo*1e3+700:
        print('%9.1f %7.1f  q%d  %-14s grid %d' % (t, (r[1]-r[0])/1e3, r[3], short(r[2]), r[4]))
