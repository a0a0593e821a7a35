"""Per-queue kernel list (start, duration, grid) of the last evaluation's first N ms from a rocprofv3 kernel trace."""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Queue_Id']),
                     int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y'])))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'gram_kernel' in r[2]]
ev = rows[starts[-1]:]
t0 = ev[0][0]
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
def short(n):
    m = re.search(r'(potf2|trsm_kernel|gemm_kernel|gram|wtz|nll_reduce|aug_rows)', n)
    s = m.group(1) if m else n[:24]
    if 'gemm_kernel' in n:
        s += '<' + ('kk' if 'Lb1ELb1' in n else ('km' if 'Lb1ELb0' in n else 'mm')) + (',64>' if 'Li64E' in n else ',128>')
    return s
for r in ev:
    if (r[0] - t0) / 1e6 > lim: break
    print('q%d %8.1f +%7.1f us  %-20s %dx%d' % (r[3], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, short(r[2]), r[4], r[5]))
