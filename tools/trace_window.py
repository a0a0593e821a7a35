"""All kernels (every queue) around the largest panel-to-panel gap of the last evaluation in a rocprofv3 kernel trace."""
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
potf2 = [r for r in ev if 'potf2' in r[2]]
gaps = [(b[0] - a[0], i) for i, (a, b) in enumerate(zip(potf2, potf2[1:]))]
g, i = max(gaps)
if len(sys.argv) > 2: i = int(sys.argv[2]); g = gaps[i][0]   # explicit panel instead of the largest gap
SPAN = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lo = potf2[max(i - 1, 0)][0]; hi = potf2[min(i + SPAN, len(potf2) - 1)][1]
def short(n):
    m = re.search(r'(potf2|trsm_kernel|gemm_kernel|syrk3|split3_panel|split3_block|gram|wtz|nll_reduce)', n)
    s = m.group(1) if m else n[:24]
    if 'gemm_kernel' in n:
        s += '<' + ('kk' if 'Lb1ELb1' in n else ('km' if 'Lb1ELb0' in n else 'mm')) + (',64>' if 'Li64E' in n else ',128>')
    return s
print('largest period %.1f us after panel %d' % (g / 1e3, i))
for r in ev:
    if r[1] >= lo and r[0] <= hi:
        print('q%d %9.1f -> %9.1f (%7.1f us) %-22s wgs=%dx%d' % (r[3], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, short(r[2]), r[4], r[5]))
