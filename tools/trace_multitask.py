"""Per-launch kernel list (start, duration, grid) of the last cfg4 evaluation in a rocprofv3 kernel trace.
usage: trace_multitask.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last evaluation = from the last gram_kernel on
last = max(i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = [int(r[k]) // max(1, int(r[k.replace('Grid', 'Workgroup')])) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z')]
    print(f'{s:10.1f} {d:9.1f} us  q{r.get("Queue_Id","?"):>2s} grid {g[0]:5d}x{g[1]:3d}x{g[2]:3d}  {name}')
