"""Sum PMC counters per kernel name for the largest dispatch of each kernel from a rocprofv3 counter_collection csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
by = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if pat and pat not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:70], r['Dispatch_Id'], r['Grid_Size'])
    by[key][r['Counter_Name']] += float(r['Counter_Value'])
# largest grid per kernel name
best = {}
for (name, did, grid), c in by.items():
    if name not in best or int(grid) > int(best[name][1]):
        best[name] = (did, grid, c)
for name, (did, grid, c) in best.items():
    print(name, 'dispatch', did, 'grid', grid)
    for k, v in sorted(c.items()):
        print('   %-28s %.4e' % (k, v))
