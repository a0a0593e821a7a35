"""rocprofv3 counter_collection csv files (FETCH_SIZE pass, WRITE_SIZE pass) -> per-kernel averages per launch (KB),
the format of profiles/r0N_pmc_hbm.json that bench.py reads for roofline.traffic.
usage: python tools/pmc_to_json.py <fetch.csv> <write.csv> > profiles/r01_pmc_hbm.json"""
import collections, csv, json, re, sys
out = collections.OrderedDict()
for path in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> dispatch -> value
    cname = None
    for r in csv.DictReader(open(path)):
        cname = r['Counter_Name']
        name = re.sub(r'^void \(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name)
        per[name][r['Dispatch_Id']] += float(r['Counter_Value'])
    for name, d in per.items():
        e = out.setdefault(name, collections.OrderedDict())
        e[cname + '_KB'] = sum(d.values()) / len(d)
        e['launches'] = len(d)
json.dump(out, sys.stdout, indent=1)
print()
