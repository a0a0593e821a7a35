"""Timing of hbo_tune settings on cfg 4 (T tasks or the heaviest 8-task shard): scan_mt.py T|shard8 "a=1 b=2" ...  (median of 3 rounds of 5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
from tests.helpers import flatten
data, raw = bench.cfg4_inputs()
full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
ds = parallel.shard_dataset(full, 0, 8) if sys.argv[1] == 'shard8' else {k: full[k] for k in sorted(full)[:int(sys.argv[1])]}
dev = objectives.DeviceDataset(ds)
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
res, ref = {}, None
for rnd in range(3):
    for st in sys.argv[2:]:
        for kv in st.split():
            ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        v, g = f(); f()
        t0 = time.perf_counter()
        for _ in range(5): f()
        res.setdefault(st, []).append(2e2 * (time.perf_counter() - t0))
        g = flatten(g)
        if ref is None: ref = (v, g)
        res[('err', st)] = (abs(v - ref[0]) / abs(ref[0]), np.abs(g - ref[1]).max() / np.abs(ref[1]).max())
for st in sys.argv[2:]:
    print('%s tasks  %-36s %.3f ms  (%s)  vs first setting: nll %.1e grad %.1e' % (sys.argv[1], st, sorted(res[st])[1], ' '.join('%.2f' % v for v in res[st]), *res[('err', st)]))
