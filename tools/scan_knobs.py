"""Timing of settings of hbo_tune knobs on the cfg-2 shape: scan_knobs.py N "a=1 b=2" "a=3" ...  (median of 3 rounds of 10)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1])
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
res = {}
for rnd in range(3):
    for st in sys.argv[2:]:
        for kv in st.split():
            ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        f(); f()
        t0 = time.perf_counter()
        for _ in range(10): f()
        res.setdefault(st, []).append(1e2 * (time.perf_counter() - t0))
for st in sys.argv[2:]:
    print('N = %d  %-40s %.3f ms  (%s)' % (n, st, sorted(res[st])[1], ' '.join('%.2f' % v for v in res[st])))
