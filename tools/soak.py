"""Determinism soak: the same evaluation repeated many times must return the same bits (a race in a persistent kernel, a
missed event or a stale workspace would show up as a differing value sooner or later).  soak.py [seconds per case] [poison]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
from tests.helpers import flatten
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
if 'poison' in sys.argv[2:]:   # every evaluation recomputes into NaN-filled buffers: skipped work cannot repeat the previous bits
    nat.default_context().set_option('poison', 1)
def run(name, f):
    v0, g0 = f(); g0 = flatten(g0)
    t0 = time.perf_counter(); k = 0; bad = 0
    while time.perf_counter() - t0 < budget:
        v, g = f(); k += 1
        if v != v0 or not np.array_equal(flatten(g), g0): bad += 1
    print('%-44s %5d evaluations, %d differing' % (name, k, bad), flush=True)
    return bad
total = 0
to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
for n in (8192, 2900, 4500):
    x, y, raw = bench.cfg2_inputs(n=n)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    p = defs.GPParams(model=raw)
    total += run('fp64 N = %d' % n, lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC))
    dev.close()
    d32 = objectives.DeviceDataset({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))})
    p32 = defs.GPParams(model=to32(raw))
    total += run('fp32 N = %d' % n, lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p32, d32, utils.DEFAULT_WARP_FUNC))
    d32.close()
data, raw4 = bench.cfg4_inputs()
for T in (64, 8, 3):
    dev = objectives.DeviceDataset({k: defs.SubDataset(*data[k]) for k in sorted(data)[:T]})
    p = defs.GPParams(model=raw4)
    total += run('fp64 ragged batch of %d' % T, lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC))
    dev.close()
print('TOTAL differing:', total)
