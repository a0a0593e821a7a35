"""A/B of one hbo_tune knob on the cfg-2 shape (values and gradients compared to the bit):  ab_knob.py knob [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
from tests.helpers import flatten
knob = sys.argv[1]
sizes = [int(a) for a in sys.argv[2:]] or [1000, 2048, 4096, 8192]
ctx = nat.default_context()
for n in sizes:
    x, y, raw = bench.cfg2_inputs(n=n)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    p = defs.GPParams(model=raw)
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    res = {}
    for rnd in range(3):
        for v in (0, 1):
            ctx.set_option(knob, v)
            val, g = f(); f()
            t0 = time.perf_counter()
            for _ in range(10): f()
            t1 = time.perf_counter()
            res.setdefault(v, []).append(1e2 * (t1 - t0)); res[('r', v)] = (val, flatten(g))
    a, b = res[('r', 0)], res[('r', 1)]
    print('N = %5d  %s = 0: %.3f ms   = 1: %.3f ms   identical: %s  (value diff %.3g, grad maxdiff %.3g)' % (
        n, knob, sorted(res[0])[1], sorted(res[1])[1], a[0] == b[0] and (a[1] == b[1]).all(), abs(a[0] - b[0]), np.abs(a[1] - b[1]).max()), flush=True)
ctx.set_option(knob, 0)
