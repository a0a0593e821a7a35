"""lookahead on / off for batches of equal small tasks: scan_batch_small.py  (T x n pairs inside)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
ctx = nat.default_context()
rng = np.random.default_rng(0)
d = 4
p = defs.GPParams(model={'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)})
for T, n in ((64, 250), (64, 500), (64, 750), (64, 1000), (64, 1500), (8, 500), (8, 1000), (8, 1500), (24, 300), (2, 1000), (2, 2000)):
    dev = objectives.DeviceDataset({k: defs.SubDataset(rng.uniform(size=(n, d)), rng.normal(size=(n, 1))) for k in range(T)})
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    res = {}
    for rnd in range(3):
        for la in (1, 0):
            ctx.set_option('lookahead', la)
            f(); f()
            t0 = time.perf_counter()
            for _ in range(20): f()
            res.setdefault(la, []).append((time.perf_counter() - t0) / 20 * 1e3)
    print('%2d tasks x %4d points (%2d blocks): lookahead 1: %.3f ms   0: %.3f ms' % (T, n, (n + 127) // 128, sorted(res[1])[1], sorted(res[0])[1]), flush=True)
    dev.close()
ctx.set_option('lookahead', 1)
