"""Where an HGP acquisition's time goes: hgp_time.py [S N M]   (bench.py's `hgp` leg: 32 512 64)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.bo_utils import acfun
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
S, n, M = (int(a) for a in (sys.argv[1:4] + ['32', '512', '64'][len(sys.argv) - 1:]))
d = 8
rng = np.random.default_rng(7)
x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
xq = rng.uniform(size=(M, d))
samples = [{'lengthscale': bench.inv_softplus(np.full(d, 0.5)) + 0.2 * rng.normal(size=d), 'signal_variance': bench.inv_softplus(1.0) + 0.1 * rng.normal(),
            'noise_variance': bench.inv_softplus(1e-2) + 0.1 * rng.normal(), 'constant': np.array(0.1 * rng.normal())} for _ in range(S)]
data = {0: defs.SubDataset(x, y)}
hgp = gp.HGP(data, mean.constant, kernel.squared_exponential, defs.GPParams(model=samples[0], samples=samples), utils.DEFAULT_WARP_FUNC)
ctx = nat.default_context()
f = lambda: acfun.expected_improvement(model=hgp, sub_dataset_key=0, x_queries=xq)
def med(g, reps=10):
    g(); g()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); g(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
print('S=%d N=%d M=%d' % (S, n, M))
print('batched, model structs cached  %.3f ms' % med(f))
def fresh():
    hgp._hbo_sample_models = None
    return f()
print('batched, structs rebuilt       %.3f ms' % med(fresh))
ctx.profile_enable(1)
f()
for k, (ms, cnt) in ctx.profile_get().items():
    print('   %-14s %8.3f ms %4d launches' % (k, ms, cnt))
ctx.profile_enable(0)
def loop():
    vals = []
    for smp in samples:
        g = gp.GP(data, mean.constant, kernel.squared_exponential, defs.GPParams(model=smp), utils.DEFAULT_WARP_FUNC)
        vals.append(acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq))
    return np.mean(vals, axis=0)
print('loop over samples              %.3f ms' % med(loop, 5))
