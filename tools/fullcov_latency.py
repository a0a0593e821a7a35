"""gp.predict(full_cov=True) latency: the M x M posterior covariance against a cached factorisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
for n, M in ((2000, 256), (8000, 512), (8000, 2048)):
    x, y, raw = bench.cfg2_inputs(n=n)
    xq = np.random.default_rng(0).uniform(size=(M, x.shape[1]))
    m = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.squared_exponential, defs.GPParams(model=raw), utils.DEFAULT_WARP_FUNC)
    m.predict(xq, 0, full_cov=True)
    t0 = time.perf_counter()
    for _ in range(5): mu, cov = m.predict(xq, 0, full_cov=True)
    t1 = time.perf_counter()
    for _ in range(5): m.predict(xq, 0)
    t2 = time.perf_counter()
    print('N = %d M = %d: full_cov %.2f ms, diagonal %.2f ms  (cov %s, min eig %.2e)' % (n, M, (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3, cov.shape, np.linalg.eigvalsh(cov).min()))
from hyperbo_amd import _native as nat
ctx = nat.default_context(); ctx.profile_enable(1)
t0 = time.perf_counter(); m.predict(xq, 0, full_cov=True); t1 = time.perf_counter()
print('one call with stage events: %.2f ms' % ((t1 - t0) * 1e3), {k: round(v[0], 3) for k, v in ctx.profile_get().items()})
