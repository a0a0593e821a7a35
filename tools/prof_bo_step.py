"""Where one BO step (append one observation to a cached factorisation + posterior at 64 queries) spends its time."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8100
x, y, raw = bench.cfg2_inputs(n=n + 40)
xq = x[:64]
m = gp.GP({0: defs.SubDataset(x[:n], y[:n])}, mean.constant, kernel.squared_exponential,
          defs.GPParams(model=raw, config={'incremental_cache': True}), utils.DEFAULT_WARP_FUNC)
m.predict(xq, 0)
for i in range(4):
    m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True); m.predict(xq, 0)
pr = cProfile.Profile()
ts = []
for i in range(4, 24):
    m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True)
    pr.enable(); t0 = time.perf_counter(); m.predict(xq, 0); ts.append(time.perf_counter() - t0); pr.disable()
print('append + predict: median %.3f ms' % (1e3 * np.median(ts)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
ctx = nat.default_context(); ctx.profile_enable(2)
m.update_sub_dataset((x[n + 30:n + 31], y[n + 30:n + 31]), 0, is_append=True); m.predict(xq, 0)
print({k: (round(v[0], 3), v[1]) for k, v in ctx.profile_get().items()})
