"""O(N^2) GPCache row append vs re-factorisation (the BO-loop step, bayesopt.py:186-190)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8100
x, y, raw = bench.cfg2_inputs(n=n + 8)
xq = x[:64]
for inc in (True, False):
    m = gp.GP({0: defs.SubDataset(x[:n], y[:n])}, mean.constant, kernel.squared_exponential,
              defs.GPParams(model=raw, config={'incremental_cache': inc}), utils.DEFAULT_WARP_FUNC)
    m.predict(xq, 0)
    ts = []
    for i in range(4):
        m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True)
        t0 = time.perf_counter(); mu, var = m.predict(xq, 0); ts.append(time.perf_counter() - t0)
    print(f'N={n} incremental_cache={inc}: append+predict {1e3*np.median(ts):.2f} ms  (mu[0]={mu[0,0]:.10f})')
