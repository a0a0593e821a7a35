"""Resident tile-task schedule (option dag=1) against the launch schedule (dag=0) on the cfg-2-shaped evaluation:
values, gradients and ms per evaluation.  dag_check.py N [N ...] [opt=v ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
sizes = [int(a) for a in sys.argv[1:] if '=' not in a] or [1024, 4096, 8192]
opts = [(a.split('=')[0], int(a.split('=')[1])) for a in sys.argv[1:] if '=' in a]
ctx = nat.default_context()
ctx.profile_enable(0)
for n in sizes:
    x, y, raw = bench.cfg2_inputs(n=n)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    p = defs.GPParams(model=raw)
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    out = {}
    for dag in (0, 1):
        ctx.set_option('dag', dag)
        if dag:
            for k, v in opts:
                ctx.set_option(k, v)
        t0 = time.perf_counter(); v, g = f(); first = time.perf_counter() - t0
        f()
        t0 = time.perf_counter()
        for _ in range(8): v, g = f()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        flat = np.concatenate([np.ravel(np.asarray(g[k], dtype=np.float64)) for k in sorted(g)])
        out[dag] = (float(v), flat, ms, first)
    v0, g0, ms0, _ = out[0]; v1, g1, ms1, first1 = out[1]
    print('N=%d  launch %.3f ms  dag %.3f ms (first call %.1f ms)  nll %.12e vs %.12e rel %.2e  grad rel %.2e' % (
        n, ms0, ms1, first1 * 1e3, v0, v1, abs(v0 - v1) / abs(v0), np.max(np.abs(g0 - g1)) / np.max(np.abs(g0))), flush=True)
