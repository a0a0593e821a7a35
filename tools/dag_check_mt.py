"""Resident tile-task schedule on the cfg-4-shaped batch: dag_check_mt.py T [opt=v ...] -- ms per evaluation and the value
against the launch schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
T = int(sys.argv[1])
opts = [(a.split('=')[0], int(a.split('=')[1])) for a in sys.argv[2:]]
data, raw = bench.cfg4_inputs(tasks=T)
dev = objectives.DeviceDataset({k: defs.SubDataset(x, y) for k, (x, y) in data.items()})
ctx = nat.default_context()
ctx.profile_enable(0)
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
res = {}
for leg in (0, 1):
    ctx.set_option('dag', 0)
    if leg:
        for k, v in opts: ctx.set_option(k, v)
    v, g = f(); f()
    t0 = time.perf_counter()
    for _ in range(6): v, g = f()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    flat = np.concatenate([np.ravel(np.asarray(g[k], dtype=np.float64)) for k in sorted(g)])
    res[leg] = (float(v), flat, ms)
print('T=%d %s: launch %.3f ms  dag %.3f ms  nll rel %.2e  grad rel %.2e' % (T, ' '.join(sys.argv[2:]), res[0][2], res[1][2],
      abs(res[0][0] - res[1][0]) / abs(res[0][0]), np.max(np.abs(res[0][1] - res[1][1])) / np.max(np.abs(res[0][1]))), flush=True)
