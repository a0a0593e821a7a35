import os, sys, time
sys.path.insert(0, '/root/repo')
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
x, y, raw = bench.cfg2_inputs(n=8192)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
for lvl in (0, 1, 0, 1):
    ctx.profile_enable(lvl)
    f(); f()
    t0 = time.perf_counter()
    for _ in range(10): f(); ctx.profile_get()
    t1 = time.perf_counter()
    print(f'level {lvl}: {1e2*(t1-t0):.2f} ms per eval')
