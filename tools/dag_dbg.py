import faulthandler, os, sys, time
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = nat.default_context()
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
print('launch path', flush=True)
v0, g0 = f(); print(v0, flush=True)
ctx.set_option('dag', 1)
for a in sys.argv[2:]:
    ctx.set_option(a.split('=')[0], int(a.split('=')[1]))
print('dag path', flush=True)
t0 = time.perf_counter(); v1, g1 = f(); print(v1, time.perf_counter() - t0, flush=True)
t0 = time.perf_counter(); v1, g1 = f(); print(v1, time.perf_counter() - t0, flush=True)
