"""A/B of context options on the headline evaluation: ab_opt.py N opt=a,opt2=b  opt=c ...  (each argument one setting)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1])
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
ctx.profile_enable(0)
for rnd in range(2):
    for opts in sys.argv[2:]:
        for opt in opts.split(','):
            k, v = opt.split('='); ctx.set_option(k, int(v))
        f(); f()
        t0 = time.perf_counter()
        for _ in range(10): f()
        t1 = time.perf_counter()
        print(f'{opts:40s} {1e2*(t1-t0):7.3f} ms per eval')
