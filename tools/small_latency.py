"""Where a tiny (N=64) NLL+grad evaluation spends its time: Python host mirror vs the C-ABI call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
acc = {'t': 0.0, 'n': 0}
L = nat.lib()
orig = L.hbo_objective
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); acc['t'] += time.perf_counter() - t0; acc['n'] += 1; return r
L.hbo_objective = timed
for _ in range(20): f()
acc['t'] = 0; acc['n'] = 0
t0 = time.perf_counter()
for _ in range(200): f()
t1 = time.perf_counter()
print(f'N={n}: full python call {1e6*(t1-t0)/200:.1f} us; inside hbo_objective {1e6*acc["t"]/max(acc["n"],1):.1f} us x {acc["n"]/200:.1f} calls')
