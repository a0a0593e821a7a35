"""Gram stage time (HIP events) of the cfg2 evaluation; results are wrong in the HBO_GRAM_* experiment builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(0)
x = rng.uniform(size=(n, d))
ctx = nat.default_context()
import ctypes as C, time
p = defs.GPParams(model={'lengthscale': np.full(d, 0.3), 'signal_variance': 0.5})
# device-resident Gram of the objective path: time hbo_objective's gram stage
import bench
from hyperbo_amd.gp_utils import mean, objectives, utils
xx, yy, raw = bench.cfg2_inputs(n=n, d=d)
dev = objectives.DeviceDataset({0: defs.SubDataset(xx, yy)})
ctx.profile_enable(1)
pp = defs.GPParams(model=raw)
for _ in range(3):
    try:
        objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pp, dev, utils.DEFAULT_WARP_FUNC)
    except Exception as e:
        pass
    pr = ctx.profile_get()
print(f'N={n} D={d}: gram {pr["gram"][0]*1e3:.1f} us  -> {(n*(n+128)/2*8)/(pr["gram"][0]*1e-3)/1e12:.2f} TB/s written')
