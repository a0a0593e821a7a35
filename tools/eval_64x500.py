import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
rng = np.random.default_rng(0)
T, n, d = 64, 500, 4
data = {k: defs.SubDataset(rng.uniform(size=(n, d)), rng.normal(size=(n, 1))) for k in range(T)}
dev = objectives.DeviceDataset(data)
p = defs.GPParams(model={'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)})
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
from hyperbo_amd import _native as nat
ctx = nat.default_context()
for opt in sys.argv[1:]:
    ctx.set_option(opt.split('=')[0], int(opt.split('=')[1]))
for _ in range(5): f()
t0 = time.perf_counter()
for _ in range(50): f()
print(' '.join(sys.argv[1:]) or 'defaults', 'ms per evaluation', (time.perf_counter() - t0) / 50 * 1e3)
