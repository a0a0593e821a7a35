import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
rng = np.random.default_rng(0)
T, n, d, bs = 24, 500, 4, 100
data = {k: defs.SubDataset(rng.uniform(size=(n, d)), rng.normal(size=(n, 1))) for k in range(T)}
full = objectives.DeviceDataset(data)
p = defs.GPParams(model={'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)})
L = nat.lib()
acc = {}
def wrap(name):
    orig = getattr(L, name)
    def f(*a):
        t0 = time.perf_counter(); r = orig(*a); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
    setattr(L, name, f)
for nm in ('hbo_dataset_subsample', 'hbo_objective', 'hbo_dataset_free'): wrap(nm)
def step():
    ix = {k: rng.choice(n, bs, replace=False).astype(np.int32) for k in range(T)}
    b = full.subsample(ix)
    v = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, b, utils.DEFAULT_WARP_FUNC)
    b.close()
for _ in range(20): step()
acc.clear()
t0 = time.perf_counter()
for _ in range(200): step()
el = time.perf_counter() - t0
print('step %.1f us' % (el / 200 * 1e6), {k: round(v / 200 * 1e6, 1) for k, v in acc.items()})
