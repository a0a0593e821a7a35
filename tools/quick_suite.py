"""One-run timing of the legs a kernel or schedule change can move (median of 3 rounds of 10 evaluations each):
cfg 2 shape at N = 2048 / 4096 / 8192 (fp64 NLL+grad), cfg 4 (64 tasks, heaviest 8-task shard), fp32 factorisation at N = 16384.
usage: quick_suite.py [opt=v ...]     (options go to every leg)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
ctx = nat.default_context()
for opt in sys.argv[1:]:
    k, v = opt.split('='); ctx.set_option(k, int(v))
ctx.profile_enable(0)
def med(f, reps=10, rounds=3):
    f(); f()
    ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        ts.append((time.perf_counter() - t0) / reps)
    return 1e3 * sorted(ts)[len(ts) // 2]
out = []
for n in (2048, 4096, 8192):
    x, y, raw = bench.cfg2_inputs(n=n)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    p = defs.GPParams(model=raw)
    out.append(('nll%d' % n, med(lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC))))
data, raw = bench.cfg4_inputs()
full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
p4 = defs.GPParams(model=raw)
for name, ds in (('cfg4_T64', full), ('cfg4_shard8', parallel.shard_dataset(full, 0, 8))):
    dev = objectives.DeviceDataset(ds)
    out.append((name, med(lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p4, dev, utils.DEFAULT_WARP_FUNC), reps=5)))
rng = np.random.Generator(np.random.PCG64(3))
d, f, n = 32, 64, 16384
isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
model = {'lengthscale': isp(np.ones(f)), 'signal_variance': isp(1.0), 'noise_variance': isp(1e-2),
         'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
         'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
to = lambda t: {k: to(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
x = rng.uniform(size=(n, d)).astype(np.float32); y = (np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))).astype(np.float32)
g = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=to(model), config={'mlp_features': (f,)}), utils.DEFAULT_WARP_FUNC)
def fac():
    g.update_model_params(g.params.model); g.setup_predictor(0)
out.append(('factor32_16384', med(fac, reps=3)))
print('  '.join('%s %.3f' % kv for kv in out))
