"""Device-memory use over long runs of the caller-level loops (training with per-step batches, BO steps with appends, acquisition
gradients): free memory before / after, read with hipMemGetInfo."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.bo_utils import acfun
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
hip = C.CDLL('libamdhip64.so')
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
ctx = nat.default_context()
rng = np.random.default_rng(0)
d = 4
data = {k: defs.SubDataset(rng.uniform(size=(500, d)), rng.normal(size=(500, 1))) for k in range(24)}
model = lambda: {'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)}
def train(steps):
    p = defs.GPParams(model=model(), config={'method': 'adam', 'batch_size': 100, 'max_training_step': steps, 'learning_rate': 0.01, 'objective': objectives.nll})
    gp.GP(data, mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC).train(key=1)
train(50)
f0 = free_mb()
t0 = time.perf_counter(); train(3000); t1 = time.perf_counter()
f1 = free_mb()
print('3000 Adam steps with fresh batches: %.1f s, free memory %.0f -> %.0f MB (delta %+.1f)' % (t1 - t0, f0, f1, f1 - f0))
n = 3000
x, y, raw = bench.cfg2_inputs(n=n + 400)
m = gp.GP({0: defs.SubDataset(x[:n], y[:n])}, mean.constant, kernel.squared_exponential, defs.GPParams(model=raw, config={'incremental_cache': True}), utils.DEFAULT_WARP_FUNC)
m.predict(x[:64], 0)
for i in range(20):
    m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True); m.predict(x[:64], 0)
    acfun.expected_improvement.value_and_grad(model=m, sub_dataset_key=0, x_queries=x[:16])
f0 = free_mb()
trace = []
for i in range(20, 320):
    m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True); m.predict(x[:64], 0)
    acfun.expected_improvement.value_and_grad(model=m, sub_dataset_key=0, x_queries=x[:16])
    if i % 20 == 19: trace.append((n + i + 1, round(free_mb())))
f1 = free_mb()
print('free MB by observation count:', trace)
print('300 BO steps (append + posterior + acquisition gradient, two re-factorisations at the 128-row capacity edges): free memory %.0f -> %.0f MB (delta %+.1f)' % (f0, f1, f1 - f0))
