import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
from hyperbo_amd import _native as nat
rng = np.random.default_rng(0)
tasks, n, d, bs = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 2000, 4, 500)))
data = {}
for k in range(tasks):
    x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
    data[k] = defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1)))
model = lambda: {'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)}
METHOD = os.environ.get('HBO_TRAIN_METHOD', 'adam')
p = defs.GPParams(model=model(), config={'method': METHOD, 'batch_size': bs, 'max_training_step': 40 if METHOD == 'adam' else 8, 'learning_rate': 0.01, 'objective': objectives.nll})
g = gp.GP(data, mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC)
g.train(key=1)
g.params.model = model()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); g.train(key=2); el = time.perf_counter() - t0
pr.disable()
print('ms per step', el / (40 if METHOD == 'adam' else 8) * 1e3)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
ctx = nat.default_context(); ctx.profile_enable(1)
g.params.model = model(); g.train(key=3)
print({k: (round(v[0], 3), v[1]) for k, v in ctx.profile_get().items()})
