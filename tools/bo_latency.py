"""Latency of the per-iteration BO calls on a small observed set: factor, predict, acquisition value / gradient."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
from hyperbo_amd.bo_utils import acfun
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
D = 8
rng = np.random.default_rng(0)
x = rng.uniform(size=(n, D)); y = np.sin(x.sum(1, keepdims=True))
xq = rng.uniform(size=(M, D))
params = defs.GPParams(model={'constant': 0.1, 'lengthscale': np.full(D, 0.5), 'signal_variance': 0.3, 'noise_variance': -4.0})
model = gp.GP(dataset={'t': defs.SubDataset(x, y)}, mean_func=mean.constant, cov_func=kernel.squared_exponential, params=params, warp_func=utils.DEFAULT_WARP_FUNC)
def timeit(name, fn, reps=100):
  for _ in range(5): fn()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  print(f'{name:46s} {1e6*(time.perf_counter()-t0)/reps:9.1f} us')
def refactor():
  model.params.cache.clear() if hasattr(model.params, 'cache') and model.params.cache else None
  model.setup_predictor('t')
timeit(f'setup_predictor (factor) n={n}', refactor)
timeit(f'predict M={M}', lambda: model.predict(xq, 't'))
timeit(f'expected_improvement M={M}', lambda: acfun.expected_improvement(model=model, sub_dataset_key='t', x_queries=xq))
timeit(f'ucb M={M}', lambda: acfun.ucb(model=model, sub_dataset_key='t', x_queries=xq))
xs = xq[:16]
timeit('expected_improvement.value_and_grad M=16', lambda: acfun.expected_improvement.value_and_grad(model=model, sub_dataset_key='t', x_queries=xs))
timeit('append 1 observation + EI', lambda: (model.update_sub_dataset(defs.SubDataset(xq[:1], np.zeros((1, 1))), 't', is_append=True), acfun.expected_improvement(model=model, sub_dataset_key='t', x_queries=xq)), reps=20)
