import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, helpers
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
ctx = nat.default_context()
rng = np.random.default_rng(21)
d, n = 5, 4500
model = helpers.make_model(rng, 'constant', False, d)
to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
x, y = helpers.synthetic_task(rng, n, d)
for noise in (1e-2, 1e-3, 1e-4, 1e-5):
  model['noise_variance'] = np.array(helpers.inv_softplus(noise)); model['lengthscale'] = model['lengthscale'] * 0 + helpers.inv_softplus(0.8)
  v64, g64 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model), {0: defs.SubDataset(x, y)}, utils.DEFAULT_WARP_FUNC)
  ds = {0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}
  for opts in ({'bf16x3': 0}, {'bf16x3': 1, 'chol_f16x2': 0}, {'bf16x3': 1, 'chol_f16x2': 1}):
    for k, v in opts.items(): ctx.set_option(k, v)
    try:
      v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=to32(model)), ds, utils.DEFAULT_WARP_FUNC)
      f = helpers.flatten(g)
      print(noise, opts, v64, v, np.max(np.abs(f - helpers.flatten(g64))) / np.max(np.abs(helpers.flatten(g64))), flush=True)
    except Exception as e:
      print(noise, opts, 'EXC', repr(e)[:200], flush=True)
