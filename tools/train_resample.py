"""ms per Adam step when every step draws a FRESH batch (24 sub-datasets of 400 points, batch_size 100: the reference's data_utils.py:72-100 regime):
the rows are gathered on the device from the resident dataset, the evaluation is one launch (small.hip); small_fused = 0 for the blocked pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, helpers
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
rng = np.random.default_rng(0); d = 4
ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, 400, d)) for i in range(24)}
ctx = nat.default_context()
for fused in (1, 0, 1):
  ctx.set_option('small_fused', fused)
  cfg = {'method': 'adam', 'batch_size': 100, 'max_training_step': 1000, 'learning_rate': 1e-3}
  pm = defs.GPParams(model=helpers.make_model(np.random.default_rng(1), 'constant', False, d), config=cfg)
  t0 = time.perf_counter()
  out = gp.infer_parameters(mean.constant, kernel.squared_exponential, pm, ds, utils.DEFAULT_WARP_FUNC, objectives.nll, key=0)
  print('fused', fused, '%.3f ms per Adam step (24 tasks of 400 points, fresh batch of 100 per step)' % ((time.perf_counter() - t0) / 1000 * 1e3), float(np.sum(helpers.flatten(out.model))))
