"""The reference's training regime: T tasks x n points (default 24 x 100, D = 4), SE-ARD + constant mean, fp64.
Prints ms per Adam step of gp.infer_parameters (batch_size > n: the dataset stays resident, no re-sampling), the time inside
hbo_objective per evaluation, and the same with the single-workgroup evaluation switched off (hbo_tune small_fused = 0)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import helpers
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
d = 4
rng = np.random.default_rng(0)
KN = os.environ.get('KERNEL', 'squared_exponential'); MN = os.environ.get('MEAN', 'constant')
model = helpers.make_model(rng, MN, KN.endswith('_mlp'), d)
ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i in range(T)}
ctx = nat.default_context()
L = nat.lib()
acc = {'t': 0.0, 'n': 0}
orig = L.hbo_objective
def timed(*a):
  t0 = time.perf_counter(); r = orig(*a); acc['t'] += time.perf_counter() - t0; acc['n'] += 1; return r
L.hbo_objective = timed
for fused in (1, 0, 1):
  ctx.set_option('small_fused', fused)
  cfg = {'method': 'adam', 'batch_size': n + 1, 'max_training_step': steps, 'learning_rate': 1e-3, 'mlp_features': helpers.MLP_FEATURES}
  p = defs.GPParams(model=copy.deepcopy(model), config=cfg)
  gp.infer_parameters(getattr(mean, MN), getattr(kernel, KN), p, ds, utils.DEFAULT_WARP_FUNC, objectives.nll, key=0)   # warm
  acc['t'] = 0; acc['n'] = 0
  p = defs.GPParams(model=copy.deepcopy(model), config=cfg)
  t0 = time.perf_counter()
  gp.infer_parameters(getattr(mean, MN), getattr(kernel, KN), p, ds, utils.DEFAULT_WARP_FUNC, objectives.nll, key=0)
  t1 = time.perf_counter()
  print(f'{KN} / {MN}: {T} tasks x {n} points, small_fused={fused}: {1e3 * (t1 - t0) / steps:.4f} ms per Adam step; inside hbo_objective '
        f'{1e6 * acc["t"] / max(acc["n"], 1):.1f} us x {acc["n"] / steps:.2f} calls per step')
