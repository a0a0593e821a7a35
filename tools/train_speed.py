"""Steps per second of GP.train() (Adam with per-step sub-sampling, gp.py:53-195): the caller that turns NLL+grad
evaluations into pre-training time.  usage: python tools/train_speed.py [tasks] [n] [batch_size] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rng = np.random.default_rng(0)
d = 4
data = {}
for k in range(T):
    x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
    data[k] = defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1)))
def model():
    return {'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)}
for method, st in (('adam', steps), ('lbfgs', max(steps // 6, 5))):
    p = defs.GPParams(model=model(), config={'method': method, 'batch_size': bs, 'max_training_step': st, 'learning_rate': 0.01,
                                             'objective': objectives.nll})
    g = gp.GP(data, mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC)
    g.train(key=1)     # warm-up (allocations, first launches)
    g.params.model = model()
    t0 = time.perf_counter(); g.train(key=2); el = time.perf_counter() - t0
    print(f'{method}: {T} tasks x {n} points, batch_size {bs}: {st} steps in {el*1e3:.1f} ms = {el/st*1e3:.2f} ms/step, final nll {objectives.nll(mean.constant, kernel.squared_exponential, g.params, data, utils.DEFAULT_WARP_FUNC):.4f}')
