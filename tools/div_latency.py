"""EKL / Euclid divergence objectives (value + gradient) against the NLL on the same aligned dataset: ms per evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
rng = np.random.default_rng(0)
for n, m in ((500, 24), (2000, 24), (4000, 64)):
    d = 4
    x = rng.uniform(size=(n, d))
    y = rng.normal(size=(n, m))
    ds = {'al': defs.SubDataset(x, y, aligned='g')}
    model = {'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)}
    p = defs.GPParams(model=model)
    dev = objectives.DeviceBatch(ds)
    out = []
    for name, f in (('nll(all)', lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC, exclude_aligned=False)),
                    ('ekl', lambda: objectives.ekl.value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)),
                    ('euc', lambda: objectives.euc.value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC))):
        f(); f()
        t0 = time.perf_counter()
        for _ in range(10): f()
        out.append('%s %.3f ms' % (name, (time.perf_counter() - t0) / 10 * 1e3))
    print('n = %d, %d aligned columns: ' % (n, m) + '   '.join(out))
    dev.close()
