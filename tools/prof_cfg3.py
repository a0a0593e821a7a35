"""cfg3: Matern-5/2 on tanh-MLP(32->64) + linear_mlp mean, N=16384, fp32: factor once + EI over 65536 candidates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.bo_utils import acfun
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
dt = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == 'f64') else np.float32
opts = [a.split('=') for a in sys.argv[1:] if '=' in a]   # context options: name=value
rng = np.random.Generator(np.random.PCG64(3))
d, f, n, M = 32, 64, 16384, 65536
isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
model = {'lengthscale': isp(np.ones(f)), 'signal_variance': isp(1.0), 'noise_variance': isp(1e-2),
         'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
         'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
to = lambda t: {k: to(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=dt)
x = rng.uniform(size=(n, d)).astype(dt); y = (np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))).astype(dt)
xq = rng.uniform(size=(M, d)).astype(dt)
g = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=to(model), config={'mlp_features': (f,)}), utils.DEFAULT_WARP_FUNC)
ctx = nat.default_context(); ctx.profile_enable(1)
for k_, v_ in opts: ctx.set_option(k_, int(v_))
for it in range(3):
    g.update_model_params(g.params.model)   # drop the cache -> refactor
    t0 = time.perf_counter(); g.setup_predictor(0); t1 = time.perf_counter()
    pf = ctx.profile_get()
    ei = acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq); t2 = time.perf_counter()
    pe = ctx.profile_get()
print(dt.__name__, 'factor (incl. chol export over PCIe) %.1f ms' % ((t1 - t0) * 1e3), {k: round(v[0], 2) for k, v in pf.items()})
print('EI over %d candidates %.1f ms' % (M, (t2 - t1) * 1e3), {k: round(v[0], 2) for k, v in pe.items()}, 'finite', bool(np.isfinite(ei).all()))
print('post_gemm TFLOP/s: %.1f' % (n * n * M / (pe['post_gemm'][0] * 1e-3) / 1e12))
