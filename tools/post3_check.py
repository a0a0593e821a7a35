"""Accuracy and speed of the fp32 posterior product on the bf16 matrix cores (post3.hip) against the fp32-MFMA product and fp64.
usage: post3_check.py [n] [M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.Generator(np.random.PCG64(5))
d = 8
isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
model = {'lengthscale': isp(np.full(d, 0.7)), 'signal_variance': isp(1.3), 'noise_variance': isp(1e-2), 'constant': np.array(0.2)}
x = rng.uniform(size=(n, d)); y = np.sin(x[:, :3].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))
xq = rng.uniform(size=(M, d))
ctx = nat.default_context()
res = {}
for name, dt, opt in (('f64', np.float64, 0), ('f32 mfma', np.float32, 0), ('f32 bf16x3', np.float32, 1)):
  ctx.set_option('post_bf16x3', opt)
  to = lambda t: {k: to(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=dt)
  g = gp.GP({0: defs.SubDataset(x.astype(dt), y.astype(dt))}, mean.constant, kernel.matern52, defs.GPParams(model=to(model)), utils.DEFAULT_WARP_FUNC)
  g.setup_predictor(0)
  mu, var = g.predict(xq.astype(dt), 0)
  t0 = time.perf_counter(); mu, var = g.predict(xq.astype(dt), 0); t1 = time.perf_counter()
  res[name] = (np.asarray(mu, np.float64).ravel(), np.asarray(var, np.float64).ravel(), (t1 - t0) * 1e3)
mu0, var0, _ = res['f64']
for name in ('f32 mfma', 'f32 bf16x3'):
  mu, var, ms = res[name]
  print(f'{name:12s} n={n} M={M}: predict {ms:8.2f} ms   max|mu - mu64| {np.abs(mu - mu0).max():.3e}   max|var - var64| {np.abs(var - var0).max():.3e}'
        f'   rms var err {np.sqrt(np.mean((var - var0) ** 2)):.3e}   (var range {var0.min():.3e} .. {var0.max():.3e})')
a, b = res['f32 mfma'][1], res['f32 bf16x3'][1]
print('max |var_bf16x3 - var_mfma| %.3e' % np.abs(a - b).max())
