"""Largest single factorisation that fits beside nothing else: N = 131072 fp64 (137 GB Gram in place).  Closed-form check with the
dot-product kernel on 1-D inputs (matrix determinant lemma / Woodbury, as tests/test_gpu_parity.py::test_cfg5_full_size_closed_form),
then the SE-ARD timing of the cfg-5 shape.  usage: big_n.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
ctx = nat.default_context()
rng = np.random.default_rng(5)
x = rng.uniform(-1, 1, size=(n, 1)); y = rng.normal(size=(n, 1))
sigma, bias, noise = 0.7, 0.3, 0.1
model = {'dot_prod_sigma': np.array(sigma), 'dot_prod_bias': np.array(bias), 'noise_variance': np.array(noise)}
t0 = time.perf_counter()
v = objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, defs.GPParams(model=model), {0: defs.SubDataset(x, y)})
t1 = time.perf_counter()
c = noise + 1e-6
U = np.hstack([x / sigma, np.full((n, 1), bias)])
cap = np.eye(2) + U.T @ U / c
logdet = n * np.log(c) + np.linalg.slogdet(cap)[1]
uty = U.T @ y
quad = ((y.T @ y).item() - (uty.T @ np.linalg.solve(cap, uty)).item() / c) / c
expect = 0.5 * quad + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi)
print(f'N={n}: dot-product NLL {v:.9f} vs closed form {expect:.9f}: rel err {abs(v - expect) / abs(expect):.2e}  ({t1 - t0:.2f} s incl. upload)', flush=True)
x, y, raw = bench.cfg2_inputs(seed=5, n=n)
raw['noise_variance'] = bench.inv_softplus(1e-1)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx.profile_enable(1)
p = defs.GPParams(model=raw)
t0 = time.perf_counter(); v = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC); t1 = time.perf_counter()
prof = ctx.profile_get()
print(f'N={n}: SE-ARD NLL {v:.6f} in {t1 - t0:.2f} s; potrf {prof["potrf"][0] / 1e3:.2f} s = {n**3 / 3 / (prof["potrf"][0] * 1e-3) / 1e12:.1f} TFLOP/s; Gram {prof["gram"][0]:.1f} ms = {8 * n * (n + 1) / 2 / (prof["gram"][0] * 1e-3) / 1e12:.2f} TB/s written')
