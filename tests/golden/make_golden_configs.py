"""Golden fixtures for the BASELINE.json configurations whose oracle run is too slow for the test tiers.

  cfg1_se_n256_d4.npz    SURVEY.md 8(d) cfg 1: seed 1, X~U[0,1]^{256x4}, y = sin(2 pi X w) + 0.1 eps, SE kernel with
                         lengthscale 0.5, signal variance 1, noise 1e-2, constant mean 0, fp64 -- NLL (+ gradient, factor
                         diagonal, K^-1 y) from oracle/hyperbo_oracle.py, cross-checked here against SciPy LAPACK.
  cfg4_t64_oracle.npz    cfg 4: the 64 sub-datasets of bench.cfg4_inputs() (seed 4, N_k in [1600, 2400], D = 4), SE-ARD +
                         constant mean: per-task NLL, mean NLL and mean gradient from the oracle (about 5 minutes of
                         NumPy/LAPACK on 8 cores -- the GPU test compares all 64 tasks with this file and re-runs the
                         oracle live on a sample of them).

Generated from the oracle, NOT from the JAX reference (jax is not installable here);
tests/golden/make_golden_from_reference.py is the one-shot script that pins these against the reference itself.

  python tests/golden/make_golden_configs.py [cfg1] [cfg4]
"""
import os
import sys

import numpy as np
import scipy.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import hyperbo_oracle as o  # noqa: E402
import helpers  # noqa: E402


def cfg1_inputs(seed=1, n=256, d=4):
  """SURVEY.md 8(d) cfg 1 (NumPy Generator(PCG64(seed)); draw order: X, w, eps)."""
  rng = np.random.Generator(np.random.PCG64(seed))
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
  raw = {'lengthscale': helpers.inv_softplus(np.full(d, 0.5)), 'signal_variance': helpers.inv_softplus(1.0),
         'noise_variance': helpers.inv_softplus(1e-2), 'constant': np.array(0.0)}
  return x, y, raw


def make_cfg1():
  x, y, raw = cfg1_inputs()
  p = o.GPParams(model=raw)
  wf = o.DEFAULT_WARP_FUNC
  nll = o.neg_log_marginal_likelihood(o.constant, o.squared_exponential, p, {0: o.SubDataset(x, y)}, wf)
  nll_svd = o.neg_log_marginal_likelihood(o.constant, o.squared_exponential, p, {0: o.SubDataset(x, y)}, wf, use_cholesky=False)
  val, grads = o.nll_value_and_grad(o.constant, o.squared_exponential, p, {0: o.SubDataset(x, y)}, wf)
  chol, kinvy, ymu = o.solve_gp_linear_system(o.constant, o.squared_exponential, p, x, y, wf)
  # independent LAPACK recomputation from first principles (no oracle code): the pin of this fixture
  ls, sv, noise = 0.5 + 1e-10, 1.0 + 1e-10, 1e-2 + 1e-10    # softplus(inv_softplus(v)) + 1e-10
  d2 = ((x[:, None, :] - x[None, :, :])**2).sum(-1) / ls**2
  k = sv * np.exp(-0.5 * d2) + (noise + 1e-6) * np.eye(len(x))
  c = spla.cholesky(k, lower=True)
  alpha = spla.cho_solve((c, True), y)
  nll_lapack = float(0.5 * (y.T @ alpha)[0, 0] + np.sum(np.log(np.diag(c))) + 0.5 * len(x) * np.log(2 * np.pi))
  assert abs(nll - nll_lapack) <= 1e-11 * abs(nll_lapack), (nll, nll_lapack)
  assert abs(val - nll) <= 1e-13 * abs(nll)
  np.savez_compressed(os.path.join(HERE, 'cfg1_se_n256_d4.npz'), x=x, y=y, model_flat=helpers.flatten(raw),
                      nll=nll, nll_svd=nll_svd, nll_lapack=nll_lapack, grad_flat=helpers.flatten(grads),
                      chol_diag=np.diag(chol), kinvy=kinvy)
  print('cfg1 nll', nll, 'lapack', nll_lapack, 'svd', nll_svd)


def make_cfg4():
  import bench
  data, raw = bench.cfg4_inputs()
  p = o.GPParams(model=raw)
  wf = o.DEFAULT_WARP_FUNC
  per_task, sizes, gsum = [], [], None
  for k in sorted(data):
    xk, yk = data[k]
    v, g = o.nll_sub_dataset_value_and_grad(o.constant, o.squared_exponential, p, xk, yk, wf)
    per_task.append(v); sizes.append(len(xk))
    gf = helpers.flatten(g)
    gsum = gf if gsum is None else gsum + gf
    print('task', k, len(xk), v, flush=True)
  per_task = np.asarray(per_task)
  np.savez_compressed(os.path.join(HERE, 'cfg4_t64_oracle.npz'), sizes=np.asarray(sizes), nll_per_task=per_task,
                      nll_mean=per_task.mean(), grad_mean_flat=gsum / len(per_task), model_flat=helpers.flatten(raw))
  print('cfg4 mean nll', per_task.mean())


if __name__ == '__main__':
  which = sys.argv[1:] or ['cfg1', 'cfg4']
  if 'cfg1' in which:
    make_cfg1()
  if 'cfg4' in which:
    make_cfg4()
