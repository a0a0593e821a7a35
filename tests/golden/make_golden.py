"""Generates tests/golden/*.npz from oracle/hyperbo_oracle.py (NOT from the JAX reference: jax is
not installable here -- SURVEY.md F0.2 -- so these fixtures pin the oracle against accidental
change and give the GPU parity tests fixed inputs).  Inputs come from
numpy.random.Generator(PCG64(seed)); every array needed to recompute the outputs is stored.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import hyperbo_oracle as o  # noqa: E402
import helpers  # noqa: E402

CASES = [
    # name, kernel, mlp_kernel, mean, n, d, nq, seed
    ('se_const_n64_d4', 'squared_exponential', False, 'constant', 64, 4, 16, 11),
    ('m32_linear_n8_d1', 'matern32', False, 'linear', 8, 1, 5, 12),
    ('m52_mlp_linmlp_n256_d16', 'matern52', True, 'linear_mlp', 256, 16, 32, 13),
    ('dot_zero_n64_d4', 'dot_product', False, 'zero', 64, 4, 16, 14),
    ('se_mlp_const_n64_d4', 'squared_exponential', True, 'constant', 64, 4, 16, 15),
]


def build(case):
  name, kname, mlp, mname, n, d, nq, seed = case
  rng = np.random.Generator(np.random.PCG64(seed))
  model = helpers.make_model(rng, mname, mlp, d)
  x, y = helpers.synthetic_task(rng, n, d)
  x2, y2 = helpers.synthetic_task(rng, max(n // 2, 3), d)
  xq = rng.uniform(size=(nq, d))
  params = o.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES})
  kern = getattr(o, kname + ('_mlp' if mlp else ''))
  mean = getattr(o, mname)
  wf = o.DEFAULT_WARP_FUNC
  dataset = {0: o.SubDataset(x, y), 1: o.SubDataset(x2, y2)}
  nll, key2nll = o.neg_log_marginal_likelihood(mean, kern, params, dataset, wf, return_key2nll=True)
  nll_svd = o.neg_log_marginal_likelihood(mean, kern, params, dataset, wf, use_cholesky=False)
  val, grads = o.nll_value_and_grad(mean, kern, params, dataset, wf)
  chol, kinvy, ymu = o.solve_gp_linear_system(mean, kern, params, x, y, wf)
  mu, var = o.predict(mean, kern, params, x, y, xq, wf)
  _, cov = o.predict(mean, kern, params, x, y, xq, wf, full_cov=True)
  mu_n, var_n = o.gp_predict_postprocess(params, dataset, mu, var, wf, False, True, True)
  target = float(np.max(y))
  # divergence objectives on an aligned sub-dataset (objectives.py:29-106) and d acquisition / d x (bayesopt.py:116-125)
  ya = np.sin(3 * x[:, :1]) + 0.3 * rng.normal(size=(n, 6))
  aligned = {'al': o.SubDataset(x, ya, aligned='al'), 0: o.SubDataset(x, y)}
  ekl, ekl_g = o.divergence_value_and_grad('ekl', mean, kern, params, aligned, wf)
  euc, euc_g = o.divergence_value_and_grad('euc', mean, kern, params, aligned, wf)
  noise = float(np.squeeze(o.retrieve_params(params, ['noise_variance'], wf)[0]))
  ei_v, ei_g = o.acquisition_value_and_grad('ei', mean, kern, params, x, y, xq, target, wf, add_noise=noise, scale=2.0)
  ucb_v, ucb_g = o.acquisition_value_and_grad('ucb', mean, kern, params, x, y, xq, 3.0, wf, add_noise=noise, scale=2.0)
  out = dict(
      y_aligned=ya, ekl=ekl, ekl_grad_flat=helpers.flatten(ekl_g), euc=euc, euc_grad_flat=helpers.flatten(euc_g),
      ei_value=ei_v, ei_dx=ei_g, ucb_value=ucb_v, ucb_dx=ucb_g,
      model_flat=helpers.flatten(model), x=x, y=y, x2=x2, y2=y2, xq=xq,
      gram=kern(params, x, warp_func=wf), cross=kern(params, x, xq, warp_func=wf),
      mean_x=mean(params, x, warp_func=wf),
      nll=nll, nll0=key2nll[0], nll1=key2nll[1], nll_svd=nll_svd, grad_flat=helpers.flatten(grads),
      chol=chol, kinvy=kinvy, ymu=ymu, mu=mu, var=var, cov=cov,
      ei=o.expected_improvement_sub(mu_n, np.sqrt(var_n), target),
      pi=o.probability_of_improvement_sub(mu_n, np.sqrt(var_n), target + 0.1),
      ucb=o.ucb_sub(mu_n, np.sqrt(var_n), 3.0),
  )
  return name, out


def main():
  for case in CASES:
    name, out = build(case)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, 'nll', float(out['nll']))


if __name__ == '__main__':
  main()
