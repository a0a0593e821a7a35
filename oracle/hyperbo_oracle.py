"""CPU oracle: NumPy/SciPy restatement of HyperBO's GP hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / reported baseline.  `hyperbo_amd/` never imports it.

PARITY STATUS: **parity unpinned against the JAX reference.**  The reference
(google-research/hyperbo, /root/reference) is pure Python on JAX/Flax; `jax` is
not installed in the build container nor on the GPU box, and the reference's own
tests pin no numbers.  This restatement is therefore
pinned by (tests/test_oracle_*.py):
  (1) 50-digit `mpmath` recomputation of NLL / alpha / posterior / EI at N<=48,
  (2) central finite differences of the NLL for every parameter leaf,
  (3) an independent torch.autograd re-expression of the same formulas (CPU),
  (4) identities the reference's tests assert (SVD-NLL == Cholesky-NLL,
      diag(full_cov) == var, GP.predict == predict + noise, Gram symmetric PSD),
  (5) the NumPy-seeded matrices of hyperbo/basics/linalg_test.py:57-110,
  (6) scikit-learn's GaussianProcessRegressor (an independent third-party implementation of the same formulas: Gram, log
      marginal likelihood + gradient, posterior) for SE / Matern-3/2 / 5/2.
Golden fixtures generated from this file live in tests/golden/ (see
tests/golden/make_golden.py).

Every function cites the reference file:line it restates (paths relative to
/root/reference).  dtype follows the inputs (float64 or float32) like the
reference under JAX_ENABLE_X64=1 / default.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple, Union

import numpy as np
import scipy.linalg as spla
import scipy.special as spsp

EPS = 1e-10  # hyperbo/gp_utils/utils.py:26, hyperbo/basics/linalg.py:26


# ----------------------------------------------------------------------------
# containers -- hyperbo/basics/definitions.py:23-46
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class GPCache:
  chol: np.ndarray
  kinvy: np.ndarray
  needs_update: bool


class SubDataset(NamedTuple):
  x: np.ndarray
  y: np.ndarray
  aligned: Optional[Union[int, str, bool, Tuple[str, ...]]] = None


@dataclasses.dataclass
class GPParams:
  config: Dict[str, Any] = dataclasses.field(default_factory=dict)
  model: Dict[str, Any] = dataclasses.field(default_factory=dict)
  cache: Dict[Union[int, str], GPCache] = dataclasses.field(default_factory=dict)
  samples: List[Dict[str, Any]] = dataclasses.field(default_factory=list)


# ----------------------------------------------------------------------------
# warps -- hyperbo/gp_utils/utils.py:28-81
# ----------------------------------------------------------------------------
def identity_warp(x):
  return x


def softplus_warp(x):
  """jax.nn.softplus == logaddexp(x, 0)."""
  x = np.asarray(x)
  return np.logaddexp(x, np.zeros_like(x))


def default_softplus(x):
  return softplus_warp(x) + EPS  # utils.py:73


DEFAULT_WARP_FUNC = {  # utils.py:75-81
    'constant': identity_warp,
    'lengthscale': default_softplus,
    'signal_variance': default_softplus,
    'noise_variance': default_softplus,
    'dot_prod_sigma': default_softplus,
}


def warp_derivative(fn, x):
  """d warp / d raw for the closed set of warps (what jax.grad would chain)."""
  x = np.asarray(x)
  if fn is identity_warp:
    return np.ones_like(x)
  if fn is default_softplus or fn is softplus_warp:
    return spsp.expit(x)
  raise NotImplementedError(f'no analytic derivative for warp {fn}')


def retrieve_params(params, keys, warp_func=None):
  """hyperbo/basics/params_utils.py:90-111."""
  model_params = params.model
  if not set(keys).issubset(set(model_params.keys())):
    raise ValueError(f'Expected parameters are {sorted(keys)}, '
                     f'but received {sorted(model_params.keys())}.')
  if warp_func:
    return [warp_func[k](model_params[k]) if k in warp_func else model_params[k]
            for k in keys]
  return [model_params[k] for k in keys]


# ----------------------------------------------------------------------------
# MLP basis -- hyperbo/gp_utils/basis_functions.py:24-36 (flax Dense: y = x@W+b)
# ----------------------------------------------------------------------------
def mlp_apply(mlp_params, x):
  """tanh(Dense(f)) for every layer, tanh on the last too."""
  n_layers = len(mlp_params)
  for l in range(n_layers):
    layer = mlp_params[f'Dense_{l}']
    x = np.tanh(x @ np.asarray(layer['kernel']) + np.asarray(layer['bias']))
  return x


# ----------------------------------------------------------------------------
# kernels -- hyperbo/gp_utils/kernel.py:29-183
# ----------------------------------------------------------------------------
def _scaled_sqdist(vx1, vx2, lengthscale):
  a = vx1 / lengthscale
  b = vx2 / lengthscale
  d = a[:, None, :] - b[None, :, :]
  return np.sum(d * d, axis=-1)


def _pair_kernel_matrix(name, params, vx1, vx2, warp_func):
  if name == 'dot_product':  # kernel.py:126-145
    sigma, bias = retrieve_params(params, ['dot_prod_sigma', 'dot_prod_bias'], warp_func)
    return (vx1 @ vx2.T) / np.square(sigma) + np.square(bias)
  lengthscale, signal_variance = retrieve_params(
      params, ['lengthscale', 'signal_variance'], warp_func)
  lengthscale = np.asarray(lengthscale, dtype=vx1.dtype)
  sv = np.squeeze(np.asarray(signal_variance, dtype=vx1.dtype))
  u = _scaled_sqdist(vx1, vx2, lengthscale)
  if name == 'squared_exponential':  # kernel.py:63-81
    return sv * np.exp(-u / 2)
  if name == 'matern32':  # kernel.py:84-102
    r = np.sqrt(vx1.dtype.type(3)) * np.sqrt(u)
    return sv * (1 + r) * np.exp(-r)
  if name == 'matern52':  # kernel.py:105-123
    r = np.sqrt(vx1.dtype.type(5)) * np.sqrt(u)
    return sv * (1 + r + r**2 / 3) * np.exp(-r)
  raise ValueError(name)


def _make_kernel(name):
  def matrix_map(params, vx1, vx2=None, warp_func=None, diag=False):
    """kernel.py:33-58: (n1,n2) Gram; vx2=None -> vx1; diag only with vx2=None."""
    vx1 = np.asarray(vx1)
    if vx2 is None:
      if diag:
        # k(x_i, x_i) per point (kernel.py:54-57)
        if name == 'dot_product':
          sigma, bias = retrieve_params(params, ['dot_prod_sigma', 'dot_prod_bias'], warp_func)
          return np.sum(vx1 * vx1, axis=1) / np.square(sigma) + np.square(bias)
        _, sv = retrieve_params(params, ['lengthscale', 'signal_variance'], warp_func)
        return np.full((vx1.shape[0],), np.squeeze(np.asarray(sv, dtype=vx1.dtype)), dtype=vx1.dtype)
      vx2 = vx1
    return _pair_kernel_matrix(name, params, vx1, np.asarray(vx2), warp_func)
  matrix_map.__name__ = name
  matrix_map.base_name = name
  matrix_map.uses_mlp = False
  return matrix_map


squared_exponential = _make_kernel('squared_exponential')
matern32 = _make_kernel('matern32')
matern52 = _make_kernel('matern52')
dot_product = _make_kernel('dot_product')


def with_mlp_bases(kernel):
  """kernel.py:148-183."""
  def kernel_mlp(params, vx1, vx2=None, warp_func=None, diag=False):
    mlp_params, = retrieve_params(params, ['mlp_params'], warp_func)
    vx1 = mlp_apply(mlp_params, np.asarray(vx1))
    if vx2 is not None:
      vx2 = mlp_apply(mlp_params, np.asarray(vx2))
    return kernel(params, vx1, vx2, warp_func=warp_func, diag=diag)
  kernel_mlp.__name__ = 'kernel_mlp'  # the reference's name (contains 'mlp', gp.py:361)
  kernel_mlp.base_name = kernel.base_name
  kernel_mlp.uses_mlp = True
  return kernel_mlp


dot_product_mlp = with_mlp_bases(dot_product)
squared_exponential_mlp = with_mlp_bases(squared_exponential)
matern32_mlp = with_mlp_bases(matern32)
matern52_mlp = with_mlp_bases(matern52)


# ----------------------------------------------------------------------------
# means -- hyperbo/gp_utils/mean.py:30-79
# ----------------------------------------------------------------------------
def zero(params, vx, warp_func=None):
  return np.zeros((np.asarray(vx).shape[0], 1), dtype=np.asarray(vx).dtype)  # mean.py:54-57


def constant(params, vx, warp_func=None):
  val, = retrieve_params(params, ['constant'], warp_func)  # mean.py:60-64
  vx = np.asarray(vx)
  return np.full((vx.shape[0], 1), np.squeeze(np.asarray(val)), dtype=vx.dtype)


def linear(params, vx, warp_func=None):
  linear_mean, = retrieve_params(params, ['linear_mean'], warp_func)  # mean.py:67-70
  return np.asarray(vx) @ np.asarray(linear_mean['kernel']) + np.asarray(linear_mean['bias'])


def linear_mlp(params, vx, warp_func=None):
  mlp_params, = retrieve_params(params, ['mlp_params'], warp_func)  # mean.py:73-79
  return linear(params, mlp_apply(mlp_params, np.asarray(vx)), warp_func=warp_func)


# ----------------------------------------------------------------------------
# linalg -- hyperbo/basics/linalg.py:29-110
# ----------------------------------------------------------------------------
def solve_linear_system(coeff, b):
  """linalg.py:29-33.  Non-PD input -> NaNs (JAX cholesky semantics), no raise."""
  try:
    chol = spla.cholesky(coeff, lower=True, check_finite=False)
  except spla.LinAlgError:
    nan = np.full_like(coeff, np.nan)
    return nan, np.full_like(b, np.nan)
  kinvy = spla.cho_solve((chol, True), b, check_finite=False)
  return chol, kinvy


def compute_delta_y_and_cov(mean_func, cov_func, params, x, y, warp_func=None, eps=1e-6):
  """linalg.py:36-69."""
  y = y - np.atleast_2d(mean_func(params, x, warp_func=warp_func))
  noise_variance, = retrieve_params(params, ['noise_variance'], warp_func=warp_func)
  cov = cov_func(params, x, warp_func=warp_func)
  cov = cov + np.eye(len(x), dtype=cov.dtype) * cov.dtype.type(np.squeeze(noise_variance) + eps)
  return y.astype(cov.dtype), cov


def solve_gp_linear_system(mean_func, cov_func, params, x, y, warp_func=None, eps=1e-6):
  """linalg.py:72-110 -> (chol, kinvy, y - mu)."""
  y, cov = compute_delta_y_and_cov(mean_func, cov_func, params, x, y, warp_func, eps)
  chol, kinvy = solve_linear_system(cov, y)
  return chol, kinvy, y


# ----------------------------------------------------------------------------
# objective -- hyperbo/gp_utils/objectives.py:109-210
# ----------------------------------------------------------------------------
def _nll_sub_dataset_cholesky(mean_func, cov_func, params, vx, vy, warp_func):
  chol, kinvy, vy = solve_gp_linear_system(mean_func, cov_func, params, vx, vy, warp_func)
  # objectives.py:153-155 incl. the (m,m)+scalar broadcast quirk for m>1.
  with np.errstate(invalid='ignore', divide='ignore'):
    val = np.sum(0.5 * np.dot(vy.T, kinvy) + np.sum(np.log(np.diag(chol))) +
                 0.5 * len(vx) * np.log(2 * np.pi))
  return val


def _nll_sub_dataset_svd(mean_func, cov_func, params, vx, vy, warp_func):
  vy, cov = compute_delta_y_and_cov(mean_func, cov_func, params, vx, vy, warp_func)
  u, s, v = spla.svd(cov)  # objectives.py:166
  kinv = np.dot(v.T, np.dot(np.diag(s**-1), u.T))
  kinvy = np.dot(kinv, vy)
  return 0.5 * np.sum(np.dot(vy.T, kinvy) + np.sum(np.log(s)) + len(vx) * np.log(2 * np.pi))


def included_sub_datasets(dataset, exclude_aligned=True):
  """objectives.py:181-185 selection rule (skip aligned, skip empty)."""
  out = []
  for k, s in dataset.items():
    if exclude_aligned and s.aligned is not None:
      continue
    if s.x.shape[0] == 0:
      continue
    out.append((k, s))
  return out


def neg_log_marginal_likelihood(mean_func, cov_func, params, dataset, warp_func=None,
                                exclude_aligned=True, return_key2nll=False,
                                use_cholesky=True):
  total_nll = 0.
  key2nll = {}
  num = 0
  for k, s in included_sub_datasets(dataset, exclude_aligned):
    fn = _nll_sub_dataset_cholesky if use_cholesky else _nll_sub_dataset_svd
    key2nll[k] = fn(mean_func, cov_func, params, np.asarray(s.x), np.asarray(s.y), warp_func)
    total_nll += key2nll[k]
    num += 1
  total_nll = 0. if num == 0 else total_nll / num  # objectives.py:192-195
  if 'priors' in params.config:  # objectives.py:198-207
    for k in params.model:
      if k in params.config['priors']:
        val, = retrieve_params(params, [k], warp_func)
        total_nll -= params.config['priors'][k](val)
  if return_key2nll:
    return total_nll, key2nll
  return total_nll


# ----------------------------------------------------------------------------
# analytic gradient of the Cholesky NLL (what jax.value_and_grad returns at
# hyperbo/gp_utils/gp.py:134 / hyperbo/basics/lbfgs.py:238), via
#   dnll/dK = 1/2 (m^2 K^-1 - s s^T),  s = K^-1 (y-mu) 1_m ;  dnll/dmu = -m s.
# ----------------------------------------------------------------------------
def _tree_zeros_like(tree):
  if isinstance(tree, dict):
    return {k: _tree_zeros_like(v) for k, v in tree.items()}
  return np.zeros_like(np.asarray(tree, dtype=np.float64))


def _tree_add(a, b):
  if isinstance(a, dict):
    return {k: _tree_add(a[k], b[k]) for k in a}
  return a + b


def _tree_scale(a, c):
  if isinstance(a, dict):
    return {k: _tree_scale(v, c) for k, v in a.items()}
  return a * c


def _mlp_forward_cache(mlp_params, x):
  acts = [x]
  for l in range(len(mlp_params)):
    layer = mlp_params[f'Dense_{l}']
    x = np.tanh(x @ np.asarray(layer['kernel']) + np.asarray(layer['bias']))
    acts.append(x)
  return acts


def _mlp_backward(mlp_params, acts, dfeat):
  grads = {}
  g = dfeat
  for l in reversed(range(len(mlp_params))):
    layer = mlp_params[f'Dense_{l}']
    dz = g * (1 - acts[l + 1]**2)
    grads[f'Dense_{l}'] = {'kernel': acts[l].T @ dz, 'bias': dz.sum(axis=0)}
    g = dz @ np.asarray(layer['kernel']).T
  return grads


def nll_sub_dataset_value_and_grad(mean_func, cov_func, params, vx, vy, warp_func=None, eps=1e-6):
  """Value and d/d(raw params.model) of one sub-dataset's Cholesky NLL (float64 math)."""
  model = params.model
  vx = np.asarray(vx, dtype=np.float64)
  vy = np.asarray(vy, dtype=np.float64)
  n, m = vy.shape
  base = cov_func.base_name
  use_mlp_k = cov_func.uses_mlp
  mean_name = mean_func.__name__
  wf = warp_func or {}

  def warped(key):
    raw = np.asarray(model[key], dtype=np.float64)
    return (wf[key](raw) if key in wf else raw), raw

  def chain(key, g_warped):
    raw = np.asarray(model[key], dtype=np.float64)
    if key in wf:
      return np.asarray(g_warped) * warp_derivative(wf[key], raw)
    return np.asarray(g_warped)

  acts = None
  if use_mlp_k or mean_name == 'linear_mlp':
    mlp_params = model['mlp_params']
    acts = _mlp_forward_cache(mlp_params, vx)
  feat = acts[-1] if use_mlp_k else vx

  # mean
  if mean_name == 'zero':
    mu = np.zeros((n, 1))
  elif mean_name == 'constant':
    mu = np.full((n, 1), float(np.squeeze(warped('constant')[0])))
  elif mean_name == 'linear':
    lm = model['linear_mean']
    mu = vx @ np.asarray(lm['kernel'], dtype=np.float64) + np.asarray(lm['bias'], dtype=np.float64)
  elif mean_name == 'linear_mlp':
    lm = model['linear_mean']
    mu = acts[-1] @ np.asarray(lm['kernel'], dtype=np.float64) + np.asarray(lm['bias'], dtype=np.float64)
  else:
    raise ValueError(mean_name)
  r = vy - mu

  noise, _ = warped('noise_variance')
  noise = float(np.squeeze(noise))
  if base == 'dot_product':
    sigma, _ = warped('dot_prod_sigma')
    bias, _ = warped('dot_prod_bias')
    sigma = float(np.squeeze(sigma)); bias = float(np.squeeze(bias))
    dots = feat @ feat.T
    kmat = dots / sigma**2 + bias**2
  else:
    ls, _ = warped('lengthscale')
    sv, _ = warped('signal_variance')
    sv = float(np.squeeze(sv))
    ls_vec = np.broadcast_to(np.asarray(ls, dtype=np.float64).reshape(-1), (feat.shape[1],)) \
        if np.asarray(ls).size in (1, feat.shape[1]) else None
    if ls_vec is None:
      raise ValueError('lengthscale must be scalar or of feature dimension')
    diff = feat[:, None, :] - feat[None, :, :]
    u = np.sum((diff / ls_vec)**2, axis=-1)
    if base == 'squared_exponential':
      kmat = sv * np.exp(-u / 2)
      dk_du = -0.5 * kmat
    elif base == 'matern32':
      rr = np.sqrt(3.0 * u)
      kmat = sv * (1 + rr) * np.exp(-rr)
      dk_du = -sv * 3.0 * np.exp(-rr) / 2
    elif base == 'matern52':
      rr = np.sqrt(5.0 * u)
      kmat = sv * (1 + rr + rr**2 / 3) * np.exp(-rr)
      dk_du = -sv * 5.0 * np.exp(-rr) * (1 + rr) / 6
    else:
      raise ValueError(base)
    # hyperbo/basics/linalg.py:183-188: where the sqrt argument is exactly 0 the
    # cotangent is 1e6*g, multiplied by d u/d theta = 0 -> contributes 0.
    if base != 'squared_exponential':
      dk_du = np.where(u == 0, 0.0, dk_du)
  cov = kmat + np.eye(n) * (noise + eps)
  try:
    chol = spla.cholesky(cov, lower=True)
  except spla.LinAlgError:
    return float('nan'), _tree_scale(_tree_zeros_like(model), float('nan'))
  alpha = spla.cho_solve((chol, True), r)
  s = alpha.sum(axis=1, keepdims=True)
  value = float(0.5 * np.sum(r.T @ alpha) + m * m * (np.sum(np.log(np.diag(chol))) + 0.5 * n * np.log(2 * np.pi)))
  kinv = spla.cho_solve((chol, True), np.eye(n))
  gmat = 0.5 * (m * m * kinv - s @ s.T)
  dmu = -m * s  # (n,1)

  grads = _tree_zeros_like(model)
  grads['noise_variance'] = chain('noise_variance', np.trace(gmat)).reshape(np.shape(model['noise_variance']))
  dfeat = np.zeros_like(feat)
  if base == 'dot_product':
    g_sigma = np.sum(gmat * dots) * (-2.0 / sigma**3)
    g_bias = np.sum(gmat) * 2.0 * bias
    grads['dot_prod_sigma'] = chain('dot_prod_sigma', g_sigma).reshape(np.shape(model['dot_prod_sigma']))
    grads['dot_prod_bias'] = chain('dot_prod_bias', g_bias).reshape(np.shape(model['dot_prod_bias']))
    if use_mlp_k:
      dfeat = dfeat + 2.0 * (gmat @ feat) / sigma**2
  else:
    grads['signal_variance'] = chain('signal_variance', np.sum(gmat * kmat) / sv).reshape(
        np.shape(model['signal_variance']))
    gw = gmat * dk_du  # (n,n)
    # d u / d ls_d = -2 diff_d^2 / ls_d^3
    per_dim = np.einsum('ij,ijd->d', gw, diff**2) * (-2.0 / ls_vec**3)
    if np.asarray(model['lengthscale']).size == 1:
      g_ls = np.sum(per_dim)
    else:
      g_ls = per_dim
    grads['lengthscale'] = chain('lengthscale', np.reshape(g_ls, np.shape(model['lengthscale'])))
    if use_mlp_k:
      # d u_ij / d f_i = 2 diff_ij / ls^2 ; symmetric contributions (i and j roles)
      dfeat = dfeat + 4.0 * np.einsum('ij,ijd->id', gw, diff) / ls_vec**2
  # mean parameters
  if mean_name == 'constant':
    grads['constant'] = chain('constant', np.sum(dmu)).reshape(np.shape(model['constant']))
  elif mean_name in ('linear', 'linear_mlp'):
    inp = vx if mean_name == 'linear' else acts[-1]
    lm = model['linear_mean']
    grads['linear_mean'] = {
        'kernel': (inp.T @ dmu).reshape(np.shape(lm['kernel'])),
        'bias': np.sum(dmu, axis=0).reshape(np.shape(lm['bias'])),
    }
    if mean_name == 'linear_mlp':
      dfeat_mean = dmu @ np.asarray(lm['kernel'], dtype=np.float64).T
      if use_mlp_k:
        dfeat = dfeat + dfeat_mean
      else:
        dfeat = dfeat_mean
  if acts is not None and 'mlp_params' in model:
    if use_mlp_k or mean_name == 'linear_mlp':
      grads['mlp_params'] = _mlp_backward(model['mlp_params'], acts, dfeat)
  return value, grads


def nll_value_and_grad(mean_func, cov_func, params, dataset, warp_func=None,
                       exclude_aligned=True, priors_grad=None):
  """Mean-over-tasks NLL and its gradient pytree (shape of params.model).

  `priors_grad`: optional dict key -> callable(warped value) -> d log_prior / d warped
  (the reference chains arbitrary prior callables through jax.grad; here analytic).
  """
  total = 0.
  grads = _tree_zeros_like(params.model)
  num = 0
  for _, s in included_sub_datasets(dataset, exclude_aligned):
    v, g = nll_sub_dataset_value_and_grad(mean_func, cov_func, params, s.x, s.y, warp_func)
    total += v
    grads = _tree_add(grads, g)
    num += 1
  if num:
    total /= num
    grads = _tree_scale(grads, 1.0 / num)
  if 'priors' in params.config:
    wf = warp_func or {}
    for k in params.model:
      if k in params.config['priors']:
        val, = retrieve_params(params, [k], warp_func)
        total -= float(params.config['priors'][k](val))
        if priors_grad is not None and k in priors_grad:
          raw = np.asarray(params.model[k], dtype=np.float64)
          dwarp = warp_derivative(wf[k], raw) if k in wf else np.ones_like(raw)
          grads[k] = grads[k] - np.reshape(priors_grad[k](val), raw.shape) * dwarp
  return total, grads


# ----------------------------------------------------------------------------
# posterior -- hyperbo/gp_utils/gp.py:242-305, GP.predict post-processing :562-620
# ----------------------------------------------------------------------------
def predict(mean_func, cov_func, params, x_observed, y_observed, x_query,
            warp_func=None, full_cov=False, cache=None):
  if x_observed is None or x_observed.shape[0] == 0:  # gp.py:275-282
    mu = mean_func(params, x_query, warp_func=warp_func)
    cov = cov_func(params, x_query, warp_func=warp_func, diag=not full_cov)
    return (mu, cov) if full_cov else (mu, cov[:, None])
  if cache is None:
    chol, kinvy, _ = solve_gp_linear_system(mean_func, cov_func, params, x_observed,
                                            y_observed, warp_func)
  else:
    chol, kinvy = cache.chol, cache.kinvy
  cov = cov_func(params, x_observed, x_query, warp_func=warp_func)  # gp.py:295
  mu = np.dot(cov.T, kinvy) + mean_func(params, x_query, warp_func=warp_func)
  v = spla.solve_triangular(chol, cov, lower=True, check_finite=False)  # gp.py:297
  if full_cov:
    return mu, cov_func(params, x_query, warp_func=warp_func) - np.dot(v.T, v)
  var = cov_func(params, x_query, warp_func=warp_func, diag=True) - np.sum(v * v, axis=0)
  return mu, var[:, None]


def gp_predict_postprocess(params, dataset, mu, cov, warp_func, full_cov, with_noise, unbiased):
  """gp.py:607-619."""
  cov = np.array(cov, copy=True)
  if with_noise:
    noise_variance, = retrieve_params(params, ['noise_variance'], warp_func=warp_func)
    nv = cov.dtype.type(np.squeeze(noise_variance))
    if full_cov:
      cov += np.eye(cov.shape[0], dtype=cov.dtype) * nv
    else:
      cov += nv
  if unbiased:
    len_dataset = len([k for k, v in dataset.items() if v.aligned is None])
    if len_dataset > 1:
      cov *= cov.dtype.type(len_dataset / (len_dataset - 1.))
  return mu, cov


# ----------------------------------------------------------------------------
# acquisition -- hyperbo/bo_utils/acfun.py:96-185
# ----------------------------------------------------------------------------
def _norm_pdf(x):
  return np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi).astype(x.dtype) if isinstance(x, np.ndarray) \
      else math.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def _norm_cdf(x):
  return spsp.ndtr(x)


def expected_improvement_sub(mu, std, target):
  gamma = (target - mu) / std  # acfun.py:108-110
  return (_norm_pdf(gamma) - gamma * (1 - _norm_cdf(gamma))) * std


def probability_of_improvement_sub(mu, std, target):
  return -((target - mu) / std)  # acfun.py:125-126


def ucb_sub(mu, std, beta=3.):
  return mu + beta * std  # acfun.py:142


def ei_callback_default(dataset, key):
  if key not in dataset or dataset[key].y.shape[0] == 0:  # acfun.py:145-148
    return 0.0
  return float(np.max(dataset[key].y))


def pi_callback_default(dataset, key, zeta=0.1, use_std=False):
  if key not in dataset or dataset[key].y.shape[0] == 0:  # acfun.py:160-166
    return 0.0
  if use_std:
    return float(np.max(dataset[key].y) + zeta * np.std(dataset[key].y))
  return float(np.max(dataset[key].y) + zeta)


# ----------------------------------------------------------------------------
# priors -- hyperbo/gp_utils/priors.py:37-45 (tfp Normal / LogNormal log_prob sums)
# ----------------------------------------------------------------------------
def normal_log_prob(x, loc, scale):
  x = np.asarray(x, dtype=np.float64)
  return -0.5 * ((x - loc) / scale)**2 - np.log(scale) - 0.5 * np.log(2 * np.pi)


def lognormal_log_prob(x, loc, scale):
  x = np.asarray(x, dtype=np.float64)
  return normal_log_prob(np.log(x), loc, scale) - np.log(x)


noise_prior = lambda x: float(np.sum(normal_log_prob(x, 0., 0.1)))
lognormal_prior = lambda x: float(np.sum(lognormal_log_prob(x, 0., 1.)))
constant_prior = lambda x: float(np.sum(normal_log_prob(x, 0., 1.)))
DEFAULT_PRIORS = {'noise_variance': noise_prior, 'signal_variance': lognormal_prior,
                  'constant': constant_prior}
DEFAULT_PRIORS_GRAD = {
    'noise_variance': lambda x: -np.asarray(x, dtype=np.float64) / 0.01,
    'signal_variance': lambda x: -(np.log(np.asarray(x, dtype=np.float64)) + 1.0) / np.asarray(x, dtype=np.float64),
    'constant': lambda x: -np.asarray(x, dtype=np.float64),
}


# ----------------------------------------------------------------------------
# divergence objectives -- hyperbo/gp_utils/objectives.py:29-106, hyperbo/gp_utils/utils.py:84-173,
# hyperbo/basics/linalg.py:113-126 (svd_matrix_sqrt)
# ----------------------------------------------------------------------------
def svd_matrix_sqrt(cov):
  u, s, _ = spla.svd(cov)  # linalg.py:122
  factor = u * np.sqrt(s[..., None, :])
  tol = s.max() * np.finfo(s.dtype).eps / 2. * np.sqrt(2 * cov.shape[0] + 1.)
  rank = np.count_nonzero(s > tol)
  return factor[:, :rank]


def partial_kl_mvn(mu0, cov0, mu1, cov1):
  """utils.py:84-106: tr(cov1^-1 cov0) + (mu1-mu0)^T cov1^-1 (mu1-mu0) + logdet cov1."""
  mu_diff = mu1 - mu0
  chol1 = spla.cholesky(cov1, lower=True)
  trcov1invcov0 = np.trace(spla.cho_solve((chol1, True), cov0))
  mahalanobis = float(np.dot(mu_diff, spla.cho_solve((chol1, True), mu_diff)))
  logdetcov1 = np.sum(2 * np.log(np.diag(chol1)))
  return trcov1invcov0 + mahalanobis + logdetcov1


def kl_multivariate_normal(mu0, cov0, mu1, cov1, weight=1.0, eps=0.0, partial=True):
  """utils.py:109-148."""
  cov0 = np.atleast_2d(cov0); cov1 = np.atleast_2d(cov1)
  if eps > 0.:
    cov0 = cov0 + np.eye(cov0.shape[0]) * eps
    cov1 = cov1 + np.eye(cov1.shape[0]) * eps
  if partial:
    return weight * partial_kl_mvn(mu0, cov0, mu1, cov1)
  chol0 = svd_matrix_sqrt(cov0)
  chol0inv = np.linalg.pinv(chol0)
  mu1 = np.dot(chol0inv, mu1 - mu0)
  cov1 = np.dot(np.dot(chol0inv, cov1), chol0inv.T)
  mu0 = np.zeros_like(mu1)
  cov0 = np.eye(cov1.shape[0])
  return weight * 0.5 * (partial_kl_mvn(mu0, cov0, mu1, cov1) - chol0.shape[1])


def euclidean_multivariate_normal(mu0, cov0, mu1, cov1, mean_weight=1., cov_weight=1., **unused_kwargs):
  """utils.py:151-173 (safe_l2norm == l2 norm in value)."""
  return mean_weight * np.sqrt(np.sum((mu0 - mu1)**2)) + cov_weight * np.sqrt(np.sum((cov0 - cov1)**2))


def included_aligned_sub_datasets(dataset):
  """objectives.py:85-96: aligned sub-datasets with data; malformed y raises."""
  out = []
  for k, s in dataset.items():
    if s.aligned is None:
      continue
    if s.x.shape[0] == 0:
      continue
    if s.y.shape[1] == 0 or s.y.shape[0] != s.x.shape[0]:
      raise ValueError(f'dataset[{k}].x has shape {s.x.shape} but dataset[{k}].y has shape {s.y.shape}')
    out.append((k, s))
  return out


def multivariate_normal_divergence(mean_func, cov_func, params, dataset, warp_func=None,
                                   distance=kl_multivariate_normal):
  """objectives.py:29-101: mean over aligned sub-datasets of distance(N(mu_data, cov_data), GP model)."""
  total, num = 0., 0
  for _, s in included_aligned_sub_datasets(dataset):
    y = np.asarray(s.y); x = np.asarray(s.x)
    mu_data = np.mean(y, axis=1)
    cov_data = np.atleast_2d(np.cov(y, bias=True))
    mu_model = mean_func(params, x, warp_func=warp_func).flatten()
    noise_variance, = retrieve_params(params, ['noise_variance'], warp_func=warp_func)
    cov_model = cov_func(params, x, warp_func=warp_func) + np.eye(x.shape[0]) * np.squeeze(noise_variance)
    total += distance(mu0=mu_data, cov0=cov_data, mu1=mu_model, cov1=cov_model)
    num += 1
  return 0. if num == 0 else total / num


def _divergence_sub_dataset_value_and_grad(kind, mean_func, cov_func, params, vx, vy, warp_func):
  """Value and d/d(raw params.model) of one aligned sub-dataset's partial KL ('ekl') or Euclidean ('euc')
  divergence.  Implemented by re-using the NLL machinery's chain rule with
     ekl:  f = tr(K1^-1 C0) + d^T K1^-1 d + logdet K1,   df/dK1 = K1^-1 - K1^-1 (C0 + d d^T) K1^-1,  df/dmu1 = 2 K1^-1 d
     euc:  f = |d| + |C0 - K1|_F,                          df/dK1 = (K1 - C0)/|C0-K1|_F,               df/dmu1 = d/|d|
  (d = mu1 - mu0, K1 = K + noise I, no jitter: objectives.py:63-65)."""
  model = params.model
  vx = np.asarray(vx, dtype=np.float64); vy = np.asarray(vy, dtype=np.float64)
  n, m = vy.shape
  mu0 = vy.mean(axis=1)
  yc = vy - mu0[:, None]
  c0 = yc @ yc.T / m
  # forward pieces through the NLL helper's internals: recompute K, features, etc.
  val_dummy, _ = 0.0, None
  base = cov_func.base_name; use_mlp_k = cov_func.uses_mlp; mean_name = mean_func.__name__
  wf = warp_func or {}

  def warped(key):
    raw = np.asarray(model[key], dtype=np.float64)
    return wf[key](raw) if key in wf else raw

  def chain(key, g):
    raw = np.asarray(model[key], dtype=np.float64)
    return np.asarray(g) * (warp_derivative(wf[key], raw) if key in wf else 1.0)

  acts = None
  if use_mlp_k or mean_name == 'linear_mlp':
    acts = _mlp_forward_cache(model['mlp_params'], vx)
  feat = acts[-1] if use_mlp_k else vx
  if mean_name == 'zero':
    mu1 = np.zeros(n)
  elif mean_name == 'constant':
    mu1 = np.full(n, float(np.squeeze(warped('constant'))))
  else:
    lm = model['linear_mean']
    inp = vx if mean_name == 'linear' else acts[-1]
    mu1 = (inp @ np.asarray(lm['kernel'], dtype=np.float64) + np.asarray(lm['bias'], dtype=np.float64))[:, 0]
  noise = float(np.squeeze(warped('noise_variance')))
  if base == 'dot_product':
    sigma = float(np.squeeze(warped('dot_prod_sigma'))); bias = float(np.squeeze(warped('dot_prod_bias')))
    dots = feat @ feat.T
    kmat = dots / sigma**2 + bias**2
  else:
    ls = np.broadcast_to(np.asarray(warped('lengthscale'), dtype=np.float64).reshape(-1), (feat.shape[1],))
    sv = float(np.squeeze(warped('signal_variance')))
    diff = feat[:, None, :] - feat[None, :, :]
    u = np.sum((diff / ls)**2, axis=-1)
    if base == 'squared_exponential':
      kmat = sv * np.exp(-u / 2); dk_du = -0.5 * kmat
    elif base == 'matern32':
      rr = np.sqrt(3.0 * u); kmat = sv * (1 + rr) * np.exp(-rr); dk_du = np.where(u == 0, 0.0, -sv * 1.5 * np.exp(-rr))
    else:
      rr = np.sqrt(5.0 * u); kmat = sv * (1 + rr + rr**2 / 3) * np.exp(-rr)
      dk_du = np.where(u == 0, 0.0, -sv * 5.0 * np.exp(-rr) * (1 + rr) / 6)
  k1 = kmat + np.eye(n) * noise
  d = mu1 - mu0
  if kind == 'ekl':
    chol = spla.cholesky(k1, lower=True)
    kinv = spla.cho_solve((chol, True), np.eye(n))
    value = float(np.sum(kinv * c0) + d @ kinv @ d + 2 * np.sum(np.log(np.diag(chol))))
    gmat = kinv - kinv @ (c0 + np.outer(d, d)) @ kinv
    dmu = 2 * kinv @ d
  else:
    nd = np.sqrt(np.sum(d * d)); fn = np.sqrt(np.sum((c0 - k1)**2))
    value = float(nd + fn)
    gmat = (k1 - c0) / fn if fn > 0 else np.zeros_like(k1)
    dmu = d / nd if nd > 0 else np.zeros_like(d)
  dmu = dmu[:, None]
  grads = _tree_zeros_like(model)
  grads['noise_variance'] = chain('noise_variance', np.trace(gmat)).reshape(np.shape(model['noise_variance']))
  dfeat = np.zeros_like(acts[-1]) if acts is not None else None   # d f / d (MLP output)
  if base == 'dot_product':
    grads['dot_prod_sigma'] = chain('dot_prod_sigma', np.sum(gmat * dots) * (-2.0 / sigma**3)).reshape(np.shape(model['dot_prod_sigma']))
    grads['dot_prod_bias'] = chain('dot_prod_bias', np.sum(gmat) * 2.0 * bias).reshape(np.shape(model['dot_prod_bias']))
    if use_mlp_k:
      dfeat += 2.0 * (gmat @ feat) / sigma**2
  else:
    grads['signal_variance'] = chain('signal_variance', np.sum(gmat * kmat) / sv).reshape(np.shape(model['signal_variance']))
    gw = gmat * dk_du
    per_dim = np.einsum('ij,ijd->d', gw, diff**2) * (-2.0 / ls**3)
    g_ls = np.sum(per_dim) if np.asarray(model['lengthscale']).size == 1 else per_dim
    grads['lengthscale'] = chain('lengthscale', np.reshape(g_ls, np.shape(model['lengthscale'])))
    if use_mlp_k:
      dfeat += 4.0 * np.einsum('ij,ijd->id', gw, diff) / ls**2
  if mean_name == 'constant':
    grads['constant'] = chain('constant', np.sum(dmu)).reshape(np.shape(model['constant']))
  elif mean_name in ('linear', 'linear_mlp'):
    inp = vx if mean_name == 'linear' else acts[-1]
    lm = model['linear_mean']
    grads['linear_mean'] = {'kernel': (inp.T @ dmu).reshape(np.shape(lm['kernel'])),
                            'bias': np.sum(dmu, axis=0).reshape(np.shape(lm['bias']))}
    if mean_name == 'linear_mlp':
      dfeat = dfeat + dmu @ np.asarray(lm['kernel'], dtype=np.float64).T
  if acts is not None and (use_mlp_k or mean_name == 'linear_mlp'):
    grads['mlp_params'] = _mlp_backward(model['mlp_params'], acts, dfeat)
  return value, grads


def divergence_value_and_grad(kind, mean_func, cov_func, params, dataset, warp_func=None):
  """Mean over aligned sub-datasets (objectives.py:98-101) of the 'ekl' / 'euc' divergence and its gradient."""
  total = 0.; grads = _tree_zeros_like(params.model); num = 0
  for _, s in included_aligned_sub_datasets(dataset):
    v, g = _divergence_sub_dataset_value_and_grad(kind, mean_func, cov_func, params, s.x, s.y, warp_func)
    total += v; grads = _tree_add(grads, g); num += 1
  if num:
    total /= num; grads = _tree_scale(grads, 1.0 / num)
  return total, grads


# ----------------------------------------------------------------------------
# d acquisition / d x_query -- what jaxopt.ScipyBoundedMinimize differentiates in bayesopt()
# (hyperbo/bo_utils/bayesopt.py:116-125: f(x) = -ac_func(model, key, x[None]))
# ----------------------------------------------------------------------------
def acquisition_value_and_grad(acq_name, mean_func, cov_func, params, x_observed, y_observed, x_query, acq_param,
                               warp_func=None, add_noise=0.0, scale=1.0):
  """Values (M,1) and d value_q / d x_query[q] (M,D) of EI / PI / UCB on the posterior of gp.py:242-305 with the
  GP.predict post-processing var' = (var + add_noise) * scale (gp.py:607-619).  Queries are independent.

  mu = k(x,X) alpha + m(x),  var = k(x,x) - k(x,X) K^-1 k(X,x):
    d mu/dx = sum_i alpha_i dk(x,X_i)/dx + dm/dx,   d var/dx = dk(x,x)/dx - 2 sum_i beta_i dk(x,X_i)/dx,  beta = K^-1 k(X,x)
  EI = s (phi(u) + u Phi(u)), u = (mu - t)/s: dEI/dmu = Phi(u), dEI/ds = phi(u);  PI(-gamma) = (mu - t)/s;  UCB = mu + beta s.
  Matern kernels: zero-distance pairs contribute 0 (linalg.py:183-188 safe sqrt)."""
  xq = np.asarray(x_query, dtype=np.float64)
  mq, dim = xq.shape
  model = params.model
  base = cov_func.base_name; use_mlp_k = cov_func.uses_mlp; mean_name = mean_func.__name__
  has_data = x_observed is not None and np.shape(x_observed)[0] > 0
  mlp_needed = use_mlp_k or mean_name == 'linear_mlp'
  acts_q = _mlp_forward_cache(model['mlp_params'], xq) if mlp_needed else None
  fq = acts_q[-1] if use_mlp_k else xq
  mu0 = np.asarray(mean_func(params, xq, warp_func=warp_func), dtype=np.float64)[:, 0]
  kdiag = np.asarray(cov_func(params, xq, warp_func=warp_func, diag=True), dtype=np.float64)
  if has_data:
    xo = np.asarray(x_observed, dtype=np.float64)
    chol, kinvy, _ = solve_gp_linear_system(mean_func, cov_func, params, xo, np.asarray(y_observed, dtype=np.float64),
                                            warp_func)
    fo = mlp_apply(model['mlp_params'], xo) if use_mlp_k else xo
    kxq = np.asarray(cov_func(params, xo, xq, warp_func=warp_func), dtype=np.float64)     # (n, M)
    alpha = kinvy[:, 0]
    beta = spla.cho_solve((chol, True), kxq)                                               # (n, M)
    mu = kxq.T @ alpha + mu0
    var = kdiag - np.sum(kxq * beta, axis=0)
  else:
    mu, var = mu0, kdiag
  v2 = (var + add_noise) * scale
  sd = np.sqrt(v2)
  if acq_name == 'ei':
    u = (mu - acq_param) / sd
    val = sd * (_norm_pdf(u) + u * _norm_cdf(u)); da_dmu = _norm_cdf(u); da_dsd = _norm_pdf(u)
  elif acq_name == 'pi':
    val = (mu - acq_param) / sd; da_dmu = 1.0 / sd; da_dsd = -(mu - acq_param) / sd**2
  else:
    val = mu + acq_param * sd; da_dmu = np.ones_like(mu); da_dsd = np.full_like(mu, acq_param)
  da_dvar = da_dsd / (2 * sd) * scale
  gfeat = np.zeros_like(fq)
  if base == 'dot_product':
    sigma, = retrieve_params(params, ['dot_prod_sigma'], warp_func)
    s2 = float(np.squeeze(sigma))**2
    gfeat += (da_dvar * 2.0 / s2)[:, None] * fq
    if has_data:
      coef = da_dmu[None, :] * alpha[:, None] - 2.0 * da_dvar[None, :] * beta               # (n, M)
      gfeat += coef.T @ fo / s2
  elif has_data:
    ls, sv = retrieve_params(params, ['lengthscale', 'signal_variance'], warp_func)
    ls = np.broadcast_to(np.asarray(ls, dtype=np.float64).reshape(-1), (fq.shape[1],)); sv = float(np.squeeze(sv))
    diff = fq[:, None, :] - fo[None, :, :]                                                 # (M, n, F)
    u2 = np.sum((diff / ls)**2, axis=-1)
    if base == 'squared_exponential':
      dk_du = -0.5 * sv * np.exp(-u2 / 2)
    elif base == 'matern32':
      rr = np.sqrt(3.0 * u2); dk_du = np.where(u2 == 0, 0.0, -sv * 1.5 * np.exp(-rr))
    else:
      rr = np.sqrt(5.0 * u2); dk_du = np.where(u2 == 0, 0.0, -sv * 5.0 * np.exp(-rr) * (1 + rr) / 6)
    coef = (da_dmu[None, :] * alpha[:, None] - 2.0 * da_dvar[None, :] * beta).T             # (M, n)
    gfeat += 2.0 * np.einsum('qi,qid->qd', coef * dk_du, diff) / ls**2
  grad = np.zeros((mq, dim))
  gmlp = np.zeros_like(acts_q[-1]) if mlp_needed else None
  if use_mlp_k:
    gmlp += gfeat
  else:
    grad += gfeat
  if mean_name == 'linear':
    grad += da_dmu[:, None] * np.asarray(model['linear_mean']['kernel'], dtype=np.float64)[:, 0][None, :]
  elif mean_name == 'linear_mlp':
    gmlp += da_dmu[:, None] * np.asarray(model['linear_mean']['kernel'], dtype=np.float64)[:, 0][None, :]
  if mlp_needed:
    # backward through tanh(Dense) layers to the inputs
    n_layers = len(model['mlp_params'])
    g = gmlp
    for l in range(n_layers - 1, -1, -1):
      out = acts_q[l + 1]   # acts[0] is the input
      dz = g * (1.0 - out * out)
      g = dz @ np.asarray(model['mlp_params'][f'Dense_{l}']['kernel'], dtype=np.float64).T
    grad += g
  return val[:, None], grad
