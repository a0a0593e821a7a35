"""GP model object and posterior -- hyperbo/gp_utils/gp.py:242-305 (predict), :308-620 (GP).

Same method names / arguments / cache semantics as the reference; factorisation, posterior and
NLL run on the GPU; `infer_parameters` / `GP.train` (gp.py:53-195, 454-485) are the host loops
(Adam, L-BFGS) around the native value_and_grad.
"""
import ctypes as C
from typing import Any, Callable, Dict, List, Tuple, Union

import concurrent.futures

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.basics import linalg
from hyperbo_amd.basics import params_utils
from hyperbo_amd.gp_utils import objectives as obj

retrieve_params = params_utils.retrieve_params
GPCache = defs.GPCache
SubDataset = defs.SubDataset
GPParams = defs.GPParams


class _Adam:
  """optax.adam(learning_rate) defaults: b1=0.9, b2=0.999, eps=1e-8 (used at gp.py:124-144)."""

  def __init__(self, lr, b1=0.9, b2=0.999, eps=1e-8):
    self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
    self.m = self.v = None
    self.t = 0

  def step(self, x, g):
    if self.m is None:
      self.m, self.v = np.zeros_like(x), np.zeros_like(x)
    self.t += 1
    self.m = self.b1 * self.m + (1 - self.b1) * g
    self.v = self.b2 * self.v + (1 - self.b2) * g * g
    mhat = self.m / (1 - self.b1**self.t)
    vhat = self.v / (1 - self.b2**self.t)
    return x - self.lr * mhat / (np.sqrt(vhat) + self.eps)


def _value_and_grad_of(objective):
  """The `.value_and_grad` companion of an objective (native NLL / EKL / Euclid and their add / mul sums)."""
  vg = getattr(objective, 'value_and_grad', None)
  if vg is None:
    raise NotImplementedError(
        f'objective {objective!r} has no value_and_grad companion: there is no autodiff on this path, '
        'attach `objective.value_and_grad = fn(mean_func, cov_func, params, dataset, warp_func)`')
  return vg


def sample_from_gp(key, mean_func, cov_func, params, x, warp_func=None, num_samples=1, method='cholesky', eps=1e-6):
  """gp.py:198-240: (n, num_samples) draws of N(mean(x), K(x,x) + (noise + eps) I).  The Gram matrix comes from the device
  (hbo_gram); the factor `F` with F F^T = cov follows jax.random.multivariate_normal's three methods: 'cholesky' on the device
  (hbo_spd_solve), 'svd' (F = U sqrt(s)) and 'eigh' (F = V sqrt(w)) on host LAPACK like the SVD of the SVD-NLL (a reporting
  path); the O(n^2 S) product with the normal draws is on the host.  `key`: NumPy Generator or seed (the
  threefry streams of a JAX PRNG key cannot be reproduced without jax)."""
  if method not in ('cholesky', 'svd', 'eigh'):
    raise ValueError("method must be one of {'svd', 'eigh', 'cholesky'}")   # the message of jax.random.multivariate_normal
  from hyperbo_amd.basics import linalg
  rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(0 if key is None else key)
  x = np.asarray(x)
  n = x.shape[0]
  mu = np.asarray(mean_func(params, x, warp_func=warp_func), dtype=np.float64).reshape(-1)
  noise_variance, = params_utils.retrieve_params(params, ['noise_variance'], warp_func=warp_func)
  cov = np.asarray(cov_func(params, x, warp_func=warp_func), dtype=np.float64)
  cov = cov + np.eye(n) * (float(np.squeeze(noise_variance)) + eps)
  if method == 'cholesky':
    factor, _ = linalg.solve_linear_system(cov, np.zeros((n, 1)))
  elif method == 'svd':
    u, sv, _ = np.linalg.svd(cov)
    factor = u * np.sqrt(sv[None, :])
  else:
    w, v = np.linalg.eigh(cov)
    factor = v * np.sqrt(w[None, :])
  return mu[:, None] + factor @ rng.standard_normal((n, num_samples))


def infer_parameters(mean_func, cov_func, init_params, dataset, warp_func=None,
                     objective=obj.neg_log_marginal_likelihood, key=None, get_params_path=None, callback=None):
  """Posterior inference for a meta GP -- the training driver of hyperbo/gp_utils/gp.py:53-195.

  Host loop (Adam on per-step sub-sampled batches, or L-BFGS on one sub-sample) around the native
  `nll_value_and_grad`.  `key`: numpy Generator or seed (JAX PRNG keys are not reproducible here).
  NaN at step 0 raises, a non-finite loss later stops and keeps the last finite parameters
  (gp.py:135-142); the cache is cleared on return (gp.py:194).
  """
  from hyperbo_amd.basics import data_utils, lbfgs as lbfgs_lib
  if not dataset:
    return init_params
  rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(0 if key is None else key)
  params = init_params
  method = params.config['method']
  batch_size = params.config['batch_size']
  max_training_step = params.config['max_training_step']
  if max_training_step <= 0 and method != 'slice_sample':
    return init_params
  vg = _value_and_grad_of(objective)

  def loss_and_grad(model_params, batch):
    p = GPParams(model=model_params, config=init_params.config)
    return vg(mean_func=mean_func, cov_func=cov_func, params=p, dataset=batch, warp_func=warp_func)

  def make_device(batch):
    return obj.DeviceBatch(batch) if getattr(vg, 'accepts_device_batch', False) else batch

  if method == 'adam':
    needs_resample = any(s.x.shape[0] >= batch_size for s in dataset.values())
    # native objectives: the dataset stays resident in HBM and every step's batch is gathered there from freshly drawn row
    # indices (the reference indexes device arrays too); otherwise the host iterator of data_utils.py:72-100
    resident = obj.DeviceBatch(dataset) if (needs_resample and getattr(vg, 'accepts_device_batch', False)) else None
    if resident is not None:
      index_iter = data_utils.sub_sample_index_iterator(rng, dataset, batch_size)
      dataset_iter = (resident.subsample(ix) for ix in index_iter)
      make_device = lambda b: b
    else:
      dataset_iter = data_utils.sub_sample_dataset_iterator(rng, dataset, batch_size)
    x, unflatten = lbfgs_lib.tree_flatten(params.model)
    opt = _Adam(params.config['learning_rate'])
    dev = None
    current_loss = None
    # the next batch is drawn on a helper thread while the device evaluates the current one (the C call releases the GIL;
    # only this iterator uses `rng`, so the draws keep their order): 64 tasks x 2000 points, batch 500: 2.9 -> 1.9 ms per step,
    # 1.45 with the batch gathered on the device
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1) if needs_resample else None
    pending = pool.submit(next, dataset_iter) if pool else None
    try:
      for i in range(max_training_step):
        if dev is None or needs_resample:
          if dev is not None and hasattr(dev, 'close'):
            dev.close()
          if pool:
            batch = pending.result()
            pending = pool.submit(next, dataset_iter) if i + 1 < max_training_step else None
          else:
            batch = next(dataset_iter)
          dev = make_device(batch)
        model_now = unflatten(x)   # (one tree per step: it is evaluated, then kept as the last finite parameters -- nothing writes into it)
        current_loss, grads = loss_and_grad(model_now, dev)
        if np.isnan(current_loss) and i == 0:
          raise ValueError(f'Encountered NaN in loss function. current_loss = {current_loss}, grads = {grads}.')
        if np.isfinite(current_loss):
          params.model = model_now
        else:
          break
        gvec, _ = lbfgs_lib.tree_flatten(grads)
        x = opt.step(x, gvec)
        if callback:
          callback(i, params.model, current_loss)
    finally:
      if pool:
        pool.shutdown(wait=True)
      if resident is not None:
        resident.close()
    if dev is not None:
      final_loss, _ = loss_and_grad(unflatten(x), dev)
      if np.isfinite(final_loss):
        params.model = unflatten(x)
      if hasattr(dev, 'close'):
        dev.close()
  elif method == 'lbfgs':
    batch = next(data_utils.sub_sample_dataset_iterator(rng, dataset, batch_size))   # gp.py:102-107
    dev = make_device(batch)
    alpha = params.config.get('alpha', 1.0)
    _, params.model, _ = lbfgs_lib.lbfgs(None, params.model, steps=max_training_step, alpha=alpha,
                                         val_and_grad_fn=lambda mp: loss_and_grad(mp, dev), callback=callback)
    if hasattr(dev, 'close'):
      dev.close()
  else:
    raise ValueError(f'Optimization method {method} is not supported.')
  params.cache = {}
  return params


def _predict_native(mean_func, cov_func, params, x_query, warp_func, full_cov, handle, input_dim):
  xq = np.asarray(x_query)
  dtype = handle.dtype if handle is not None else _model.infer_dtype(xq)
  xq = np.ascontiguousarray(xq, dtype=dtype)
  nq = xq.shape[0]
  mu = np.empty((nq, 1), dtype=dtype)
  cov = np.empty((nq, nq) if full_cov else (nq, 1), dtype=dtype)
  if nq == 0:
    return mu, cov
  bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, dtype, input_dim)
  ctx = handle.ctx if handle is not None else nat.default_context()
  ctx.check(nat.lib().hbo_predict(ctx.handle, bm.ref(), handle.handle if handle is not None else None,
                                  nat.ptr(xq), nq, int(full_cov), nat.ptr(mu), nat.ptr(cov)))
  return mu, cov


def predict(mean_func, cov_func, params, x_observed, y_observed, x_query, warp_func=None, full_cov=False,
            cache=None):
  """Posterior mean (n',1) and covariance (n',n') / variance (n',1) at x_query (gp.py:242-305)."""
  x_query = np.asarray(x_query)
  if x_observed is None or np.asarray(x_observed).shape[0] == 0:
    return _predict_native(mean_func, cov_func, params, x_query, warp_func, full_cov, None,
                           x_query.shape[1])
  handle = getattr(cache, 'handle', None) if cache is not None else None
  owned = False
  if handle is None:
    handle = linalg.factor(mean_func, cov_func, params, x_observed, y_observed, warp_func)
    owned = True
  try:
    return _predict_native(mean_func, cov_func, params, x_query, warp_func, full_cov, handle,
                           np.asarray(x_observed).shape[1])
  finally:
    if owned:
      handle.close()


class GP:
  """A Gaussian process that supports learning with historical data (gp.py:308-620)."""
  dataset: Dict[Union[int, str], SubDataset]

  def __init__(self, dataset, mean_func: Callable[..., np.ndarray], cov_func: Callable[..., np.ndarray],
               params: GPParams, warp_func=None):
    self.mean_func = mean_func
    self.cov_func = cov_func
    self.params = params if params is not None else GPParams()
    self.warp_func = warp_func
    self.set_dataset(dataset)
    if 'objective' not in self.params.config:
      self.params.config['objective'] = obj.neg_log_marginal_likelihood
    self.rng = None

  def initialize_params(self, key):
    """gp.py:346-401 with a numpy Generator (or int seed) in place of a JAX PRNG key."""
    if not self.dataset:
      raise ValueError('Cannot initialize GPParams without dataset.')
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    if isinstance(self.params.config['objective'], str):
      self.params.config['objective'] = getattr(obj, self.params.config['objective'])
    model = self.params.model
    if 'mlp' in self.mean_func.__name__ or 'mlp' in self.cov_func.__name__:
      if not isinstance(self.params.config.get('mlp_features'), tuple):
        self.params.config['mlp_features'] = (2 * self.input_dim,)
      last_layer_size = self.params.config['mlp_features'][-1]
      if not isinstance(model.get('mlp_params'), dict):
        fin = self.input_dim
        mlp = {}
        for l, f in enumerate(self.params.config['mlp_features']):
          # flax Dense default: lecun_normal kernel, zero bias
          mlp[f'Dense_{l}'] = {'kernel': rng.normal(size=(fin, f)) / np.sqrt(fin), 'bias': np.zeros((f,))}
          fin = f
        model['mlp_params'] = mlp
    else:
      last_layer_size = self.input_dim
    if 'linear' in self.mean_func.__name__:
      if not isinstance(model.get('linear_mean'), dict):
        fin = last_layer_size if 'mlp' in self.mean_func.__name__ else self.input_dim
        model['linear_mean'] = {'kernel': rng.normal(size=(fin, 1)) / np.sqrt(fin), 'bias': np.zeros((1,))}
    if isinstance(model.get('lengthscale'), float):
      if 'mlp' not in self.cov_func.__name__:
        last_layer_size = self.input_dim
      model['lengthscale'] = np.ones(last_layer_size) * model['lengthscale']
    self.rng = rng

  def set_dataset(self, dataset):
    """Reset GP dataset (gp.py:403-419); clears the cache."""
    self.dataset = {}
    self._drop_cache()
    self._drop_sample_caches()
    if isinstance(dataset, list):
      dataset = {i: dataset[i] for i in range(len(dataset))}
    for key, val in dataset.items():
      self.dataset[key] = SubDataset(*val)

  def _drop_sample_caches(self):
    """Per-sample factorisations an HGP acquisition gradient keeps (bo_utils/acfun.py): released when the observations change
    or the model is re-trained; update_model_params keeps them -- they are keyed by the SAMPLES' content, not by params.model."""
    if getattr(self, '_hbo_sample_caches', None) is not None:
      from hyperbo_amd.bo_utils import acfun as _acfun
      _acfun.drop_sample_caches(self)

  def _drop_cache(self):
    for c in getattr(self.params, 'cache', {}).values():
      h = getattr(c, 'handle', None)
      if h is not None:
        h.close()
    self.params.cache = {}

  @property
  def input_dim(self) -> int:
    key = list(self.dataset.keys())[0]
    return self.dataset[key].x.shape[1]

  def update_sub_dataset(self, sub_dataset, sub_dataset_key: Union[int, str] = 0, is_append: bool = False):
    """gp.py:426-452."""
    sub_dataset = SubDataset(*sub_dataset)
    if is_append:
      if sub_dataset_key not in self.dataset:
        assert self.dataset, 'dataset cannot be empty.'
        self.dataset[sub_dataset_key] = SubDataset(x=np.empty((0, self.input_dim)), y=np.empty((0, 1)))
      new_x = np.vstack((self.dataset[sub_dataset_key].x, sub_dataset.x))
      new_y = np.vstack((self.dataset[sub_dataset_key].y, sub_dataset.y))
      self.dataset[sub_dataset_key] = SubDataset(x=new_x, y=new_y)
    else:
      self.dataset[sub_dataset_key] = sub_dataset
    self._drop_sample_caches()
    if sub_dataset_key in self.params.cache:
      cache = self.params.cache[sub_dataset_key]
      cache.needs_update = True
      # remember what was appended so that setup_predictor can update the factor in O(N^2)
      pending = getattr(cache, 'pending', None)
      if is_append and pending is not None:
        pending.append((np.asarray(sub_dataset.x), np.asarray(sub_dataset.y)))
      elif hasattr(cache, 'pending'):
        cache.pending = None

  def train(self, key=None, get_params_path=None, callback=None) -> GPParams:
    """Fit the GP hyper-parameters to the dataset (gp.py:454-485)."""
    if key is None:
      if self.rng is None:
        self.rng = np.random.default_rng(0)
      key = self.rng
    self._drop_cache()
    self._drop_sample_caches()
    self.params = infer_parameters(
        mean_func=self.mean_func, cov_func=self.cov_func, init_params=self.params, dataset=self.dataset,
        warp_func=self.warp_func, objective=self.params.config['objective'], key=key,
        get_params_path=get_params_path, callback=callback)
    return self.params

  def neg_log_marginal_likelihood(self):
    """Total nll and key->nll dict with the SVD variant, as the reference (gp.py:487-497): device Gram + mean,
    host LAPACK SVD -- finite where the Cholesky variant would report NaN for a numerically low-rank covariance."""
    return obj.neg_log_marginal_likelihood(
        mean_func=self.mean_func, cov_func=self.cov_func, params=self.params, dataset=self.dataset,
        warp_func=self.warp_func, return_key2nll=True, use_cholesky=False)

  def empirical_divergence(self, distance=None) -> float:
    """Empirical divergence between sample mean / covariance and the model (gp.py:499-510)."""
    return obj.multivariate_normal_divergence(
        mean_func=self.mean_func, cov_func=self.cov_func, params=self.params, dataset=self.dataset,
        warp_func=self.warp_func, distance=distance)

  def stats(self, verbose=True):
    """(nll, ekl, ekl_partial, euc, key2nll) of the current model (gp.py:512-533)."""
    import functools
    from hyperbo_amd.gp_utils import utils as _utils
    nll, key2nll = self.neg_log_marginal_likelihood()
    ekl = self.empirical_divergence(functools.partial(_utils.kl_multivariate_normal, eps=1e-6, partial=False))
    ekl_partial = self.empirical_divergence(functools.partial(_utils.kl_multivariate_normal, eps=1e-6, partial=True))
    euc = self.empirical_divergence(_utils.euclidean_multivariate_normal)
    if verbose:
      print(f'nll = {nll}, ekl = {ekl}, ekl_partial = {ekl_partial}, euc = {euc}')
    return nll, ekl, ekl_partial, euc, key2nll

  def update_model_params(self, model_params: Dict[str, Any]):
    """gp.py:535-538."""
    self.params.model = model_params
    self._drop_cache()

  def has_observations(self, sub_dataset_key) -> bool:
    """True when the sub-dataset exists and is non-empty; an empty one predicts from the prior (gp.py:275-282)."""
    return sub_dataset_key in self.dataset and np.shape(self.dataset[sub_dataset_key].x)[0] > 0

  def setup_predictor(self, sub_dataset_key: Union[int, str] = 0):
    """gp.py:540-560."""
    if not self.has_observations(sub_dataset_key):
      return   # nothing to factorise: predict() takes the prior branch
    if sub_dataset_key in self.params.cache and not self.params.cache[sub_dataset_key].needs_update:
      return
    old = self.params.cache.get(sub_dataset_key)
    if (old is not None and getattr(old, 'handle', None) is not None and getattr(old, 'pending', None)
        and self.params.config.get('incremental_cache', True)):
      # rows were only appended since the factorisation: O(N^2) update instead of the reference's
      # O(N^3) re-factorisation (same result up to rounding; gp.py:284 anticipates it)
      ok = True
      for xa, ya in old.pending:
        ok = ok and old.handle.append(self.params, xa, ya)
        if not ok:
          break
      if ok:
        old.pending = []
        old.needs_update = False
        old.invalidate_arrays()
        return
    if old is not None and getattr(old, 'handle', None) is not None:
      old.handle.close()
    sd = self.dataset[sub_dataset_key]
    handle = linalg.factor(self.mean_func, self.cov_func, self.params, sd.x, sd.y, self.warp_func)
    self.params.cache[sub_dataset_key] = GPCache(needs_update=False, handle=handle)   # chol/kinvy exported lazily

  def predict(self, queried_inputs, sub_dataset_key: Union[int, str] = 0, full_cov: bool = False,
              with_noise: bool = True, unbiased: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """gp.py:562-620."""
    if not self.has_observations(sub_dataset_key):
      mu, cov = predict(self.mean_func, self.cov_func, self.params, None, None, queried_inputs,
                        warp_func=self.warp_func, full_cov=full_cov)
    else:
      self.setup_predictor(sub_dataset_key)
      sd = self.dataset[sub_dataset_key]
      mu, cov = predict(self.mean_func, self.cov_func, self.params, sd.x, sd.y, queried_inputs,
                        warp_func=self.warp_func, full_cov=full_cov,
                        cache=self.params.cache[sub_dataset_key])
    add_noise, scale = self.predict_noise_and_scale(with_noise, unbiased)
    if with_noise:
      if full_cov:
        cov = cov + np.eye(cov.shape[0], dtype=cov.dtype) * cov.dtype.type(add_noise)
      else:
        cov = cov + cov.dtype.type(add_noise)
    if scale != 1.0:
      cov = cov * cov.dtype.type(scale)
    return mu, cov

  def predict_noise_and_scale(self, with_noise=True, unbiased=True):
    """The (noise_variance, T/(T-1)) post-processing constants of gp.py:607-619."""
    add_noise = 0.0
    if with_noise:
      nv, = retrieve_params(self.params, ['noise_variance'], warp_func=self.warp_func)
      add_noise = float(np.squeeze(nv))
    scale = 1.0
    if unbiased:
      len_dataset = len([k for k, v in self.dataset.items() if v.aligned is None])
      if len_dataset > 1:
        scale = len_dataset / (len_dataset - 1.)
    return add_noise, scale


class HGP(GP):
  """Hierarchical GP: predictions for every sample of model params (gp.py:623-682)."""

  def get_model_params_samples(self):
    return self.params.samples if self.params.samples else [self.params.model]

  def stats(self, verbose=True):
    """Mean over the parameter samples of GP.stats (gp.py:633-664)."""
    samples = self.get_model_params_samples()
    rows, acc, key2nll = [], {}, {}
    for model_params in samples:
      self.update_model_params(model_params)
      nll, ekl, ekl_partial, euc, key2nll = super().stats(verbose=False)
      rows.append((nll, ekl, ekl_partial, euc))
      for k, v in key2nll.items():
        acc[k] = acc.get(k, 0.) + v
    for k in key2nll:
      acc[k] /= len(samples)
    nll, ekl, ekl_partial, euc = np.mean(np.asarray(rows, dtype=np.float64), axis=0)
    if verbose:
      print(f'HGP nll = {nll}, ekl = {ekl}, ekl_partial = {ekl_partial}, euc = {euc}')
    return nll, ekl, ekl_partial, euc, acc

  def predict(self, queried_inputs, sub_dataset_key=0, full_cov=False, with_noise=True):
    results = []
    for model_params in self.get_model_params_samples():
      self.update_model_params(model_params)
      results.append(super().predict(queried_inputs, sub_dataset_key, full_cov, with_noise))
    return results
