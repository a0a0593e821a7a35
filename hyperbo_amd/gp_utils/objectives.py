"""Training objectives -- hyperbo/gp_utils/objectives.py:29-247 on the GPU.

`neg_log_marginal_likelihood`, `multivariate_normal_divergence` (kl / ekl) and
`multivariate_normal_euc_distance` (euc) keep the reference signatures.  Because there is no autodiff,
every objective carries a `.value_and_grad` companion that returns what `jax.value_and_grad(loss_func)`
returns at hyperbo/gp_utils/gp.py:134 / hyperbo/basics/lbfgs.py:238: (scalar, pytree shaped like
params.model); `add` / `mul` (objectives.py:221-247) compose the companions too.
`DeviceDataset` keeps the sub-datasets resident in HBM across evaluations (only theta changes);
`DeviceBatch` holds the i.i.d. and the aligned selection of one training batch.
"""
import ctypes as C
import logging

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import params_utils

retrieve_params = params_utils.retrieve_params


OBJ_NLL, OBJ_EKL, OBJ_EUC = 0, 1, 2   # include/hbo.h hbo_objective_id


def included_sub_datasets(dataset, exclude_aligned=True, only_aligned=False):
  """Selection rules: objectives.py:181-185 (NLL: skip aligned if asked, skip empty) and
  objectives.py:85-96 (divergences: aligned and non-empty only)."""
  out = []
  for k, s in dataset.items():
    if only_aligned:
      if s.aligned is None:
        continue
    elif exclude_aligned and s.aligned is not None:
      continue
    if s.x.shape[0] == 0:
      continue
    out.append((k, s))
  return out


class DeviceDataset:
  """Sub-datasets uploaded once (hbo_dataset); tasks are independent given theta."""

  def __init__(self, dataset, exclude_aligned=True, dtype=None, ctx=None, keys=None, only_aligned=False):
    self.ctx = ctx or nat.default_context()
    self.only_aligned = only_aligned
    items = included_sub_datasets(dataset, exclude_aligned, only_aligned)
    if keys is not None:
      keyset = set(keys)
      items = [(k, s) for k, s in items if k in keyset]
    self.keys = [k for k, _ in items]
    if dtype is None:
      dtype = _model.infer_dtype(*[a for _, s in items for a in (s.x, s.y)]) if items else np.float64
    self.dtype = np.dtype(dtype)
    self._xs = [np.ascontiguousarray(np.asarray(s.x), dtype=self.dtype) for _, s in items]
    self._ys = [np.ascontiguousarray(np.asarray(s.y), dtype=self.dtype) for _, s in items]
    for (k, _), x, y in zip(items, self._xs, self._ys):
      if y.ndim != 2 or y.shape[0] != x.shape[0] or y.shape[1] == 0:
        raise ValueError(f'dataset[{k}].x has shape {x.shape} but dataset[{k}].y has shape {y.shape}')
    if self._xs:
      self.input_dim = self._xs[0].shape[1]
    else:   # nothing selected: keep the dataset's input dimension so that params still validate
      dims = [np.shape(s.x)[1] for s in dataset.values() if np.ndim(s.x) == 2]
      self.input_dim = dims[0] if dims else 1
    self.num_tasks = len(items)
    self._sizes = {k: x.shape[0] for (k, _), x in zip(items, self._xs)}
    self._h = C.c_void_p()
    # device order: largest task first (stable) -- mirrors hbo_dataset_create
    order = sorted(range(len(items)), key=lambda i: -self._xs[i].shape[0])
    self.device_order_keys = [self.keys[i] for i in order]
    if items:
      tasks = (nat.Task * len(items))()
      for i, (x, y) in enumerate(zip(self._xs, self._ys)):
        tasks[i].x, tasks[i].y = x.ctypes.data, y.ctypes.data   # (the address as an int: data_as() + cast cost 3 us per array)
        tasks[i].n, tasks[i].m = x.shape[0], y.shape[1]
      self.ctx.check(nat.lib().hbo_dataset_create(self.ctx.handle, nat.dtype_code(self.dtype), self.input_dim,
                                                  tasks, len(items), C.byref(self._h)), allow_not_pd=False)

  def subsample(self, index_map):
    """A new DeviceDataset holding, for every key of this one, the rows index_map[key] (int array, in that order) -- or the whole
    sub-dataset when the key is missing / None.  Gathered on the device from the resident inputs (hbo_dataset_subsample): only
    the indices travel.  The per-step batch of infer_parameters' Adam loop (gp.py:101-111)."""
    new = DeviceDataset.__new__(DeviceDataset)
    new.ctx, new.only_aligned, new.keys, new.dtype, new.input_dim = self.ctx, self.only_aligned, list(self.keys), self.dtype, self.input_dim
    new._xs, new._ys = [], []
    new.num_tasks = self.num_tasks
    new._h = C.c_void_p()
    order = self.device_order_keys
    ixs = [index_map.get(k) for k in order]
    counts = np.fromiter((-1 if ix is None else len(ix) for ix in ixs), dtype=np.int64, count=len(order))
    parts = [ix for ix in ixs if ix is not None]
    sizes = {k: (self._sizes[k] if ix is None else len(ix)) for k, ix in zip(order, ixs)}
    new._sizes = sizes
    # the library keeps the tasks largest first, stable in the order it was given them: mirror it
    vals = [sizes[k] for k in order]
    new.device_order_keys = list(order) if all(a >= b for a, b in zip(vals, vals[1:])) else \
        [order[i] for i in sorted(range(len(order)), key=lambda i: -vals[i])]
    if order:
      idx = np.ascontiguousarray(np.concatenate(parts), dtype=np.int32) if parts else np.zeros(1, dtype=np.int32)
      self.ctx.check(nat.lib().hbo_dataset_subsample(self.ctx.handle, self._h, counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     idx.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(new._h)), allow_not_pd=False)
    return new

  def close(self):
    if self._h:
      nat.lib().hbo_dataset_free(self.ctx.handle, self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def evaluate_sharded(self, mean_func, cov_func, params, warp_func=None, objective=OBJ_NLL, comm=None):
    """This rank's shard through hbo_objective_sharded: the sums over ALL ranks of `comm`'s communicator, reduced on the device
    and all-reduced in place (RCCL over xGMI).  Returns (value_sum, task_count, flat grad_sum (warped), BuiltModel)."""
    input_dim = self.input_dim
    if self.num_tasks == 0:
      input_dim = _model.infer_input_dim(mean_func, cov_func, params) or self.input_dim
    bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, self.dtype, input_dim)
    val, cnt = C.c_double(0.0), C.c_double(0.0)
    g = (C.c_double * max(bm.layout.total, 1))()
    timing = (C.c_double * 2)()
    self.ctx.check(nat.lib().hbo_objective_sharded(self.ctx.handle, bm.ref(), self._h if self.num_tasks else None, objective,
                                                   C.byref(val), C.byref(cnt), g, timing))
    if comm is not None:
      comm.last_timing = (timing[0], timing[1])
    return val.value, cnt.value, np.array(list(g)[:bm.layout.total], dtype=np.float64), bm

  def evaluate(self, mean_func, cov_func, params, warp_func=None, want_grad=False, per_task=False,
               objective=OBJ_NLL):
    """Returns (value_sum, per_task dict or None, flat grad_sum (warped) or None, BuiltModel)."""
    input_dim = self.input_dim
    if self.num_tasks == 0:
      # an empty selection (e.g. a task shard of a rank beyond the task count) carries no inputs: take D from the
      # parameters so that this rank builds the same model -- and the same gradient layout -- as its peers and
      # reaches the all-reduce with zeros instead of raising on a shape mismatch
      input_dim = _model.infer_input_dim(mean_func, cov_func, params) or self.input_dim
    bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, self.dtype, input_dim)
    nll = C.c_double(0.0)
    pt = (C.c_double * max(self.num_tasks, 1))() if per_task else None
    g = (C.c_double * max(bm.layout.total, 1))() if want_grad else None
    if self.num_tasks:
      self.ctx.check(nat.lib().hbo_objective(self.ctx.handle, bm.ref(), self._h, objective, C.byref(nll), pt, g))
    key2nll = None
    if per_task:
      vals = dict(zip(self.device_order_keys, list(pt)[:self.num_tasks]))
      key2nll = {k: vals[k] for k in self.keys}
    grad = np.frombuffer(g, dtype=np.float64, count=bm.layout.total).copy() if want_grad else None
    return nll.value, key2nll, grad, bm


class DeviceBatch:
  """One training batch resident in HBM: lazily uploaded selections of the same dataset dict --
  'iid' (aligned excluded, the NLL's default), 'all' (NLL with exclude_aligned=False) and 'aligned'
  (the divergence objectives)."""

  def __init__(self, dataset, ctx=None):
    self.dataset = dataset
    self.ctx = ctx
    self._dev = {}

  def get(self, selection):
    if selection not in self._dev:
      self._dev[selection] = DeviceDataset(self.dataset, exclude_aligned=(selection == 'iid'),
                                           only_aligned=(selection == 'aligned'), ctx=self.ctx)
    return self._dev[selection]

  def subsample(self, index_map):
    """A batch whose selections are device-side sub-samples of this (resident) one."""
    return _SubsampledBatch(self, index_map)

  def close(self):
    for d in self._dev.values():
      d.close()
    self._dev = {}


class _SubsampledBatch(DeviceBatch):
  def __init__(self, parent, index_map):   # pylint: disable=super-init-not-called
    self.parent, self.index_map = parent, index_map
    self.dataset, self.ctx, self._dev = parent.dataset, parent.ctx, {}

  def get(self, selection):
    if selection not in self._dev:
      self._dev[selection] = self.parent.get(selection).subsample(self.index_map)
    return self._dev[selection]


def _as_device(dataset, exclude_aligned, ctx=None, only_aligned=False):
  if isinstance(dataset, DeviceDataset):
    if dataset.only_aligned != only_aligned:
      raise ValueError('DeviceDataset was built with a different sub-dataset selection than this objective needs')
    return dataset, False
  if isinstance(dataset, DeviceBatch):
    return dataset.get('aligned' if only_aligned else ('iid' if exclude_aligned else 'all')), False
  return DeviceDataset(dataset, exclude_aligned=exclude_aligned, ctx=ctx, only_aligned=only_aligned), True


def _apply_priors(total_nll, params, warp_func):
  # objectives.py:198-207
  if 'priors' in params.config:
    for k in params.model:
      if k in params.config['priors']:
        log_prior_fn = params.config['priors'][k]
        val, = retrieve_params(params, [k], warp_func)
        total_nll -= float(log_prior_fn(val))
      else:
        logging.warning('No prior provided for param %s', k)
  return total_nll


SVD_WARN_N = 4096   # _nll_sub_dataset_svd logs its cost from this many points on


def _nll_sub_dataset_svd(mean_func, cov_func, params, vx, vy, warp_func):
  """objectives.py:157-176: 0.5 * sum(y^T K^-1 y + sum(log s) + n log 2 pi) with K^-1 = V^T diag(1/s) U^T from the SVD
  of the jittered Gram matrix.  Mean and Gram come from the device (hbo_mean / hbo_gram through mean_func / cov_func);
  the SVD itself is host LAPACK -- the reference's reporting path for covariances that are numerically low rank, where
  the Cholesky variant returns NaN.  For y of shape (n, m > 1) the (m, m) + scalar broadcast of the reference is kept.

  Cost: the n x n Gram matrix crosses PCIe (8 n^2 bytes) and the SVD is O(n^3) float64 on the host -- seconds at n = 4096,
  minutes and > 6 GB of host memory at n = 16384, per sub-dataset (and per parameter sample under HGP.stats).  A warning is
  logged from SVD_WARN_N points on; neg_log_marginal_likelihood(use_cholesky=True) is the device path."""
  from hyperbo_amd.basics import linalg
  if len(vx) >= SVD_WARN_N:
    logging.warning('SVD variant of the NLL on %d points: the %d x %d Gram matrix is copied to the host and decomposed by LAPACK '
                    '(O(n^3) float64, %.1f GB); use_cholesky=True runs on the GPU', len(vx), len(vx), len(vx), 3 * 8e-9 * len(vx) ** 2)
  vy, cov = linalg.compute_delta_y_and_cov(mean_func, cov_func, params, vx, vy, warp_func=warp_func)
  u, s, vt = np.linalg.svd(np.asarray(cov, dtype=np.float64))
  if s[-1] <= 0:
    logging.warning('Covariance matrix is low rank. s = %s', s)
  with np.errstate(divide='ignore', invalid='ignore'):
    kinvy = vt.T @ ((u.T @ np.asarray(vy, dtype=np.float64)) / s[:, None])
    return float(0.5 * np.sum(vy.T @ kinvy + np.sum(np.log(s)) + len(vx) * np.log(2 * np.pi)))


def neg_log_marginal_likelihood(mean_func, cov_func, params, dataset, warp_func=None, exclude_aligned=True,
                                return_key2nll=False, use_cholesky=True):
  """Negative log marginal likelihood of a (multi-task) GP: mean over included sub-datasets
  (objectives.py:109-210).  use_cholesky=False is the SVD variant (objectives.py:157-176)."""
  if not use_cholesky:
    if isinstance(dataset, (DeviceDataset, DeviceBatch)):
      raise TypeError('use_cholesky=False takes the host dataset dict (the SVD runs on host LAPACK)')
    key2nll = {k: _nll_sub_dataset_svd(mean_func, cov_func, params, s.x, s.y, warp_func)
               for k, s in included_sub_datasets(dataset, exclude_aligned)}
    total = sum(key2nll.values()) / len(key2nll) if key2nll else 0.
    total = _apply_priors(total, params, warp_func)
    return (total, key2nll) if return_key2nll else total
  dev, owned = _as_device(dataset, exclude_aligned)
  try:
    nll_sum, key2nll, _, _ = dev.evaluate(mean_func, cov_func, params, warp_func, per_task=return_key2nll)
    total = 0. if dev.num_tasks == 0 else nll_sum / dev.num_tasks
  finally:
    if owned:
      dev.close()
  total = _apply_priors(total, params, warp_func)
  if return_key2nll:
    return total, (key2nll or {})
  return total


def nll_value_and_grad(mean_func, cov_func, params, dataset, warp_func=None, exclude_aligned=True,
                       comm=None):
  """(value, grads) of neg_log_marginal_likelihood w.r.t. params.model (raw, un-warped values).

  `comm`: optional hyperbo_amd.parallel communicator -- `dataset` then holds this rank's task shard
  and [nll_sum, count, grad_sum] are sum-all-reduced before the mean over tasks is taken.
  """
  dev, owned = _as_device(dataset, exclude_aligned)
  native = comm is not None and getattr(comm, 'native_sharded', False)
  try:
    if native:
      nll_sum, count, grad, bm = dev.evaluate_sharded(mean_func, cov_func, params, warp_func, comm=comm)
    else:
      nll_sum, _, grad, bm = dev.evaluate(mean_func, cov_func, params, warp_func, want_grad=True)
      count = float(dev.num_tasks)
  finally:
    if owned:
      dev.close()
  if comm is not None and not native:
    buf = np.concatenate([[nll_sum, count], grad])
    buf = comm.allreduce_sum(buf)
    nll_sum, count, grad = buf[0], buf[1], buf[2:]
  if count <= 0:
    value, grad = 0., grad * 0.
  else:   # (a NaN count -- a peer that failed locally contributes NaN to every slot -- takes this branch: the objective is NaN)
    value, grad = nll_sum / count, grad / count
  grads = bm.unflatten_grad(grad)
  if 'priors' in params.config:
    value = _apply_priors(value, params, warp_func)
    from hyperbo_amd.gp_utils import priors as _priors
    from hyperbo_amd.gp_utils import utils as _utils
    wf = warp_func or {}
    for k in params.model:
      fn = params.config['priors'].get(k)
      if fn is None:
        continue
      dfn = _priors.gradient_of(fn)
      val, = retrieve_params(params, [k], warp_func)
      raw = np.asarray(params.model[k], dtype=np.float64)
      dwarp = _utils.warp_derivative(wf[k], raw) if k in wf else np.ones_like(raw)
      grads[k] = grads[k] - np.reshape(dfn(val), raw.shape) * dwarp
  return value, grads


neg_log_marginal_likelihood.value_and_grad = nll_value_and_grad
nll_value_and_grad.accepts_device_batch = True


def _divergence(objective_id, mean_func, cov_func, params, dataset, warp_func, want_grad, comm=None):
  dev, owned = _as_device(dataset, True, only_aligned=True)
  native = want_grad and comm is not None and getattr(comm, 'native_sharded', False)
  try:
    if native:
      total, count, grad, bm = dev.evaluate_sharded(mean_func, cov_func, params, warp_func, objective=objective_id, comm=comm)
    else:
      total, _, grad, bm = dev.evaluate(mean_func, cov_func, params, warp_func, want_grad=want_grad,
                                        objective=objective_id)
      count = float(dev.num_tasks)
  finally:
    if owned:
      dev.close()
  if not want_grad:
    return (0. if count == 0 else total / count), None
  if comm is not None and not native:
    buf = comm.allreduce_sum(np.concatenate([[total, count], grad]))
    total, count, grad = buf[0], buf[1], buf[2:]
  if count <= 0:
    return 0., bm.unflatten_grad(grad * 0.)
  return total / count, bm.unflatten_grad(grad / count)   # (incl. a NaN count: a peer failed locally)


def _divergence_host(mean_func, cov_func, params, dataset, warp_func, distance):
  """objectives.py:53-101 for an arbitrary `distance` callable (e.g. functools.partial(utils.kl_multivariate_normal,
  eps=1e-6, partial=False) in GP.stats, gp.py:512-533): sample statistics on the host, model mean / covariance from
  the device (hbo_mean / hbo_gram), then distance(mu0=, cov0=, mu1=, cov1=) -- the utils distances factorise on the
  device themselves."""
  total, count = 0., 0
  for key, sd in dataset.items():
    if sd.aligned is None or sd.x.shape[0] == 0:
      continue
    if sd.y.shape[1] == 0 or sd.y.shape[0] != sd.x.shape[0]:
      raise ValueError(f'dataset[{key}].x has shape {sd.x.shape} but dataset[{key}].y has shape {sd.y.shape}')
    y = np.asarray(sd.y, dtype=np.float64)
    mu_data = np.mean(y, axis=1)
    yc = y - mu_data[:, None]
    cov_data = yc @ yc.T / y.shape[1]                     # jnp.cov(y, bias=True)
    mu_model = np.asarray(mean_func(params, sd.x, warp_func=warp_func), dtype=np.float64).flatten()
    noise_variance, = retrieve_params(params, ['noise_variance'], warp_func=warp_func)
    cov_model = np.asarray(cov_func(params, sd.x, warp_func=warp_func), dtype=np.float64) \
        + np.eye(sd.x.shape[0]) * float(np.squeeze(noise_variance))
    total += float(distance(mu0=mu_data, cov0=cov_data, mu1=mu_model, cov1=cov_model))
    count += 1
  return total / count if count else 0.


def multivariate_normal_divergence(mean_func, cov_func, params, dataset, warp_func=None, distance=None):
  """objectives.py:29-101: mean over the aligned sub-datasets of distance(N(mean_a y, cov_a y), GP prior).

  `distance`: None / utils.kl_multivariate_normal (partial KL, utils.py:109-148 defaults) or
  utils.euclidean_multivariate_normal run fused on the device (hbo_objective).  Any other callable -- e.g. the
  functools.partial(kl_multivariate_normal, eps=..., partial=...) of GP.stats -- is evaluated per sub-dataset from
  the device Gram matrix and mean (`_divergence_host`); such a distance has no gradient companion.
  """
  from hyperbo_amd.gp_utils import utils as _utils
  if distance is None or distance is _utils.kl_multivariate_normal:
    oid = OBJ_EKL
  elif distance is _utils.euclidean_multivariate_normal:
    oid = OBJ_EUC
  elif callable(distance) and not isinstance(dataset, (DeviceDataset, DeviceBatch)):
    return _divergence_host(mean_func, cov_func, params, dataset, warp_func, distance)
  else:
    raise NotImplementedError('multivariate_normal_divergence: a custom distance needs the host dataset dict')
  return _divergence(oid, mean_func, cov_func, params, dataset, warp_func, False)[0]


def multivariate_normal_euc_distance(mean_func, cov_func, params, dataset, warp_func=None):
  """objectives.py:104-106."""
  return _divergence(OBJ_EUC, mean_func, cov_func, params, dataset, warp_func, False)[0]


def ekl_value_and_grad(mean_func, cov_func, params, dataset, warp_func=None, comm=None):
  """(value, grads w.r.t. raw params.model) of multivariate_normal_divergence (partial KL)."""
  return _divergence(OBJ_EKL, mean_func, cov_func, params, dataset, warp_func, True, comm)


def euc_value_and_grad(mean_func, cov_func, params, dataset, warp_func=None, comm=None):
  """(value, grads w.r.t. raw params.model) of multivariate_normal_euc_distance."""
  return _divergence(OBJ_EUC, mean_func, cov_func, params, dataset, warp_func, True, comm)


multivariate_normal_divergence.value_and_grad = ekl_value_and_grad
multivariate_normal_euc_distance.value_and_grad = euc_value_and_grad
ekl_value_and_grad.accepts_device_batch = True
euc_value_and_grad.accepts_device_batch = True

nll = neg_log_marginal_likelihood
kl = multivariate_normal_divergence
ekl = kl
euc = multivariate_normal_euc_distance
regkl = kl
regeuc = euc


def _tree_combine(fn, *trees):
  first = trees[0]
  if isinstance(first, dict):
    return {k: _tree_combine(fn, *[t[k] for t in trees]) for k in first}
  if isinstance(first, (list, tuple)):
    return type(first)(_tree_combine(fn, *parts) for parts in zip(*trees))
  return fn(*trees)


def add(*objectives):
  """objectives.py:221-226; the sum also carries the summed value_and_grad when every term has one."""
  def added_objective(*args, **kwargs):
    return sum([obj(*args, **kwargs) for obj in objectives])
  if all(getattr(o, 'value_and_grad', None) is not None for o in objectives):
    def added_value_and_grad(*args, **kwargs):
      pairs = [o.value_and_grad(*args, **kwargs) for o in objectives]
      return sum(v for v, _ in pairs), _tree_combine(lambda *g: sum(np.asarray(x) for x in g), *[g for _, g in pairs])
    added_value_and_grad.accepts_device_batch = all(
        getattr(o.value_and_grad, 'accepts_device_batch', False) for o in objectives)
    added_objective.value_and_grad = added_value_and_grad
  return added_objective


def mul(c, obj):
  """objectives.py:229-234."""
  def multiplied_objective(*args, **kwargs):
    return c * obj(*args, **kwargs)
  if getattr(obj, 'value_and_grad', None) is not None:
    def multiplied_value_and_grad(*args, **kwargs):
      v, g = obj.value_and_grad(*args, **kwargs)
      return c * v, _tree_combine(lambda x: c * np.asarray(x), g)
    multiplied_value_and_grad.accepts_device_batch = getattr(obj.value_and_grad, 'accepts_device_batch', False)
    multiplied_objective.value_and_grad = multiplied_value_and_grad
  return multiplied_objective


# objectives.py:237-247 (the regeuc01 / regeuc10 aliases use regkl there too)
nll_regkl = lambda c: add(nll, mul(c, regkl))
nll_regeuc = lambda c: add(nll, mul(c, regeuc))

nll_regkl1 = nll_regkl(1.)
nll_regeuc1 = nll_regeuc(1.)
nll_regkl01 = nll_regkl(.1)
nll_regeuc01 = nll_regkl(.1)

nll_regkl10 = nll_regkl(10.)
nll_regeuc10 = nll_regkl(10.)
