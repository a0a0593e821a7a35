"""Training objectives -- hyperbo/gp_utils/objectives.py:109-210 on the GPU.

`neg_log_marginal_likelihood` keeps the reference signature.  Because there is no autodiff, the
companion `nll_value_and_grad` returns what `jax.value_and_grad(loss_func)` returns at
hyperbo/gp_utils/gp.py:134 / hyperbo/basics/lbfgs.py:238: (scalar, pytree shaped like params.model).
`DeviceDataset` keeps the sub-datasets resident in HBM across evaluations (only theta changes).
"""
import ctypes as C
import logging

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import params_utils

retrieve_params = params_utils.retrieve_params


def included_sub_datasets(dataset, exclude_aligned=True):
  """Selection rule of objectives.py:181-185: skip aligned (if asked) and empty sub-datasets."""
  out = []
  for k, s in dataset.items():
    if exclude_aligned and s.aligned is not None:
      continue
    if s.x.shape[0] == 0:
      continue
    out.append((k, s))
  return out


class DeviceDataset:
  """Sub-datasets uploaded once (hbo_dataset); tasks are independent given theta."""

  def __init__(self, dataset, exclude_aligned=True, dtype=None, ctx=None, keys=None):
    self.ctx = ctx or nat.default_context()
    items = included_sub_datasets(dataset, exclude_aligned)
    if keys is not None:
      keyset = set(keys)
      items = [(k, s) for k, s in items if k in keyset]
    self.keys = [k for k, _ in items]
    if dtype is None:
      dtype = _model.infer_dtype(*[a for _, s in items for a in (s.x, s.y)]) if items else np.float64
    self.dtype = np.dtype(dtype)
    self._xs = [np.ascontiguousarray(np.asarray(s.x), dtype=self.dtype) for _, s in items]
    self._ys = [np.ascontiguousarray(np.asarray(s.y), dtype=self.dtype) for _, s in items]
    for x, y in zip(self._xs, self._ys):
      if y.ndim != 2 or y.shape[0] != x.shape[0] or y.shape[1] == 0:
        raise ValueError(f'sub-dataset x has shape {x.shape} but y has shape {y.shape}')
    self.input_dim = self._xs[0].shape[1] if self._xs else 1
    self.num_tasks = len(items)
    self._h = C.c_void_p()
    # device order: largest task first (stable) -- mirrors hbo_dataset_create
    order = sorted(range(len(items)), key=lambda i: -self._xs[i].shape[0])
    self.device_order_keys = [self.keys[i] for i in order]
    if items:
      tasks = (nat.Task * len(items))()
      for i, (x, y) in enumerate(zip(self._xs, self._ys)):
        tasks[i].x, tasks[i].y = nat.ptr(x).value, nat.ptr(y).value
        tasks[i].n, tasks[i].m = x.shape[0], y.shape[1]
      self.ctx.check(nat.lib().hbo_dataset_create(self.ctx.handle, nat.dtype_code(self.dtype), self.input_dim,
                                                  tasks, len(items), C.byref(self._h)), allow_not_pd=False)

  def close(self):
    if self._h:
      nat.lib().hbo_dataset_free(self.ctx.handle, self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def evaluate(self, mean_func, cov_func, params, warp_func=None, want_grad=False, per_task=False):
    """Returns (nll_sum, per_task dict or None, flat grad_sum (warped) or None, BuiltModel)."""
    bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, self.dtype, self.input_dim)
    nll = C.c_double(0.0)
    pt = (C.c_double * max(self.num_tasks, 1))() if per_task else None
    g = (C.c_double * max(bm.layout.total, 1))() if want_grad else None
    if self.num_tasks:
      self.ctx.check(nat.lib().hbo_nll(self.ctx.handle, bm.ref(), self._h, C.byref(nll), pt, g))
    key2nll = None
    if per_task:
      vals = dict(zip(self.device_order_keys, list(pt)[:self.num_tasks]))
      key2nll = {k: vals[k] for k in self.keys}
    grad = np.array(list(g)[:bm.layout.total], dtype=np.float64) if want_grad else None
    return nll.value, key2nll, grad, bm


def _as_device(dataset, exclude_aligned, ctx=None):
  if isinstance(dataset, DeviceDataset):
    return dataset, False
  return DeviceDataset(dataset, exclude_aligned=exclude_aligned, ctx=ctx), True


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


def neg_log_marginal_likelihood(mean_func, cov_func, params, dataset, warp_func=None, exclude_aligned=True,
                                return_key2nll=False, use_cholesky=True):
  """Negative log marginal likelihood of a (multi-task) GP: mean over included sub-datasets."""
  if not use_cholesky:
    raise NotImplementedError('the SVD variant (objectives.py:157-176) is a CPU reporting path; '
                              'it is restated in oracle/ only')
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
  try:
    nll_sum, _, grad, bm = dev.evaluate(mean_func, cov_func, params, warp_func, want_grad=True)
    count = float(dev.num_tasks)
  finally:
    if owned:
      dev.close()
  if comm is not None:
    buf = np.concatenate([[nll_sum, count], grad])
    buf = comm.allreduce_sum(buf)
    nll_sum, count, grad = buf[0], buf[1], buf[2:]
  if count > 0:
    value, grad = nll_sum / count, grad / count
  else:
    value, grad = 0., grad * 0.
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


nll = neg_log_marginal_likelihood


def add(*objectives):
  def added_objective(*args, **kwargs):
    return sum([obj(*args, **kwargs) for obj in objectives])
  return added_objective


def mul(c, obj):
  def multiplied_objective(*args, **kwargs):
    return c * obj(*args, **kwargs)
  return multiplied_objective
