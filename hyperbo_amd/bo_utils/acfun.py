"""Acquisition functions -- hyperbo/bo_utils/acfun.py:36-185.

For a plain GP the posterior and the EI/PI/UCB epilogue run fused on the GPU (hbo_acq).  For an HGP
(acfun.py:72-82: the mean over the model-parameter samples) all samples are factorised as ONE batch and
their posteriors + epilogues queue up on the device (hbo_acq_samples) -- the counterpart of a `jax.vmap`
over the draws; the `*_sub` functions are the host restatement of the maths, used for callables outside
the registry.
"""
import functools
from typing import Any, Callable, Union

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.basics.params_utils import retrieve_params
from hyperbo_amd.gp_utils import gp

partial = functools.partial


def random_search(model, x_queries, **unused_kwargs):
  """Uniformly sampled random array (acfun.py:29-34) from model.rng (numpy Generator)."""
  assert model.rng is not None, 'Random search requires random key.'
  return model.rng.uniform(size=(x_queries.shape[0], 1))


def _norm_pdf(x):
  return np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


def _norm_cdf(x):
  from math import erfc
  return 0.5 * np.vectorize(erfc)(-np.asarray(x, dtype=np.float64) / np.sqrt(2.0))


def expected_improvement_sub(mu, std, target):
  gamma = (target - mu) / std
  return (_norm_pdf(gamma) - gamma * (1 - _norm_cdf(gamma))) * std


def probability_of_improvement_sub(mu, std, target):
  gamma = (target - mu) / std
  return -gamma


def ucb_sub(mu, std, beta=3.):
  return mu + beta * std


_NATIVE_ID = {expected_improvement_sub: nat.ACQ_EI, probability_of_improvement_sub: nat.ACQ_PI,
              ucb_sub: nat.ACQ_UCB}


def _leaves(t):
  if isinstance(t, dict):
    for k in sorted(t):
      yield from _leaves(t[k])
  else:
    yield t


def _samples_fingerprint(model, samples, dtype):
  """Content fingerprint of an HGP's parameter samples: the bytes of every leaf (an in-place edit that keeps a sum, or a replaced
  array on a recycled id, both change it), the config entries a BuiltModel reads, input_dim, warp / mean / covariance identities."""
  import hashlib
  h = hashlib.blake2b(digest_size=16)
  for smp in samples:
    for v in _leaves(smp):
      a = np.ascontiguousarray(v)
      h.update(a.dtype.str.encode()); h.update(repr(a.shape).encode()); h.update(a.tobytes())
    h.update(b'|')
  cfg = model.params.config
  return (np.dtype(dtype).str, id(model.warp_func), model.mean_func, model.cov_func, model.input_dim,
          repr(cfg.get('mlp_features')), len(samples), h.digest())


def _sample_models(model, samples, dtype):
  """One hbo_model per parameter sample, rebuilt only when the fingerprint moved (0.05-0.1 ms of Python each)."""
  finger = _samples_fingerprint(model, samples, dtype)
  cached = getattr(model, '_hbo_sample_models', None)
  if cached is None or cached[0] != finger:
    built, noises = [], []
    for smp in samples:
      ps = defs.GPParams(config=model.params.config, model=smp)
      built.append(_model.BuiltModel(model.mean_func, model.cov_func, ps, model.warp_func, dtype, model.input_dim))
      nv, = retrieve_params(ps, ['noise_variance'], warp_func=model.warp_func)
      noises.append(float(np.squeeze(nv)))
    cached = (finger, built, noises)
    model._hbo_sample_models = cached
  return cached[1], cached[2]


def _samples_per_call(n, dtype, n_samples):
  """How many samples hbo_acq_samples may factorise at once: every sample holds a complete cache (Gram -> L, L^-1 and the
  K^-1 scratch, 3 matrices of npad x ld) -- at most HALF of the device's memory, and the entry point's own limit of 4096."""
  npad = -(-n // 128) * 128
  es = np.dtype(dtype).itemsize
  per = 3 * (npad + 128) * (npad + 128 // es) * es + (1 << 20)
  mem = nat.C.c_int64(0); cus = nat.C.c_int32(0)
  ctx = nat.default_context()
  nat.lib().hbo_device_info(ctx.device, None, 0, nat.C.byref(cus), nat.C.byref(mem))
  budget = (mem.value or (64 << 30)) // 2
  return int(max(1, min(n_samples, 4096, budget // per)))


def hgp_sample_values(model, sub_dataset_key, x_queries, acq_id, acfun_param):
  """(S, M, 1) acquisition values of every model-parameter sample of an HGP from batched device calls: the S Gram matrices of
  the sub-dataset are built, factorised and inverted together (one ModelDev per task), the S posterior + acquisition passes
  follow one another on the stream (hbo_acq_samples).  gp.py:666-682 / acfun.py:72-82 as a batch.  The samples go in chunks
  that fit the device (`_samples_per_call`); a chunk that still fails to allocate is halved down to one sample per call."""
  samples = model.get_model_params_samples()
  sub = model.dataset[sub_dataset_key]
  x = np.asarray(sub.x)
  y = np.asarray(sub.y)
  dtype = _model.infer_dtype(x, y)
  x = np.ascontiguousarray(x, dtype=dtype)
  y = np.ascontiguousarray(y.reshape(x.shape[0], -1), dtype=dtype)
  xq = np.ascontiguousarray(x_queries, dtype=dtype)
  _, scale = model.predict_noise_and_scale(True, True)
  built, noises = _sample_models(model, samples, dtype)
  s_count = len(samples)
  out = np.empty((s_count, xq.shape[0], 1), dtype=dtype)
  ctx = nat.default_context()
  chunk = _samples_per_call(x.shape[0], dtype, s_count)
  s0 = 0
  while s0 < s_count:
    sc = min(chunk, s_count - s0)
    structs = (nat.Model * sc)(*[b.struct for b in built[s0:s0 + sc]])
    prm = (nat.C.c_double * sc)(*([float(acfun_param)] * sc))
    nse = (nat.C.c_double * sc)(*noises[s0:s0 + sc])
    part = out[s0:s0 + sc]
    rc = nat.lib().hbo_acq_samples(ctx.handle, structs, sc, nat.ptr(x), x.shape[0], nat.ptr(y), y.shape[1], nat.ptr(xq), xq.shape[0],
                                   int(acq_id), prm, nse, float(scale), part.ctypes.data_as(nat.C.c_void_p))
    if rc == nat.HBO_ERR_HIP and sc > 1:      # out of device memory after all: smaller chunks, same result
      chunk = max(1, sc // 2)
      continue
    ctx.check(rc)
    s0 += sc
  return out


def _array_digest(*arrays):
  import hashlib
  h = hashlib.blake2b(digest_size=16)
  for a in arrays:
    a = np.ascontiguousarray(a)
    h.update(a.dtype.str.encode()); h.update(repr(a.shape).encode()); h.update(a.tobytes())
  return h.digest()


def drop_sample_caches(model):
  """Close the per-sample factorisations an HGP acquisition gradient left on the model (called when its dataset changes)."""
  cached = getattr(model, '_hbo_sample_caches', None)
  if cached is not None:
    for h in cached[1]:
      h.close()
    model._hbo_sample_caches = None


def _hgp_sample_caches(model, sub_dataset_key, samples, dtype):
  """Factorisations of one sub-dataset under the parameter samples, kept on the model until the samples or the observations
  change (CONTENT fingerprint of x / y: an in-place edit of the observations must not be served a stale factor): bayesopt()'s inner
  L-BFGS-B (bayesopt.py:116-125) differentiates the acquisition dozens of times with the model fixed, and the reference would
  re-factorise all S samples every time (gp.py:676-678 drops the cache per sample).  Only as many samples as fit HALF of the
  device memory are kept (`_samples_per_call`, the budget of hgp_sample_values); the caller factorises the rest per call."""
  from hyperbo_amd.basics import linalg
  sd = model.dataset[sub_dataset_key]
  finger = (_samples_fingerprint(model, samples, dtype), sub_dataset_key, _array_digest(sd.x, sd.y))
  cached = getattr(model, '_hbo_sample_caches', None)
  if cached is not None and cached[0] == finger:
    return cached[1]
  drop_sample_caches(model)
  keep = _samples_per_call(np.shape(sd.x)[0], dtype, len(samples))
  handles = []
  try:
    for smp in samples[:keep]:
      ps = defs.GPParams(config=model.params.config, model=smp)
      handles.append(linalg.factor(model.mean_func, model.cov_func, ps, sd.x, sd.y, model.warp_func))
  except nat.HboError:
    # out of device memory after all: keep the factors that were built, the rest go one at a time
    if not handles:
      raise
  model._hbo_sample_caches = (finger, handles)
  return handles


def _hgp_value_and_grad(model, sub_dataset_key, x_queries, acq_id, acfun_param):
  """Mean over the parameter samples of (acquisition, d acquisition / d x): what jax differentiates when bayesopt() runs on an
  HGP (acfun.py:72-82 under bayesopt.py:116-125).  One hbo_acq_grad per sample against that sample's cached factor (samples
  beyond the cache budget: factorised, used and released on the spot)."""
  from hyperbo_amd.basics import linalg
  samples = model.get_model_params_samples()
  has_obs = model.has_observations(sub_dataset_key)
  dtype = _model.infer_dtype(model.dataset[sub_dataset_key].x, model.dataset[sub_dataset_key].y) if has_obs \
      else _model.infer_dtype(x_queries)
  xq = np.ascontiguousarray(x_queries, dtype=dtype)
  nq = xq.shape[0]
  val = np.zeros((nq, 1), dtype=np.float64)
  grad = np.zeros((nq, model.input_dim), dtype=np.float64)
  if nq == 0:
    return val.astype(dtype), grad
  built, noises = _sample_models(model, samples, dtype)
  handles = _hgp_sample_caches(model, sub_dataset_key, samples, dtype) if has_obs else []
  _, scale = model.predict_noise_and_scale(True, True)
  out = np.empty((nq, 1), dtype=dtype)
  g = np.zeros((nq, model.input_dim), dtype=np.float64)
  ctx = nat.default_context()
  for i, (bm, noise) in enumerate(zip(built, noises)):
    h, transient = (handles[i] if i < len(handles) else None), False
    if h is None and has_obs:
      sd = model.dataset[sub_dataset_key]
      ps = defs.GPParams(config=model.params.config, model=samples[i])
      h, transient = linalg.factor(model.mean_func, model.cov_func, ps, sd.x, sd.y, model.warp_func), True
    try:
      c = h.ctx if h is not None else ctx
      c.check(nat.lib().hbo_acq_grad(c.handle, bm.ref(), h.handle if h is not None else None, nat.ptr(xq), nq, int(acq_id),
                                     float(acfun_param), float(noise), float(scale), nat.ptr(out),
                                     g.ctypes.data_as(nat.C.POINTER(nat.C.c_double))))
    finally:
      if transient:
        h.close()
    val += out
    grad += g
  model.update_model_params(samples[-1])     # the side effect of the reference's loop over samples (gp.py:674-678)
  return (val / len(samples)).astype(dtype), grad / len(samples)


def acfun_wrapper(acfun_sub, acfun_callback_default):
  """acfun.py:36-93."""

  def acquisition_function(*, model, sub_dataset_key, x_queries, acfun_callback=acfun_callback_default):
    x_queries = np.asarray(x_queries)
    if isinstance(model, gp.HGP):
      acfun_param = acfun_callback(model, sub_dataset_key)
      if acfun_sub in _NATIVE_ID and model.has_observations(sub_dataset_key) and x_queries.shape[0] > 0:
        vals = hgp_sample_values(model, sub_dataset_key, x_queries, _NATIVE_ID[acfun_sub], acfun_param)
        # the reference's loop (gp.py:674-678) leaves the LAST sample in params.model and the cache dropped: same here
        model.update_model_params(model.get_model_params_samples()[-1])
        return np.mean(vals, axis=0)
      predicts = model.predict(x_queries, sub_dataset_key=sub_dataset_key, full_cov=False, with_noise=True)
      ac_vals = [acfun_sub(mu, np.sqrt(var), acfun_param) for mu, var in predicts]
      return np.mean(ac_vals, axis=0)
    acfun_param = acfun_callback(model, sub_dataset_key)
    # fused device path: posterior (gp.py:562-620 incl. +noise and T/(T-1)) + acquisition epilogue
    handle = None
    if model.has_observations(sub_dataset_key):
      model.setup_predictor(sub_dataset_key)
      handle = model.params.cache[sub_dataset_key].handle
    dtype = handle.dtype if handle is not None else _model.infer_dtype(x_queries)
    xq = np.ascontiguousarray(x_queries, dtype=dtype)
    out = np.empty((xq.shape[0], 1), dtype=dtype)
    if xq.shape[0] == 0:
      return out
    add_noise, scale = model.predict_noise_and_scale(True, True)
    bm = _model.BuiltModel(model.mean_func, model.cov_func, model.params, model.warp_func, dtype,
                           model.input_dim)
    ctx = handle.ctx if handle is not None else nat.default_context()
    ctx.check(nat.lib().hbo_acq(ctx.handle, bm.ref(), handle.handle if handle is not None else None,
                                nat.ptr(xq), xq.shape[0], _NATIVE_ID[acfun_sub], float(acfun_param),
                                float(add_noise), float(scale), nat.ptr(out)))
    return out

  def value_and_grad(*, model, sub_dataset_key, x_queries, acfun_callback=acfun_callback_default):
    """(values (M,1), d value_q / d x_queries[q] (M,D) in float64) -- the pair jax.value_and_grad of
    `lambda x: ac_func(model=..., x_queries=x[None])` yields inside bayesopt() (bayesopt.py:116-125)."""
    x_queries = np.asarray(x_queries)
    acfun_param = acfun_callback(model, sub_dataset_key)
    if isinstance(model, gp.HGP):
      return _hgp_value_and_grad(model, sub_dataset_key, x_queries, _NATIVE_ID[acfun_sub], acfun_param)
    handle = None
    if model.has_observations(sub_dataset_key):
      model.setup_predictor(sub_dataset_key)
      handle = model.params.cache[sub_dataset_key].handle
    dtype = handle.dtype if handle is not None else _model.infer_dtype(x_queries)
    xq = np.ascontiguousarray(x_queries, dtype=dtype)
    out = np.empty((xq.shape[0], 1), dtype=dtype)
    grad = np.zeros((xq.shape[0], model.input_dim), dtype=np.float64)
    if xq.shape[0] == 0:
      return out, grad
    add_noise, scale = model.predict_noise_and_scale(True, True)
    bm = _model.BuiltModel(model.mean_func, model.cov_func, model.params, model.warp_func, dtype,
                           model.input_dim)
    ctx = handle.ctx if handle is not None else nat.default_context()
    ctx.check(nat.lib().hbo_acq_grad(ctx.handle, bm.ref(), handle.handle if handle is not None else None,
                                     nat.ptr(xq), xq.shape[0], _NATIVE_ID[acfun_sub], float(acfun_param),
                                     float(add_noise), float(scale), nat.ptr(out),
                                     grad.ctypes.data_as(nat.C.POINTER(nat.C.c_double))))
    return out, grad

  acquisition_function.value_and_grad = value_and_grad
  return acquisition_function


def ei_callback_default(model, key, **unused_kwargs):
  if key not in model.dataset or model.dataset[key].y.shape[0] == 0:
    return 0.0
  return np.max(model.dataset[key].y)


expected_improvement = acfun_wrapper(acfun_sub=expected_improvement_sub,
                                     acfun_callback_default=ei_callback_default)
ei = expected_improvement


def pi_callback_default(model, key, zeta=0.1, use_std=False, **unused_kwargs):
  if key not in model.dataset or model.dataset[key].y.shape[0] == 0:
    return 0.0
  if use_std:
    return np.max(model.dataset[key].y) + zeta * np.std(model.dataset[key].y)
  return np.max(model.dataset[key].y) + zeta


probability_of_improvement = acfun_wrapper(acfun_sub=probability_of_improvement_sub,
                                           acfun_callback_default=pi_callback_default)
pi = probability_of_improvement
pi2 = acfun_wrapper(acfun_sub=probability_of_improvement_sub,
                    acfun_callback_default=partial(pi_callback_default, use_std=True))
pi3 = acfun_wrapper(acfun_sub=probability_of_improvement_sub,
                    acfun_callback_default=partial(pi_callback_default, zeta=0.05))

ucb4 = acfun_wrapper(acfun_sub=ucb_sub, acfun_callback_default=lambda a, b: 4.)
ucb3 = acfun_wrapper(acfun_sub=ucb_sub, acfun_callback_default=lambda a, b: 3.)
ucb2 = acfun_wrapper(acfun_sub=ucb_sub, acfun_callback_default=lambda a, b: 2.)
ucb = ucb3
rand = random_search
