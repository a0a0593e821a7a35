"""Translation between the reference's (mean_func, cov_func, GPParams, warp_func) and hbo_model.

Warping stays on the host (hyperbo/basics/params_utils.py:97-111): the C ABI receives already
warped values and returns gradients w.r.t. the warped values; `unflatten_grad` applies the chain
rule of the warp (what jax.grad does for the reference at hyperbo/gp_utils/gp.py:134).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from hyperbo_amd import _native as nat
from hyperbo_amd.basics import params_utils
from hyperbo_amd.gp_utils import utils as gp_utils_utils


def _mlp_layers(mlp_params):
  layers = []
  l = 0
  while f'Dense_{l}' in mlp_params:
    layers.append(mlp_params[f'Dense_{l}'])
    l += 1
  if not layers:
    raise ValueError("mlp_params must hold flax-style 'Dense_0', 'Dense_1', ... entries")
  return layers


class BuiltModel:
  """hbo_model plus the NumPy buffers it points into (kept alive here)."""

  def __init__(self, mean_func, cov_func, params, warp_func, dtype, input_dim, eps=1e-6):
    kernel_id = getattr(cov_func, 'kernel_id', None)
    mean_id = getattr(mean_func, 'mean_id', None)
    if kernel_id is None:
      raise TypeError(f'cov_func {cov_func!r} is not one of hyperbo_amd.gp_utils.kernel (closed registry '
                      'hyperbo/bo_utils/const.py:29-35); arbitrary Python pair-kernels cannot run natively')
    if mean_id is None:
      raise TypeError(f'mean_func {mean_func!r} is not one of hyperbo_amd.gp_utils.mean')
    self.dtype = np.dtype(dtype)
    self.code = nat.dtype_code(self.dtype)
    self.uses_mlp_kernel = bool(getattr(cov_func, 'uses_mlp', False))
    self.kernel_id, self.mean_id = kernel_id, mean_id
    self.warp_func = warp_func
    self.params = params
    self._keep = []
    m = nat.Model()
    m.kernel_id, m.mean_id, m.dtype, m.input_dim = kernel_id, mean_id, self.code, int(input_dim)
    m.kernel_uses_mlp = int(self.uses_mlp_kernel)
    m.eps = float(eps)
    need_mlp = self.uses_mlp_kernel or mean_id == nat.MEAN_LINEAR_MLP
    retrieve = lambda keys: params_utils.retrieve_params(params, keys, warp_func)

    def arr(v):
      a = np.ascontiguousarray(np.asarray(v), dtype=self.dtype)
      self._keep.append(a)
      return a

    feat_dim = int(input_dim)
    if need_mlp:
      mlp_params, = retrieve(['mlp_params'])
      layers = _mlp_layers(mlp_params)
      if len(layers) > nat.MAX_MLP_LAYERS:
        raise ValueError('too many MLP layers')
      m.n_layers = len(layers)
      fin = int(input_dim)
      for l, layer in enumerate(layers):
        w = arr(layer['kernel'])
        b = arr(np.reshape(layer['bias'], (-1,)))
        if w.shape[0] != fin or b.shape[0] != w.shape[1]:
          raise ValueError(f'mlp layer {l}: kernel {w.shape} / bias {b.shape} do not chain from {fin}')
        m.features[l] = w.shape[1]
        m.mlp_kernel[l] = nat.ptr(w).value
        m.mlp_bias[l] = nat.ptr(b).value
        fin = w.shape[1]
      self.mlp_shapes = [(np.shape(l['kernel']), np.shape(l['bias'])) for l in layers]
      if self.uses_mlp_kernel:
        feat_dim = fin
      self.mlp_out = fin
    else:
      self.mlp_shapes = []
      self.mlp_out = 0
    self.feat_dim = feat_dim

    noise, = retrieve(['noise_variance'])
    m.noise_variance = float(np.squeeze(noise))
    if kernel_id == nat.KERNEL_DOT:
      sigma, bias = retrieve(['dot_prod_sigma', 'dot_prod_bias'])
      m.dot_prod_sigma = float(np.squeeze(sigma))
      m.dot_prod_bias = float(np.squeeze(bias))
      m.n_lengthscale = 0
    else:
      ls, sv = retrieve(['lengthscale', 'signal_variance'])
      ls = arr(np.reshape(ls, (-1,)))
      if ls.size not in (1, feat_dim):
        raise ValueError(f'lengthscale has {ls.size} entries, expected 1 or {feat_dim}')
      m.lengthscale = nat.ptr(ls).value
      m.n_lengthscale = ls.size
      m.signal_variance = float(np.squeeze(sv))
    if mean_id == nat.MEAN_CONSTANT:
      const, = retrieve(['constant'])
      m.constant = float(np.squeeze(const))
    if mean_id in (nat.MEAN_LINEAR, nat.MEAN_LINEAR_MLP):
      lm, = retrieve(['linear_mean'])
      w = arr(np.reshape(lm['kernel'], (-1,)))
      fin = int(input_dim) if mean_id == nat.MEAN_LINEAR else self.mlp_out
      if w.size != fin:
        raise ValueError(f'linear_mean kernel has {w.size} entries, expected {fin}')
      m.linear_kernel = nat.ptr(w).value
      m.linear_bias = float(np.squeeze(lm['bias']))
    self.struct = m
    self.layout = nat.GradLayout()
    rc = nat.lib().hbo_grad_layout_of(C.byref(m), C.byref(self.layout))
    if rc != nat.HBO_OK:
      raise nat.HboError(rc, 'hbo_grad_layout_of failed')

  def ref(self):
    return C.byref(self.struct)

  # -- gradient pytree -----------------------------------------------------------------
  def unflatten_grad(self, flat):
    """flat d/d(warped) -> pytree shaped like params.model with d/d(raw) (warp chain rule)."""
    model = self.params.model
    lay = self.layout
    wf = self.warp_func or {}

    def chain(key, g):
      raw = np.asarray(model[key], dtype=np.float64)
      g = np.reshape(np.asarray(g, dtype=np.float64), raw.shape) if np.size(g) == raw.size else \
          np.full(raw.shape, np.sum(g))
      if key in wf:
        g = g * gp_utils_utils.warp_derivative(wf[key], raw)
      return g

    def zeros_like_tree(t):
      if isinstance(t, dict):
        return {k: zeros_like_tree(v) for k, v in t.items()}
      return np.zeros(np.shape(t), dtype=np.float64)

    grads = zeros_like_tree(model)
    if lay.lengthscale >= 0 and 'lengthscale' in model:
      n_ls = self.struct.n_lengthscale
      grads['lengthscale'] = chain('lengthscale', flat[lay.lengthscale:lay.lengthscale + n_ls])
    for key, off in (('signal_variance', lay.signal_variance), ('noise_variance', lay.noise_variance),
                     ('constant', lay.constant), ('dot_prod_sigma', lay.dot_prod_sigma),
                     ('dot_prod_bias', lay.dot_prod_bias)):
      if off >= 0 and key in model:
        grads[key] = chain(key, flat[off:off + 1])
    if lay.linear_kernel >= 0:
      lm = model['linear_mean']
      fin = np.size(lm['kernel'])
      grads['linear_mean'] = {
          'kernel': np.reshape(flat[lay.linear_kernel:lay.linear_kernel + fin], np.shape(lm['kernel'])),
          'bias': np.reshape(flat[lay.linear_bias:lay.linear_bias + 1], np.shape(lm['bias'])),
      }
    if self.mlp_shapes and 'mlp_params' in model:
      g = {}
      for l, (wshape, bshape) in enumerate(self.mlp_shapes):
        wo, bo = lay.mlp_kernel[l], lay.mlp_bias[l]
        g[f'Dense_{l}'] = {
            'kernel': np.reshape(flat[wo:wo + int(np.prod(wshape))], wshape),
            'bias': np.reshape(flat[bo:bo + int(np.prod(bshape))], bshape),
        }
      grads['mlp_params'] = g
    return grads


def infer_dtype(*arrays):
  """Computation dtype for the given inputs, following JAX's promotion under JAX_ENABLE_X64: float32 only when every
  FLOATING input is float32; any float64 input -- and integer / bool / list inputs, which x64 JAX widens -- gives
  float64 (mixed float32 / float64 inputs are promoted, never silently narrowed)."""
  saw32 = False
  for a in arrays:
    if a is None:
      continue
    dt = np.asarray(a).dtype
    if dt == np.float32 or dt == np.float16:
      saw32 = True
    elif np.issubdtype(dt, np.integer) or dt == np.bool_:
      continue   # JAX: int + float32 -> float32, int alone -> float64 under x64
    else:
      return np.dtype(np.float64)
  return np.dtype(np.float32) if saw32 else np.dtype(np.float64)


def infer_input_dim(mean_func, cov_func, params):
  """Input dimension D implied by the parameters alone (first MLP layer, linear mean, ARD lengthscale), or None.
  Used where no data is at hand: a rank whose task shard is empty must still build the same model as its peers."""
  model = params.model
  uses_mlp = bool(getattr(cov_func, 'uses_mlp', False)) or getattr(mean_func, 'mean_id', None) == nat.MEAN_LINEAR_MLP
  if uses_mlp and isinstance(model.get('mlp_params'), dict) and 'Dense_0' in model['mlp_params']:
    return int(np.shape(model['mlp_params']['Dense_0']['kernel'])[0])
  if getattr(mean_func, 'mean_id', None) == nat.MEAN_LINEAR and isinstance(model.get('linear_mean'), dict):
    return int(np.size(model['linear_mean']['kernel']))
  if getattr(cov_func, 'kernel_id', None) != nat.KERNEL_DOT and 'lengthscale' in model and np.size(model['lengthscale']) > 1:
    return int(np.size(model['lengthscale']))
  return None
