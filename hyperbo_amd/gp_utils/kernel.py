"""Kernel library -- same call surface as hyperbo/gp_utils/kernel.py:33-58,63-183.

Each kernel is a callable `(params, vx1, vx2=None, warp_func=None, diag=False)` evaluated on the
GPU (hbo_gram).  They also carry `kernel_id` / `uses_mlp`, which is what the native objective /
predictor dispatch on instead of tracing an arbitrary Python pair-kernel.
"""
import ctypes as C

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.gp_utils import mean as _mean


def _make(kernel_id, name, uses_mlp):
  def matrix_map(params, vx1, vx2=None, warp_func=None, diag=False):
    """Returns the (n1, n2) kernel matrix; diag=True (with vx2=None) returns the (n1,) diagonal."""
    vx1 = np.asarray(vx1)
    dtype = _model.infer_dtype(vx1, vx2)
    vx1 = np.ascontiguousarray(vx1, dtype=dtype)
    x2 = None if vx2 is None else np.ascontiguousarray(np.asarray(vx2), dtype=dtype)
    use_diag = bool(diag) and vx2 is None   # kernel.py:54-57: diag only honoured when vx2 is None
    n1 = vx1.shape[0]
    n2 = n1 if x2 is None else x2.shape[0]
    out = np.empty((n1,) if use_diag else (n1, n2), dtype=dtype)
    if out.size == 0:
      return out
    params_for = params if 'noise_variance' in params.model else _with_dummy_noise(params)
    bm = _model.BuiltModel(_mean.zero, matrix_map, params_for, warp_func, dtype, vx1.shape[1])
    ctx = nat.default_context()
    ctx.check(nat.lib().hbo_gram(ctx.handle, bm.ref(), nat.ptr(vx1), n1, nat.ptr(x2), n2,
                                 int(use_diag), nat.ptr(out)), allow_not_pd=False)
    return out

  matrix_map.__name__ = name
  matrix_map.__qualname__ = name
  matrix_map.kernel_id = kernel_id
  matrix_map.uses_mlp = uses_mlp
  return matrix_map


def _with_dummy_noise(params):
  import copy
  p = copy.copy(params)
  p.model = dict(params.model)
  p.model['noise_variance'] = np.float64(0.0)
  return p


squared_exponential = _make(nat.KERNEL_SE, 'squared_exponential', False)
matern32 = _make(nat.KERNEL_MATERN32, 'matern32', False)
matern52 = _make(nat.KERNEL_MATERN52, 'matern52', False)
dot_product = _make(nat.KERNEL_DOT, 'dot_product', False)

# hyperbo/gp_utils/kernel.py:180-183; the reference's wrapper is named `kernel_mlp`
# ('mlp' in cov_func.__name__ drives GP.initialize_params, gp.py:361).
dot_product_mlp = _make(nat.KERNEL_DOT, 'kernel_mlp', True)
squared_exponential_mlp = _make(nat.KERNEL_SE, 'kernel_mlp', True)
matern32_mlp = _make(nat.KERNEL_MATERN32, 'kernel_mlp', True)
matern52_mlp = _make(nat.KERNEL_MATERN52, 'kernel_mlp', True)
