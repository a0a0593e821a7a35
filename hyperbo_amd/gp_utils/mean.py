"""Mean functions -- same call surface as hyperbo/gp_utils/mean.py:30-79 -> (n, 1) arrays."""
import numpy as np

from hyperbo_amd import _native as nat


def _make(mean_id, name):
  def vector_map(params, vx, warp_func=None):
    """Returns the (n, 1) mean vector of input array vx (evaluated by hbo_mean on the GPU)."""
    from hyperbo_amd import _model
    from hyperbo_amd.gp_utils import kernel
    vx = np.asarray(vx)
    dtype = _model.infer_dtype(vx)
    vx = np.ascontiguousarray(vx, dtype=dtype)
    out = np.empty((vx.shape[0], 1), dtype=dtype)
    if out.size == 0:
      return out
    p = _MeanOnlyParams(params)
    bm = _model.BuiltModel(vector_map, kernel.dot_product, p, warp_func, dtype, vx.shape[1])
    ctx = nat.default_context()
    ctx.check(nat.lib().hbo_mean(ctx.handle, bm.ref(), nat.ptr(vx), vx.shape[0], nat.ptr(out)),
              allow_not_pd=False)
    return out

  vector_map.__name__ = name
  vector_map.__qualname__ = name
  vector_map.mean_id = mean_id
  return vector_map


class _MeanOnlyParams:
  """View of GPParams that supplies neutral kernel parameters so a mean can be evaluated alone."""

  def __init__(self, params):
    self.config = params.config
    self.model = dict(params.model)
    self.model.setdefault('noise_variance', np.float64(0.0))
    self.model['dot_prod_sigma'] = np.float64(1.0)
    self.model['dot_prod_bias'] = np.float64(0.0)


class _NoWarpForDummy(dict):
  pass


zero = _make(nat.MEAN_ZERO, 'zero')
constant = _make(nat.MEAN_CONSTANT, 'constant')
linear = _make(nat.MEAN_LINEAR, 'linear')
linear_mlp = _make(nat.MEAN_LINEAR_MLP, 'linear_mlp')
