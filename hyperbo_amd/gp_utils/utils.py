"""Warp functions -- hyperbo/gp_utils/utils.py:28-81 (host side, NumPy)."""
import numpy as np

EPS = 1e-10

identity_warp = lambda x: x


def softplus_warp(x):
  x = np.asarray(x)
  return np.logaddexp(x, np.zeros_like(x))


def squareplus_warp(x):
  x = np.asarray(x)
  return 0.5 * (x + np.sqrt(x**2 + 4))


def DEFAULT_SOFTPLUS(x):  # pylint: disable=invalid-name
  return softplus_warp(x) + EPS


DEFAULT_WARP_FUNC = {
    'constant': identity_warp,
    'lengthscale': DEFAULT_SOFTPLUS,
    'signal_variance': DEFAULT_SOFTPLUS,
    'noise_variance': DEFAULT_SOFTPLUS,
    'dot_prod_sigma': DEFAULT_SOFTPLUS,
}


def warp_derivative(fn, raw):
  """d warp / d raw for the closed set of warps; the reference gets this from jax autodiff."""
  raw = np.asarray(raw, dtype=np.float64)
  if fn is identity_warp:
    return np.ones_like(raw)
  if fn is DEFAULT_SOFTPLUS or fn is softplus_warp:
    return 1.0 / (1.0 + np.exp(-raw))
  if fn is squareplus_warp:
    return 0.5 * (1.0 + raw / np.sqrt(raw**2 + 4))
  custom = getattr(fn, 'derivative', None)
  if custom is not None:
    return np.asarray(custom(raw), dtype=np.float64)
  raise NotImplementedError(
      f'warp function {fn!r} has no analytic derivative; attach one as `fn.derivative`')
