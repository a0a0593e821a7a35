"""Warp functions (hyperbo/gp_utils/utils.py:28-81, host side) and the array-level distances between
multivariate normals (utils.py:84-173) on top of the device Cholesky (hbo_spd_solve)."""
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


def partial_kl_mvn(mu0, cov0, mu1, cov1):
  """utils.py:84-106: tr(cov1^-1 cov0) + (mu1-mu0)^T cov1^-1 (mu1-mu0) + logdet cov1, the terms of
  KL(N0 || N1) that depend on (mu1, cov1).  cov1 is factorised and inverted on the device
  (potrf + trtri + lauum); the trace and the quadratic form are O(n^2) host reductions of that inverse."""
  from hyperbo_amd.basics import linalg
  mu_diff = np.asarray(mu1, dtype=np.float64) - np.asarray(mu0, dtype=np.float64)
  cov0 = np.atleast_2d(np.asarray(cov0))
  inv, logdet_half = linalg.spd_inverse(np.atleast_2d(np.asarray(cov1)))
  inv = np.asarray(inv, dtype=np.float64)
  return float(np.sum(inv * cov0) + mu_diff @ inv @ mu_diff + 2.0 * logdet_half)


def kl_multivariate_normal(mu0, cov0, mu1, cov1, weight=1.0, eps=0.0, partial=True):
  """utils.py:109-148.  partial=False whitens by the SVD square root of cov0 (linalg.svd_matrix_sqrt -- a
  small host-side LAPACK call, as in the reference) and then runs the same device path."""
  from hyperbo_amd.basics import linalg
  cov0 = np.atleast_2d(np.asarray(cov0)); cov1 = np.atleast_2d(np.asarray(cov1))
  if eps > 0.:
    cov0 = cov0 + np.eye(cov0.shape[0]) * eps
    cov1 = cov1 + np.eye(cov1.shape[0]) * eps
  if partial:
    return weight * partial_kl_mvn(mu0, cov0, mu1, cov1)
  chol0 = linalg.svd_matrix_sqrt(cov0)
  chol0inv = np.linalg.pinv(chol0)
  mu1w = chol0inv @ (np.asarray(mu1) - np.asarray(mu0))
  cov1w = chol0inv @ cov1 @ chol0inv.T
  return weight * 0.5 * (partial_kl_mvn(np.zeros_like(mu1w), np.eye(cov1w.shape[0]), mu1w, cov1w) - chol0.shape[1])


def euclidean_multivariate_normal(mu0, cov0, mu1, cov1, mean_weight=1., cov_weight=1., **unused_kwargs):
  """utils.py:151-173."""
  from hyperbo_amd.basics import linalg
  mean_diff = linalg.safe_l2norm(np.asarray(mu0) - np.asarray(mu1))
  cov_diff = linalg.safe_l2norm((np.asarray(cov0) - np.asarray(cov1)).flatten())
  return mean_weight * mean_diff + cov_weight * cov_diff
