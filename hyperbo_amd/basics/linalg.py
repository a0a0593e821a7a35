"""Linear algebra entry points -- hyperbo/basics/linalg.py:29-110 on the GPU."""
import ctypes as C

import numpy as np

from hyperbo_amd import _model
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import params_utils

EPS = 1e-10


def solve_linear_system(coeff, b):
  """Solve A x = b for SPD A (linalg.py:29-33) -> (chol, kinvy).  Non-PD -> NaNs, no raise."""
  coeff = np.asarray(coeff)
  dtype = _model.infer_dtype(coeff, b)
  a = np.ascontiguousarray(coeff, dtype=dtype)
  bb = np.ascontiguousarray(np.asarray(b), dtype=dtype)
  b2 = bb.reshape(bb.shape[0], -1)
  n = a.shape[0]
  chol = np.empty((n, n), dtype=dtype)
  x = np.empty_like(b2)
  ctx = nat.default_context()
  ctx.check(nat.lib().hbo_spd_solve(ctx.handle, nat.dtype_code(dtype), nat.ptr(a), n, nat.ptr(b2),
                                    b2.shape[1], nat.ptr(chol), None, nat.ptr(x), None))
  return chol, x.reshape(bb.shape)


def spd_inverse(coeff):
  """Full symmetric inverse via potrf + trtri + lauum (the gradient path's K^-1)."""
  dtype = _model.infer_dtype(coeff)
  a = np.ascontiguousarray(np.asarray(coeff), dtype=dtype)
  n = a.shape[0]
  inv = np.empty((n, n), dtype=dtype)
  logdet_half = C.c_double(0.0)
  ctx = nat.default_context()
  ctx.check(nat.lib().hbo_spd_solve(ctx.handle, nat.dtype_code(dtype), nat.ptr(a), n, None, 0, None,
                                    nat.ptr(inv), None, C.byref(logdet_half)))
  return inv, logdet_half.value


def inverse_spdmatrix_vector_product(spd_matrix, x, cached_cholesky=None):
  """linalg.py:129-145: spd_matrix^-1 x.  With `cached_cholesky` (a lower factor as an array, what the reference's
  GPCache.chol holds) no factorisation is repeated: the factor goes to the device and two substitution sweeps run there
  (hbo_chol_solve; the device-resident equivalent without the upload is the hbo_cache handle)."""
  if cached_cholesky is not None:
    x = np.asarray(x)
    dtype = _model.infer_dtype(spd_matrix, x)
    chol = np.ascontiguousarray(np.asarray(cached_cholesky), dtype=dtype)
    n = chol.shape[0]
    b = np.ascontiguousarray(x, dtype=dtype).reshape(n, -1)
    out = np.empty_like(b)
    ctx = nat.default_context()
    ctx.check(nat.lib().hbo_chol_solve(ctx.handle, nat.dtype_code(dtype), nat.ptr(chol), n, nat.ptr(b), b.shape[1], nat.ptr(out)),
              allow_not_pd=False)
    return out.reshape(x.shape)
  return solve_linear_system(spd_matrix, x)[1]


def svd_matrix_sqrt(cov):
  """linalg.py:113-126: A with A A^T = cov, columns truncated to the numerical rank (host LAPACK; only used
  by the non-partial KL, outside the device hot path)."""
  cov = np.asarray(cov)
  u, s, _ = np.linalg.svd(cov)
  factor_ = u * np.sqrt(s[..., None, :])
  tol = s.max() * np.finfo(s.dtype).eps / 2. * np.sqrt(2 * cov.shape[0] + 1.)
  rank = np.count_nonzero(s > tol)
  return factor_[:, :rank]


def safe_l2norm(x):
  """linalg.py:194-197: l2 norm (its custom gradient -- 0 at x = 0 -- lives in the device kernels)."""
  return float(np.sqrt(np.sum(np.asarray(x, dtype=np.float64)**2)))


def factor(mean_func, cov_func, params, x, y, warp_func=None, eps=1e-6, ctx=None):
  """Device-resident factorisation (hbo_cache handle wrapper)."""
  x = np.asarray(x)
  dtype = _model.infer_dtype(x, y)
  x = np.ascontiguousarray(x, dtype=dtype)
  y = np.ascontiguousarray(np.asarray(y), dtype=dtype)
  if y.ndim == 1:
    y = y[:, None]
  return CacheHandle(mean_func, cov_func, params, x, y, warp_func, eps, ctx or nat.default_context())


class CacheHandle:
  """Owns an hbo_cache (x, chol, chol^-1, kinvy in HBM)."""

  def __init__(self, mean_func, cov_func, params, x, y, warp_func, eps, ctx):
    self.ctx = ctx
    self.dtype = x.dtype
    self.n, self.m = y.shape
    self._h = C.c_void_p()
    self._spec = (mean_func, cov_func, warp_func, eps, x.shape[1])
    bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, x.dtype, x.shape[1], eps=eps)
    self.status = ctx.check(nat.lib().hbo_factor(ctx.handle, bm.ref(), nat.ptr(x), x.shape[0], nat.ptr(y),
                                                 y.shape[1], C.byref(self._h)))

  def export(self):
    chol = np.empty((self.n, self.n), dtype=self.dtype)
    kinvy = np.empty((self.n, self.m), dtype=self.dtype)
    ymu = np.empty((self.n, self.m), dtype=self.dtype)
    self.ctx.check(nat.lib().hbo_cache_export(self.ctx.handle, self._h, nat.ptr(chol), nat.ptr(kinvy),
                                              nat.ptr(ymu)))
    return chol, kinvy, ymu

  def append(self, params, x_new, y_new):
    """O(N^2) in-place append of observations (hbo_cache_append); `params` must hold the SAME
    hyper-parameters the cache was built with.  Returns False when the padded capacity is exhausted
    (the caller re-factorises), True otherwise."""
    mean_func, cov_func, warp_func, eps, d = self._spec
    x_new = np.ascontiguousarray(np.asarray(x_new), dtype=self.dtype).reshape(-1, d)
    y_new = np.ascontiguousarray(np.asarray(y_new), dtype=self.dtype).reshape(x_new.shape[0], -1)
    if y_new.shape[1] != self.m:
      raise ValueError(f'y has {y_new.shape[1]} columns, the cache was built with {self.m}')
    bm = _model.BuiltModel(mean_func, cov_func, params, warp_func, self.dtype, d, eps=eps)
    rc = nat.lib().hbo_cache_append(self.ctx.handle, bm.ref(), self._h, nat.ptr(x_new), x_new.shape[0],
                                    nat.ptr(y_new))
    if rc == nat.HBO_ERR_UNSUPPORTED or rc == nat.HBO_NOT_PD:
      # capacity exhausted, or the appended rows made the matrix numerically indefinite part-way (the device stopped at
      # the failing row): either way the caller re-factorises, which reproduces the reference's NaN cache in the
      # second case (gp.py:552-560) -- self.n is NOT advanced past what the device accepted
      return False
    self.status = self.ctx.check(rc)
    self.n += x_new.shape[0]
    return True

  @property
  def handle(self):
    return self._h

  def close(self):
    if self._h:
      nat.lib().hbo_cache_free(self.ctx.handle, self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


def solve_gp_linear_system(mean_func, cov_func, params, x, y, warp_func=None, eps=1e-6):
  """linalg.py:72-110 -> (chol, kinvy, y - mean)."""
  h = factor(mean_func, cov_func, params, x, y, warp_func, eps)
  try:
    return h.export()
  finally:
    h.close()


def compute_delta_y_and_cov(mean_func, cov_func, params, x, y, warp_func=None, eps=1e-6):
  """linalg.py:36-69 -> (y - mu(x), cov(x,x) + I (sigma^2 + eps))."""
  x = np.asarray(x)
  y = np.asarray(y) - np.atleast_2d(mean_func(params, x, warp_func=warp_func))
  noise_variance, = params_utils.retrieve_params(params, ['noise_variance'], warp_func=warp_func)
  cov = cov_func(params, x, warp_func=warp_func)
  cov = cov + np.eye(len(x), dtype=cov.dtype) * cov.dtype.type(np.squeeze(noise_variance) + eps)
  return y.astype(cov.dtype), cov
