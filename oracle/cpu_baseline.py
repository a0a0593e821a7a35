"""CPU baseline for bench.py's `cpu_baseline` leg ("kind": "port") -- TEST/BENCH INFRASTRUCTURE ONLY.

A memory-lean NumPy/SciPy-LAPACK port of the same algorithm the GPU path runs for the SE-ARD +
constant-mean configuration (BASELINE.json configs[1]): potrf, potri (K^-1), alpha = K^-1 r and the
contraction sum_ij G_ij dK_ij/dtheta expressed through BLAS GEMMs, so that the heavy lifting is
multi-threaded LAPACK/BLAS (OpenBLAS) like a JAX-CPU run would be.  It is cheaper than what JAX
autodiff executes (~3-4 N^3 for the Cholesky VJP), so GPU/CPU ratios against it are
conservative.  Checked against oracle/hyperbo_oracle.py in tests/test_oracle_pins.py.
Restates hyperbo/gp_utils/objectives.py:144-156 + kernel.py:63-81 + linalg.py:36-69.
"""
import numpy as np
import scipy.linalg as spla
import scipy.linalg.lapack as lapack
import scipy.special as spsp


def nll_and_grad_se_ard_constant(x, y, raw, eps=1e-6):
  """raw: dict(lengthscale[D], signal_variance, noise_variance, constant) un-warped (softplus+1e-10 warp).
  Returns (nll, grad dict w.r.t. raw)."""
  n, d = x.shape
  sp = lambda v: np.logaddexp(np.asarray(v, dtype=np.float64), 0.0) + 1e-10
  ls, sv, noise = sp(raw['lengthscale']), float(sp(raw['signal_variance'])), float(sp(raw['noise_variance']))
  const = float(np.asarray(raw['constant']))
  xs = x / ls
  sq = np.sum(xs * xs, axis=1)
  k = xs @ xs.T                     # BLAS
  k *= -2.0
  k += sq[:, None]
  k += sq[None, :]
  np.maximum(k, 0.0, out=k)
  k *= -0.5
  np.exp(k, out=k)
  k *= sv                           # k = sv * exp(-u/2)
  cov = k.copy()
  cov[np.diag_indices(n)] += noise + eps
  r = y - const
  c, info = lapack.dpotrf(cov, lower=1, overwrite_a=1)
  if info != 0:
    return float('nan'), None
  alpha = spla.cho_solve((c, True), r)
  nll = float(0.5 * (r.T @ alpha)[0, 0] + np.sum(np.log(np.diag(c))) + 0.5 * n * np.log(2 * np.pi))
  kinv, info = lapack.dpotri(c, lower=1, overwrite_c=1)   # lower triangle of K^-1
  kinv = np.tril(kinv) + np.tril(kinv, -1).T
  g = kinv
  g -= alpha @ alpha.T
  g *= 0.5                          # G = 1/2 (K^-1 - a a^T)
  tr_g = float(np.trace(g))
  g *= k                            # G o K
  sum_gk = float(g.sum())
  rows = g.sum(axis=1)
  gx = g @ xs                       # BLAS
  # sum_ij GK_ij (xs_id - xs_jd)^2 = 2 (sum_i xs_id^2 rows_i - sum_i xs_id (GK xs)_id)
  acc = 2.0 * (np.einsum('i,id->d', rows, xs * xs) - np.einsum('id,id->d', xs, gx))
  g_ls = (-0.5) * acc * (-2.0 / ls)  # dk/du = -k/2 ; du/dls_d = -2 ds_d^2 / ls_d
  sig = spsp.expit
  grad = {
      'lengthscale': g_ls * sig(np.asarray(raw['lengthscale'], dtype=np.float64)),
      'signal_variance': np.asarray(sum_gk / sv * sig(float(raw['signal_variance']))),
      'noise_variance': np.asarray(tr_g * sig(float(raw['noise_variance']))),
      'constant': np.asarray(-float(alpha.sum())),
  }
  return nll, grad
