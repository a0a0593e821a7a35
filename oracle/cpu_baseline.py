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



# ---- OpenMP + LAPACK port (the baseline bench.py reports) -----------------------------------------
_LIB = None


def _lib():
  global _LIB
  if _LIB is None:
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcpu_port.so')
    if not os.path.exists(path):
      raise RuntimeError(f'{path} missing: run __graft_entry__.build() (gcc -fopenmp oracle/cpu_port.c)')
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    lib.hbo_cpu_gram_se.argtypes = [dp, C.c_int64, C.c_int64, C.c_double, C.c_double, dp]
    lib.hbo_cpu_contract_se.argtypes = [dp, C.c_int64, C.c_int64, C.c_double, dp, dp, dp]
    lib.hbo_cpu_gram_se.restype = lib.hbo_cpu_contract_se.restype = None
    _LIB = lib
  return _LIB


def nll_and_grad_se_ard_constant_omp(x, y, raw, eps=1e-6):
  """Same algorithm as above; the O(N^2 D) Gram build and gradient contraction run in C/OpenMP
  (oracle/cpu_port.c) over all host cores, potrf/potri in LAPACK (SciPy/OpenBLAS threads)."""
  import ctypes as C
  dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
  n, d = x.shape
  assert d <= 64
  sp = lambda v: np.logaddexp(np.asarray(v, dtype=np.float64), 0.0) + 1e-10
  ls, sv, noise = sp(raw['lengthscale']), float(sp(raw['signal_variance'])), float(sp(raw['noise_variance']))
  const = float(np.asarray(raw['constant']))
  xs = np.ascontiguousarray(x / ls)
  k = np.empty((n, n))
  _lib().hbo_cpu_gram_se(dp(xs), n, d, sv, noise + eps, dp(k))
  r = y - const
  # symmetric => the C-ordered buffer is also a valid Fortran-ordered matrix: no copies in LAPACK
  c, info = lapack.dpotrf(k.T, lower=0, overwrite_a=1)     # upper in Fortran order == lower in C order
  if info != 0:
    return float('nan'), None
  alpha, info = lapack.dpotrs(c, np.asfortranarray(r), lower=0)
  nll = float(0.5 * (r.T @ alpha)[0, 0] + np.sum(np.log(np.diagonal(c))) + 0.5 * n * np.log(2 * np.pi))
  kinv, info = lapack.dpotri(c, lower=0, overwrite_c=1)
  kinv_c = kinv.T                                            # C-order view: triangle row >= col valid
  out = np.zeros(2 + d)
  a1 = np.ascontiguousarray(alpha[:, 0])
  assert kinv_c.flags.c_contiguous
  _lib().hbo_cpu_contract_se(dp(xs), n, d, sv, dp(kinv_c), dp(a1), dp(out))
  sig = spsp.expit
  grad = {
      'lengthscale': (-0.5) * out[2:] * (-2.0 / ls) * sig(np.asarray(raw['lengthscale'], dtype=np.float64)),
      'signal_variance': np.asarray(out[0] / sv * sig(float(raw['signal_variance']))),
      'noise_variance': np.asarray(out[1] * sig(float(raw['noise_variance']))),
      'constant': np.asarray(-float(alpha.sum())),
  }
  return nll, grad


# ---- all-core form (round 6): potrf / trtri / lauum over tiles, one single-threaded BLAS call per tile, OpenMP across every core ----
def _blas_fns():
  """Function pointers of dgemm / dtrsm / dsyrk / dpotrf / dtrtri out of SciPy's Cython BLAS / LAPACK capsules (the OpenBLAS SciPy
  ships), packed as oracle/cpu_port.c: hbo_blas_fns."""
  import ctypes as C
  import scipy.linalg.cython_blas as cb
  import scipy.linalg.cython_lapack as cl
  C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
  C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
  C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
  C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
  def ptr(mod, name):
    cap = mod.__pyx_capi__[name]
    return C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap))
  return (C.c_void_p * 5)(ptr(cb, 'dgemm'), ptr(cb, 'dtrsm'), ptr(cb, 'dsyrk'), ptr(cl, 'dpotrf'), ptr(cl, 'dtrtri'))


def effective_cpus():
  """CPUs this process can actually run on: the scheduler affinity mask AND the cgroup CPU quota (a container can show 256 CPUs in
  os.cpu_count() and be throttled to a handful by cpu.max -- 256 spinning OpenMP threads under such a quota ran the tiled form at
  the speed of ONE core on the GPU box)."""
  import math
  import os
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  quota = None
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
      q, per = f.read().split()[:2]
      if q != 'max':
        quota = float(q) / float(per)
  except (OSError, ValueError):
    try:
      with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:   # cgroup v1
        q, per = float(f.read()), float(g.read())
        if q > 0:
          quota = q / per
    except (OSError, ValueError):
      pass
  if quota is not None:
    n = max(1, min(n, int(math.floor(quota + 1e-9)) or 1))
  return n, quota


def set_threads(n):
  """OpenMP threads of the C port (Gram build, contraction, tile algorithms)."""
  import ctypes as C
  lib = _lib()
  lib.hbo_cpu_set_threads.argtypes = [C.c_int]
  lib.hbo_cpu_set_threads.restype = None
  lib.hbo_cpu_set_threads(int(n))


def omp_threads():
  import ctypes as C
  lib = _lib()
  lib.hbo_cpu_omp_threads.restype = C.c_int
  return int(lib.hbo_cpu_omp_threads())


def nll_and_grad_se_ard_constant_tiled(x, y, raw, eps=1e-6, nb=128, timings=None):
  """The same algorithm as nll_and_grad_se_ard_constant_omp with potrf / potri replaced by the tile algorithms of
  oracle/cpu_port.c (hbo_cpu_potrf_tiled / trtri_tiled / lauum_tiled): OpenMP over ALL host cores, every tile product one
  single-threaded OpenBLAS call -- SciPy's OpenBLAS build stops at 64 threads, which left 3/4 of a 256-core host idle."""
  import ctypes as C
  import time
  from threadpoolctl import threadpool_limits
  dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
  lib = _lib()
  for nm in ('hbo_cpu_potrf_tiled', 'hbo_cpu_trtri_tiled'):
    getattr(lib, nm).argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p]
    getattr(lib, nm).restype = C.c_int
  lib.hbo_cpu_lauum_tiled.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
  lib.hbo_cpu_lauum_tiled.restype = None
  fns = _blas_fns()
  n, d = x.shape
  assert d <= 64
  sp = lambda v: np.logaddexp(np.asarray(v, dtype=np.float64), 0.0) + 1e-10
  ls, sv, noise = sp(raw['lengthscale']), float(sp(raw['signal_variance'])), float(sp(raw['noise_variance']))
  const = float(np.asarray(raw['constant']))
  xs = np.ascontiguousarray(x / ls)
  t = [time.perf_counter()]
  k = np.empty((n, n))
  lib.hbo_cpu_gram_se(dp(xs), n, d, sv, noise + eps, dp(k))
  t.append(time.perf_counter())
  r = y - const
  # the C-ordered symmetric buffer read as a column-major matrix: its UPPER factor U (K = U^T U) is the lower factor in C order
  with threadpool_limits(limits=1, user_api='blas'):
    info = lib.hbo_cpu_potrf_tiled(dp(k), n, nb, fns)
  t.append(time.perf_counter())
  if info != 0:
    return float('nan'), None
  kf = k.T                                                   # Fortran-ordered view of the same buffer
  alpha, info = lapack.dpotrs(kf, np.asfortranarray(r), lower=0)
  nll = float(0.5 * (r.T @ alpha)[0, 0] + np.sum(np.log(np.diagonal(k))) + 0.5 * n * np.log(2 * np.pi))
  kinv = np.empty((n, n))
  with threadpool_limits(limits=1, user_api='blas'):
    info = lib.hbo_cpu_trtri_tiled(dp(k), n, nb, fns)
    t.append(time.perf_counter())
    lib.hbo_cpu_lauum_tiled(dp(k), n, nb, fns, dp(kinv))
  t.append(time.perf_counter())
  # kinv: column-major upper tiles valid == the triangle {row >= col} in C order that hbo_cpu_contract_se reads -- except inside the
  # diagonal tiles, where the GEMM wrote both triangles (symmetric): valid either way
  out = np.zeros(2 + d)
  a1 = np.ascontiguousarray(alpha[:, 0])
  lib.hbo_cpu_contract_se(dp(xs), n, d, sv, dp(kinv), dp(a1), dp(out))
  t.append(time.perf_counter())
  if timings is not None:
    timings.append(dict(zip(('gram', 'potrf', 'trtri', 'lauum', 'contract'), np.diff(t))))
  sig = spsp.expit
  grad = {
      'lengthscale': (-0.5) * out[2:] * (-2.0 / ls) * sig(np.asarray(raw['lengthscale'], dtype=np.float64)),
      'signal_variance': np.asarray(out[0] / sv * sig(float(raw['signal_variance']))),
      'noise_variance': np.asarray(out[1] * sig(float(raw['noise_variance']))),
      'constant': np.asarray(-float(alpha.sum())),
  }
  return nll, grad
