"""ctypes binding of libhbo.so (C ABI declared in include/hbo.h).

There is deliberately NO CPU fallback: if the shared library is missing or no GPU is visible
the calls raise.  (`python -c "import __graft_entry__ as g; g.build()"` builds the library.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

HBO_OK, HBO_ERR_ARG, HBO_ERR_HIP, HBO_ERR_NODEV, HBO_ERR_UNSUPPORTED, HBO_ERR_COMM = 0, -1, -2, -3, -4, -5
HBO_NOT_PD = 1
KERNEL_SE, KERNEL_MATERN32, KERNEL_MATERN52, KERNEL_DOT = 0, 1, 2, 3
MEAN_ZERO, MEAN_CONSTANT, MEAN_LINEAR, MEAN_LINEAR_MLP = 0, 1, 2, 3
F32, F64 = 0, 1
ACQ_EI, ACQ_PI, ACQ_UCB = 0, 1, 2
MAX_MLP_LAYERS = 8
MAX_FEATURE_DIM = 256
MAX_PROFILE_STAGES = 32
UNIQUE_ID_BYTES = 128

# ($HBO_LIB: another build of the same library, for the A/B tools under tools/)
_LIB_PATH = os.environ.get('HBO_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libhbo.so')


class HboError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__(f'libhbo error {code}: {msg}')
    self.code = code


class Model(C.Structure):
  _fields_ = [
      ('kernel_id', C.c_int32), ('mean_id', C.c_int32), ('dtype', C.c_int32), ('input_dim', C.c_int32),
      ('kernel_uses_mlp', C.c_int32), ('n_layers', C.c_int32), ('features', C.c_int32 * MAX_MLP_LAYERS),
      ('n_lengthscale', C.c_int32), ('reserved0', C.c_int32),
      ('eps', C.c_double), ('signal_variance', C.c_double), ('noise_variance', C.c_double),
      ('constant', C.c_double), ('dot_prod_sigma', C.c_double), ('dot_prod_bias', C.c_double),
      ('linear_bias', C.c_double),
      ('lengthscale', C.c_void_p),
      ('mlp_kernel', C.c_void_p * MAX_MLP_LAYERS), ('mlp_bias', C.c_void_p * MAX_MLP_LAYERS),
      ('linear_kernel', C.c_void_p),
  ]


class GradLayout(C.Structure):
  _fields_ = [
      ('lengthscale', C.c_int32), ('signal_variance', C.c_int32), ('noise_variance', C.c_int32),
      ('constant', C.c_int32), ('dot_prod_sigma', C.c_int32), ('dot_prod_bias', C.c_int32),
      ('linear_kernel', C.c_int32), ('linear_bias', C.c_int32),
      ('mlp_kernel', C.c_int32 * MAX_MLP_LAYERS), ('mlp_bias', C.c_int32 * MAX_MLP_LAYERS),
      ('total', C.c_int32),
  ]


class Task(C.Structure):
  _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('n', C.c_int64), ('m', C.c_int32),
              ('reserved0', C.c_int32)]


# every symbol include/hbo.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    'hbo_ctx_create': (C.c_int, [C.c_int, C.POINTER(_P)]),
    'hbo_ctx_destroy': (C.c_int, [_P]),
    'hbo_last_error': (C.c_char_p, [_P]),
    'hbo_version': (C.c_char_p, []),
    'hbo_device_count': (C.c_int, []),
    'hbo_grad_layout_of': (C.c_int, [C.POINTER(Model), C.POINTER(GradLayout)]),
    'hbo_gram': (C.c_int, [_P, C.POINTER(Model), _P, C.c_int64, _P, C.c_int64, C.c_int, _P]),
    'hbo_mean': (C.c_int, [_P, C.POINTER(Model), _P, C.c_int64, _P]),
    'hbo_dataset_create': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(Task), C.c_int, C.POINTER(_P)]),
    'hbo_dataset_free': (C.c_int, [_P, _P]),
    'hbo_dataset_subsample': (C.c_int, [_P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(_P)]),
    'hbo_nll': (C.c_int, [_P, C.POINTER(Model), _P, C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_double)]),
    'hbo_objective': (C.c_int, [_P, C.POINTER(Model), _P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    'hbo_factor': (C.c_int, [_P, C.POINTER(Model), _P, C.c_int64, _P, C.c_int32, C.POINTER(_P)]),
    'hbo_cache_export': (C.c_int, [_P, _P, _P, _P, _P]),
    'hbo_cache_free': (C.c_int, [_P, _P]),
    'hbo_cache_append': (C.c_int, [_P, C.POINTER(Model), _P, _P, C.c_int64, _P]),
    'hbo_acq_grad': (C.c_int, [_P, C.POINTER(Model), _P, _P, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double,
                               _P, C.POINTER(C.c_double)]),
    'hbo_predict': (C.c_int, [_P, C.POINTER(Model), _P, _P, C.c_int64, C.c_int, _P, _P]),
    'hbo_acq': (C.c_int, [_P, C.POINTER(Model), _P, _P, C.c_int64, C.c_int, C.c_double, C.c_double,
                          C.c_double, _P]),
    'hbo_acq_samples': (C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P, C.c_int64, C.c_int, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.c_double, _P]),
    'hbo_spd_solve': (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int32, _P, _P, _P,
                                C.POINTER(C.c_double)]),
    'hbo_chol_solve': (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int32, _P]),
    'hbo_profile_enable': (C.c_int, [_P, C.c_int]),
    'hbo_profile_get': (C.c_int, [_P, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]),
    'hbo_set_option': (C.c_int, [_P, C.c_char_p, C.c_int64]),
    'hbo_tune': (C.c_int, [_P, C.c_char_p, C.c_int64]),   # include/hbo_tune.h: measurement hooks, not the boundary
    'hbo_mfma_peak_probe': (C.c_int, [_P, C.c_double, C.POINTER(C.c_double)]),   # include/hbo_tune.h
    'hbo_comm_unique_id': (C.c_int, [_P]),
    'hbo_comm_init': (C.c_int, [_P, C.c_int, C.c_int, _P]),
    'hbo_comm_allreduce_sum': (C.c_int, [_P, C.POINTER(C.c_double), C.c_int32]),
    'hbo_comm_destroy': (C.c_int, [_P]),
    'hbo_objective_sharded': (C.c_int, [_P, C.POINTER(Model), _P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'hbo_device_info': (C.c_int, [C.c_int, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
}

_lib = None
_lib_lock = threading.Lock()


def lib():
  """Loads libhbo.so (once).  Raises if it has not been built -- never falls back."""
  global _lib
  with _lib_lock:
    if _lib is None:
      if not os.path.exists(_LIB_PATH):
        raise HboError(HBO_ERR_NODEV, f'{_LIB_PATH} not found: build it with '
                       '`make -C hyperbo_amd/csrc` (or __graft_entry__.build()); there is no CPU fallback')
      l = C.CDLL(_LIB_PATH)
      for name, (res, args) in SIGNATURES.items():
        f = getattr(l, name)
        f.restype, f.argtypes = res, args
      _lib = l
    return _lib


def np_dtype(code):
  return np.float64 if code == F64 else np.float32


def dtype_code(dt):
  dt = np.dtype(dt)
  if dt == np.float64:
    return F64
  if dt == np.float32:
    return F32
  raise TypeError(f'unsupported dtype {dt}: float32 or float64 required')


def ptr(a):
  return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
  """One hbo_ctx (one GPU).  Not thread-safe by design (hbo.h: not re-entrant per ctx)."""

  def __init__(self, device=0):
    self._h = _P()
    rc = lib().hbo_ctx_create(int(device), C.byref(self._h))
    if rc != HBO_OK:
      raise HboError(rc, (lib().hbo_last_error(None) or b'').decode())
    self.device = device

  def check(self, rc, allow_not_pd=True):
    if rc == HBO_OK or (allow_not_pd and rc == HBO_NOT_PD):
      return rc
    raise HboError(rc, (lib().hbo_last_error(self._h) or b'').decode())

  @property
  def handle(self):
    return self._h

  def close(self):
    if self._h:
      lib().hbo_ctx_destroy(self._h)
      self._h = _P()

  PUBLIC_OPTIONS = ('potrf_group', 'lookahead', 'small_nblk', 'pool_cap_mb', 'post_chunk', 'bf16x3')

  def set_option(self, name, value):
    """The options of include/hbo.h; any other name goes to the measurement hook hbo_tune (include/hbo_tune.h)."""
    f = lib().hbo_set_option if name in self.PUBLIC_OPTIONS else lib().hbo_tune
    self.check(f(self._h, name.encode(), int(value)))

  def profile_enable(self, level):
    self.check(lib().hbo_profile_enable(self._h, int(level)))

  def profile_get(self):
    names = ((C.c_char * 32) * MAX_PROFILE_STAGES)()
    ms = (C.c_double * MAX_PROFILE_STAGES)()
    cnt = (C.c_int32 * MAX_PROFILE_STAGES)()
    n = C.c_int32(0)
    self.check(lib().hbo_profile_get(self._h, C.cast(names, _P), ms, cnt, C.byref(n)))
    return {names[i].value.decode(): (ms[i], cnt[i]) for i in range(n.value)}

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


_default_ctx = None


def default_context():
  """Process-wide context on device $HBO_DEVICE / $LOCAL_RANK / 0 (one process per GPU)."""
  global _default_ctx
  if _default_ctx is None:
    dev = int(os.environ.get('HBO_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    ndev = lib().hbo_device_count()
    if ndev <= 0:
      raise HboError(HBO_ERR_NODEV, 'no HIP device visible; hyperbo_amd has no CPU fallback')
    _default_ctx = Context(dev % ndev)
  return _default_ctx
