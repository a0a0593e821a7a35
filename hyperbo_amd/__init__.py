"""hyperbo_amd: MI355X-native (gfx950) GP hot path behind the hyperbo.gp_utils / bo_utils API.

Python host code + hand-written HIP kernels loaded through a ctypes C ABI (include/hbo.h).
No PyTorch, JAX or Triton in the product path; NumPy arrays in and out.
"""
from hyperbo_amd import _native  # noqa: F401  (fails loudly if libhbo.so is missing)

__all__ = ['_native']
