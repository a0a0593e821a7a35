"""GP containers -- same fields as hyperbo/basics/definitions.py:23-46, NumPy arrays."""
import dataclasses
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple, Union

import numpy as np


class GPCache:
  """Cached factorisation (hyperbo/basics/definitions.py:23-28: chol, kinvy, needs_update).

  With a device `handle` (hyperbo_amd.basics.linalg.CacheHandle) the factor lives in HBM and
  `chol` / `kinvy` are exported over PCIe only when somebody reads them (N^2 elements: 0.5 GiB at
  N=8192), so the BO loop's predict / acquisition calls never pay for the copy.
  """

  def __init__(self, chol=None, kinvy=None, needs_update=False, handle=None):
    self._chol, self._kinvy = chol, kinvy
    self.needs_update = needs_update
    self.handle = handle
    self.pending = []     # observations appended since the factorisation ([] / list of (x, y)); None = replaced

  def invalidate_arrays(self):
    self._chol = self._kinvy = None

  def _export(self):
    if self.handle is not None and (self._chol is None or self._kinvy is None):
      self._chol, self._kinvy, _ = self.handle.export()

  @property
  def chol(self):
    self._export()
    return self._chol

  @chol.setter
  def chol(self, v):
    self._chol = v

  @property
  def kinvy(self):
    self._export()
    return self._kinvy

  @kinvy.setter
  def kinvy(self, v):
    self._kinvy = v

  def __repr__(self):
    return f'GPCache(needs_update={self.needs_update}, device={self.handle is not None})'


class SubDataset(NamedTuple):
  """Sub dataset with x: n x d and y: n x m; d, m>=1."""
  x: np.ndarray
  y: np.ndarray
  aligned: Optional[Union[int, str, bool, Tuple[str, ...]]] = None


@dataclasses.dataclass
class GPParams:
  """Parameters in a GP."""
  config: Dict[str, Any] = dataclasses.field(default_factory=lambda: {})
  model: Dict[str, Any] = dataclasses.field(default_factory=lambda: {})
  cache: Dict[Union[int, str], GPCache] = dataclasses.field(default_factory=lambda: {})
  samples: List[Dict[str, Any]] = dataclasses.field(default_factory=lambda: [])


AllowedDatasetTypes = Union[
    List[Union[Tuple[np.ndarray, ...], SubDataset]],
    Dict[Union[str, int], Union[Tuple[np.ndarray, ...], SubDataset]],
]
WarpFuncType = Optional[Dict[str, Callable[[Any], Any]]]
