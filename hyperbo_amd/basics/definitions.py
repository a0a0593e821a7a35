"""GP containers -- same fields as hyperbo/basics/definitions.py:23-46, NumPy arrays."""
import dataclasses
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple, Union

import numpy as np


@dataclasses.dataclass
class GPCache:
  """Cached factorisation.  chol/kinvy are NumPy views; `handle` keeps the device copy."""
  chol: np.ndarray
  kinvy: np.ndarray
  needs_update: bool
  handle: Any = None


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
