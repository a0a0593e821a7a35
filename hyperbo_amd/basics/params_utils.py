"""Parameter retrieval with warping -- hyperbo/basics/params_utils.py:90-111."""
from typing import Any, Callable, Dict, List, Optional

from hyperbo_amd.basics import definitions as defs

GPParams = defs.GPParams


def _verify_params(model_params: Dict[str, Any], expected_keys: List[str]):
  if not set(expected_keys).issubset(set(model_params.keys())):
    raise ValueError(f'Expected parameters are {sorted(expected_keys)}, '
                     f'but received {sorted(model_params.keys())}.')


def retrieve_params(params: GPParams, keys: List[str],
                    warp_func: Optional[Dict[str, Callable[[Any], Any]]] = None) -> List[Any]:
  """Returns a list of parameter values (warped if specified) by keys' order."""
  model_params = params.model
  _verify_params(model_params, keys)
  if warp_func:
    return [warp_func[key](model_params[key]) if key in warp_func else model_params[key]
            for key in keys]
  return [model_params[key] for key in keys]
