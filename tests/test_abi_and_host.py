"""CPU-only checks: the C-ABI library loads and exports every symbol include/hbo.h declares, the
ctypes structs match the header layout, host logic (params, warps, dataset/cache semantics,
LPT sharding) and the loud failure without a GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hyperbo_amd import _native as nat
from hyperbo_amd import parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.basics import params_utils
from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, priors, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
  """Every function any header under include/ declares (hbo.h: the boundary; hbo_tune.h: the measurement hook)."""
  names = set()
  for fn in sorted(os.listdir(os.path.join(ROOT, 'include'))):
    text = open(os.path.join(ROOT, 'include', fn)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names |= set(re.findall(r'\b(hbo_[a-z0-9_]+)\s*\(', text))
  return sorted(names)


def test_library_exports_every_declared_symbol():
  lib = nat.lib()
  declared = _header_functions()
  assert len(declared) >= 20
  for name in declared:
    assert hasattr(lib, name), f'{name} declared under include/ but not exported by libhbo.so'
  assert sorted(nat.SIGNATURES) == declared, 'python binding and header disagree on the symbol list'


def test_struct_layouts_match_header():
  assert C.sizeof(nat.Task) == 32
  # hbo_model: 6 + 8 + 2 int32 = 64 bytes, 7 doubles, 1 + 16 + 1 pointers
  assert C.sizeof(nat.Model) == 64 + 56 + 8 * 18
  assert C.sizeof(nat.GradLayout) == 4 * (8 + 16 + 1)
  assert nat.lib().hbo_version().startswith(b'hbo')


def test_grad_layout_is_pure_host_logic():
  m = nat.Model()
  m.kernel_id, m.mean_id, m.dtype, m.input_dim = nat.KERNEL_SE, nat.MEAN_CONSTANT, nat.F64, 16
  m.n_lengthscale = 16
  lay = nat.GradLayout()
  assert nat.lib().hbo_grad_layout_of(C.byref(m), C.byref(lay)) == nat.HBO_OK
  assert (lay.lengthscale, lay.signal_variance, lay.noise_variance, lay.constant, lay.total) == (0, 16, 17, 18, 19)
  assert lay.dot_prod_sigma == -1 and lay.linear_kernel == -1
  m.kernel_id, m.mean_id, m.n_lengthscale = nat.KERNEL_DOT, nat.MEAN_LINEAR, 0
  assert nat.lib().hbo_grad_layout_of(C.byref(m), C.byref(lay)) == nat.HBO_OK
  assert (lay.noise_variance, lay.dot_prod_sigma, lay.dot_prod_bias, lay.linear_kernel, lay.linear_bias, lay.total) == (0, 1, 2, 3, 19, 20)


def test_fails_loudly_without_gpu():
  if nat.lib().hbo_device_count() > 0:
    pytest.skip('a GPU is visible')
  h = C.c_void_p()
  assert nat.lib().hbo_ctx_create(0, C.byref(h)) == nat.HBO_ERR_NODEV
  assert b'no HIP device' in nat.lib().hbo_last_error(None)
  with pytest.raises(nat.HboError):
    nat.Context(0)
  p = defs.GPParams(model={'lengthscale': np.ones(2), 'signal_variance': np.array(1.0)})
  with pytest.raises(nat.HboError):  # no silent CPU fallback
    kernel.squared_exponential(p, np.zeros((3, 2)))


def test_retrieve_params_and_warps():
  p = defs.GPParams(model={'lengthscale': np.array([0.0, 1.0]), 'constant': np.array(2.0)})
  with pytest.raises(ValueError):
    params_utils.retrieve_params(p, ['noise_variance'])
  ls, c = params_utils.retrieve_params(p, ['lengthscale', 'constant'], utils.DEFAULT_WARP_FUNC)
  np.testing.assert_allclose(ls, np.log1p(np.exp([0.0, 1.0])) + 1e-10)
  assert c == 2.0
  for fn in (utils.DEFAULT_SOFTPLUS, utils.softplus_warp, utils.squareplus_warp, utils.identity_warp):
    x = np.array([-3.0, 0.2, 4.0])
    num = (fn(x + 1e-6) - fn(x - 1e-6)) / 2e-6
    np.testing.assert_allclose(utils.warp_derivative(fn, x), num, rtol=1e-6)
  with pytest.raises(NotImplementedError):
    utils.warp_derivative(lambda v: v**2, np.ones(2))
  for name, fn in priors.DEFAULT_PRIORS.items():
    v = np.array(0.37)
    num = (fn(v + 1e-6) - fn(v - 1e-6)) / 2e-6
    np.testing.assert_allclose(priors.gradient_of(fn)(v), num, rtol=1e-6)


def test_registries_and_names():
  from hyperbo_amd.bo_utils import const
  assert set(const.MEAN) == {'constant', 'linear', 'linear_mlp', 'zero'}
  assert {'squared_exponential', 'matern32', 'matern52', 'dot_product', 'dot_product_mlp'} <= set(const.KERNEL)
  assert 'mlp' in kernel.matern52_mlp.__name__ and 'mlp' not in kernel.matern52.__name__  # gp.py:361
  assert 'linear' in mean.linear_mlp.__name__ and 'mlp' in mean.linear_mlp.__name__        # gp.py:376


def test_gp_dataset_and_cache_semantics_without_device():  # gp_test.py:209-277
  x, y = np.zeros((4, 2)), np.zeros((4, 1))
  model = gp.GP([(x, y), (x[:2], y[:2])], mean.constant, kernel.squared_exponential,
                defs.GPParams(model={'constant': 1.0}))
  assert set(model.dataset) == {0, 1} and model.input_dim == 2
  model.params.cache[0] = defs.GPCache(chol=np.eye(4), kinvy=np.zeros((4, 1)), needs_update=False)
  model.update_sub_dataset((np.ones((1, 2)), np.ones((1, 1))), 0, is_append=True)
  assert model.dataset[0].x.shape == (5, 2) and model.params.cache[0].needs_update
  model.update_sub_dataset((np.ones((3, 2)), np.ones((3, 1))), 5, is_append=True)   # new key via append
  assert model.dataset[5].x.shape == (3, 2) and 5 not in model.params.cache
  model.update_sub_dataset((np.ones((2, 2)), np.ones((2, 1))), 1)                    # replace, key not cached
  assert model.dataset[1].x.shape == (2, 2) and 1 not in model.params.cache
  model.update_model_params({'constant': 2.0})
  assert model.params.cache == {}
  model.params.cache[1] = defs.GPCache(np.eye(2), np.zeros((2, 1)), False)
  model.set_dataset({'a': (x, y)})
  assert model.params.cache == {} and list(model.dataset) == ['a']
  add_noise, scale = gp.GP({0: (x, y), 1: (x, y), 2: defs.SubDataset(x, y, aligned=1)}, mean.constant,
                           kernel.squared_exponential,
                           defs.GPParams(model={'noise_variance': np.array(0.0)})).predict_noise_and_scale()
  assert scale == 2.0 and add_noise == 0.0  # no warp_func -> raw value; T counts non-aligned only


def test_selection_rule_and_lpt_partition():
  ds = {'a': defs.SubDataset(np.zeros((5, 1)), np.zeros((5, 1))),
        'b': defs.SubDataset(np.zeros((0, 1)), np.zeros((0, 1))),
        'c': defs.SubDataset(np.zeros((3, 1)), np.zeros((3, 2)), aligned=1)}
  assert [k for k, _ in objectives.included_sub_datasets(ds)] == ['a']
  assert [k for k, _ in objectives.included_sub_datasets(ds, exclude_aligned=False)] == ['a', 'c']
  rng = np.random.default_rng(4)
  sizes = {i: int(n) for i, n in enumerate(rng.integers(1600, 2401, size=64))}
  for g in (1, 2, 4, 8):
    shards = parallel.lpt_partition(sizes, g)
    assert sorted(k for s in shards for k in s) == sorted(sizes)
    loads = [sum(sizes[k]**3 for k in s) for s in shards]
    assert max(loads) / (sum(loads) / g) < 1.06   # near-balanced for cfg 4's task sizes
  # the measured cost model behind the partition: latency of the longest chain + throughput over n^3
  m = parallel.SHARD_COST_MODEL
  assert parallel.shard_cost_ms([]) == 0.0
  assert abs(parallel.shard_cost_ms([2432] * 8) - (m['c0'] + 19 * m['a'] + 8 * 2432.0**3 * m['b'])) < 1e-12
  assert 2.0 < parallel.shard_cost_ms([2432] * 8) / parallel.shard_cost_ms([2432] * 2) < 2.2      # 8 tasks cost ~2x two (measured 3.2 / 1.64 ms)
  few_long = parallel.lpt_partition({0: 2400, 1: 2400, 2: 300, 3: 300, 4: 300, 5: 300}, 2)
  assert sorted(map(sorted, few_long)) == [[0, 2, 4], [1, 3, 5]] or sorted(map(len, few_long)) == [3, 3]
  mine = parallel.shard_dataset(ds, 0, 2, exclude_aligned=False)
  other = parallel.shard_dataset(ds, 1, 2, exclude_aligned=False)
  assert set(mine) | set(other) == {'a', 'c'} and not (set(mine) & set(other))


def test_infer_dtype_follows_x64_promotion():
  from hyperbo_amd import _model
  f32, f64 = np.zeros(2, np.float32), np.zeros(2, np.float64)
  assert _model.infer_dtype(f64, f64) == np.float64
  assert _model.infer_dtype(f32, f32) == np.float32
  assert _model.infer_dtype(f32, f64) == np.float64          # promoted, never narrowed (was float32)
  assert _model.infer_dtype(f64, np.arange(3)) == np.float64  # integer y does not drop the computation to float32
  assert _model.infer_dtype(np.arange(3), [1.0, 2.0]) == np.float64
  assert _model.infer_dtype(f32, np.arange(3)) == np.float32  # JAX: int + float32 -> float32
  assert _model.infer_dtype(f32, None) == np.float32 and _model.infer_dtype() == np.float64


def test_infer_input_dim_from_parameters_alone():
  from hyperbo_amd import _model
  P = lambda m: defs.GPParams(model=m)
  assert _model.infer_input_dim(mean.constant, kernel.squared_exponential, P({'lengthscale': np.ones(7)})) == 7
  assert _model.infer_input_dim(mean.constant, kernel.squared_exponential, P({'lengthscale': np.array(1.0)})) is None
  lin = {'linear_mean': {'kernel': np.ones((5, 1)), 'bias': np.zeros(1)}, 'lengthscale': np.array(1.0)}
  assert _model.infer_input_dim(mean.linear, kernel.matern32, P(lin)) == 5
  mlp = {'mlp_params': {'Dense_0': {'kernel': np.ones((6, 9)), 'bias': np.zeros(9)}}, 'lengthscale': np.ones(9),
         'linear_mean': {'kernel': np.ones((9, 1)), 'bias': np.zeros(1)}}
  assert _model.infer_input_dim(mean.linear_mlp, kernel.matern52_mlp, P(mlp)) == 6
  assert _model.infer_input_dim(mean.zero, kernel.squared_exponential_mlp, P(mlp)) == 6
  assert _model.infer_input_dim(mean.zero, kernel.dot_product, P({'dot_prod_sigma': np.array(1.)})) is None


def test_option_surface_is_small():
  """include/hbo.h documents six options; everything else is a measurement hook behind hbo_tune (include/hbo_tune.h)."""
  text = open(os.path.join(ROOT, 'include', 'hbo.h')).read()
  block = text[text.index('/* Options (integers by name)'):text.index('int hbo_set_option')]
  documented = re.findall(r'^ \*   ([a-z0-9_]+) ', block, flags=re.M)
  assert sorted(documented) == sorted(nat.Context.PUBLIC_OPTIONS) and len(documented) == 6
