"""CPU tests of the host-side training driver (hyperbo/basics/lbfgs.py, gp.py:53-195 counterparts):
optimisers on analytic functions, and infer_parameters driven by the ORACLE's value_and_grad
injected as the objective (no GPU needed) -- mirrors gp_test.py:58-148 / objectives_test.py:206-324
("loss went down")."""
import numpy as np
import pytest

import helpers
from hyperbo_amd.basics import data_utils, definitions as defs, lbfgs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
from oracle import hyperbo_oracle as o


def test_tree_flatten_roundtrip():
  tree = {'b': np.arange(6.).reshape(2, 3), 'a': {'z': np.float64(2.0), 'y': np.ones(3, dtype=np.float32)}}
  vec, unflat = lbfgs.tree_flatten(tree)
  assert vec.shape == (10,)
  back = unflat(vec * 2)
  np.testing.assert_allclose(back['b'], tree['b'] * 2)
  assert back['a']['y'].dtype == np.float32 and back['a']['z'].shape == ()


def test_lbfgs_minimises_rosenbrock_and_quadratic():
  def rosen(p):
    x = p['x']
    val = float(np.sum(100 * (x[1:] - x[:-1]**2)**2 + (1 - x[:-1])**2))
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1]**2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1]**2)
    return val, {'x': g}
  val, p, state = lbfgs.lbfgs(None, {'x': np.array([-1.2, 1.0])}, steps=200, val_and_grad_fn=rosen, tol=1e-12)
  assert val < 1e-8 and np.allclose(p['x'], 1.0, atol=1e-4) and state is not None
  a = np.diag([1.0, 10.0, 100.0])
  quad = lambda p: (0.5 * float(p['w'] @ a @ p['w']), {'w': a @ p['w']})
  val, p, _ = lbfgs.lbfgs(None, {'w': np.ones(3)}, steps=50, val_and_grad_fn=quad, tol=1e-16)
  assert val < 1e-12
  # converged at start (g.g <= tol) -> state None (lbfgs.py:241-244)
  v0, p0, s0 = lbfgs.lbfgs(None, {'w': np.zeros(3)}, val_and_grad_fn=quad)
  assert s0 is None and v0 == 0.0
  # NaN objective: linesearch returns where it started
  nanf = lambda p: (float('nan'), {'w': np.ones(3)}) if np.any(p['w'] != 1.0) else (1.0, {'w': np.ones(3)})
  v, p, _ = lbfgs.lbfgs(None, {'w': np.ones(3)}, steps=3, val_and_grad_fn=nanf)
  assert v == 1.0 and np.all(p['w'] == 1.0)
  with pytest.raises(TypeError):
    lbfgs.lbfgs(lambda p: 0.0, {'w': np.ones(3)})


def test_sub_sample_dataset_iterator():
  rng = np.random.default_rng(0)
  ds = {'a': defs.SubDataset(np.arange(20.).reshape(10, 2), np.arange(10.)[:, None]),
        'b': defs.SubDataset(np.ones((3, 2)), np.ones((3, 1)), aligned='tag')}
  it = data_utils.sub_sample_dataset_iterator(rng, ds, 4)
  b1, b2 = next(it), next(it)
  assert b1['a'].x.shape == (4, 2) and b1['b'].x.shape == (3, 2) and b1['b'].aligned == 1
  assert not np.array_equal(b1['a'].x, b2['a'].x)
  np.testing.assert_allclose(b1['a'].x[:, 0] / 2, b1['a'].y[:, 0])   # rows stay paired



def test_index_iterator_draws_the_same_rows_as_the_dataset_iterator():
  """sub_sample_index_iterator (batches gathered on the device from a resident dataset) must consume the generator exactly like
  sub_sample_dataset_iterator (data_utils.py:72-100): same seed -> the same rows in the same order, sub-datasets below the batch
  size kept whole (no draw)."""
  rng = np.random.default_rng(3)
  ds = {'a': defs.SubDataset(rng.normal(size=(30, 2)), rng.normal(size=(30, 1))),
        'small': defs.SubDataset(rng.normal(size=(5, 2)), rng.normal(size=(5, 1))),
        7: defs.SubDataset(rng.normal(size=(12, 2)), rng.normal(size=(12, 3)), aligned='g')}
  it_d = data_utils.sub_sample_dataset_iterator(np.random.default_rng(11), ds, 8)
  it_i = data_utils.sub_sample_index_iterator(np.random.default_rng(11), ds, 8)
  for _ in range(4):
    batch, index = next(it_d), next(it_i)
    assert list(index) == list(ds)
    assert index['small'] is None and batch['small'].x.shape == (5, 2)
    for k in ('a', 7):
      assert index[k].dtype == np.int32 and index[k].shape == (8,) and len(set(index[k].tolist())) == 8
      np.testing.assert_array_equal(batch[k].x, ds[k].x[index[k]])
      np.testing.assert_array_equal(batch[k].y, ds[k].y[index[k]])


def _oracle_objective(cov_name, mean_name):
  def objective(**kw):
    raise AssertionError('value path not used by the driver')
  def value_and_grad(mean_func, cov_func, params, dataset, warp_func=None):
    dso = {k: o.SubDataset(v.x, v.y, v.aligned) for k, v in dataset.items()}
    return o.nll_value_and_grad(getattr(o, mean_name), getattr(o, cov_name),
                                o.GPParams(model=params.model, config=params.config), dso, o.DEFAULT_WARP_FUNC)
  objective.value_and_grad = value_and_grad
  return objective


@pytest.mark.parametrize('method,steps', [('adam', 30), ('lbfgs', 5)])
@pytest.mark.parametrize('cov_name,mean_name', [('squared_exponential', 'constant'), ('matern52_mlp', 'linear_mlp')])
def test_infer_parameters_reduces_loss(method, steps, cov_name, mean_name):
  rng = np.random.default_rng(1)
  d = 2
  mlp = cov_name.endswith('_mlp')
  model = helpers.make_model(rng, mean_name, mlp, d)
  ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, 25, d)) for i in range(4)}
  cfg = {'method': method, 'batch_size': 100, 'max_training_step': steps, 'learning_rate': 1e-2,
         'mlp_features': helpers.MLP_FEATURES, 'objective': _oracle_objective(cov_name, mean_name)}
  params = defs.GPParams(model=model, config=cfg)
  vg = cfg['objective'].value_and_grad
  init_loss, _ = vg(None, None, params, ds)
  losses = []
  model_obj = gp.GP(ds, getattr(mean, mean_name), getattr(kernel, cov_name), params, utils.DEFAULT_WARP_FUNC)
  model_obj.params.cache['stale'] = defs.GPCache(needs_update=False)
  out = model_obj.train(key=3, callback=lambda *a, **k: losses.append(k.get('loss', a[2] if len(a) > 2 else None)))
  final_loss, _ = vg(None, None, out, ds)
  assert final_loss < init_loss
  assert out.cache == {} and len(losses) >= 1
  assert set(out.model) == set(model)


def test_infer_parameters_guards():
  ds = {0: defs.SubDataset(np.zeros((3, 1)), np.zeros((3, 1)))}
  nan_obj = lambda **kw: None
  nan_obj.value_and_grad = lambda **kw: (float('nan'), {'constant': np.zeros(())})
  p = defs.GPParams(model={'constant': np.zeros(())}, config={'method': 'adam', 'batch_size': 5, 'max_training_step': 2,
                                                              'learning_rate': 0.1})
  with pytest.raises(ValueError):   # NaN at step 0 (gp.py:135-137)
    gp.infer_parameters(mean.constant, kernel.squared_exponential, p, ds, objective=nan_obj)
  p.config['max_training_step'] = 0
  assert gp.infer_parameters(mean.constant, kernel.squared_exponential, p, ds, objective=nan_obj) is p
  p.config.update(method='nope', max_training_step=1)
  with pytest.raises(ValueError):
    gp.infer_parameters(mean.constant, kernel.squared_exponential, p, ds, objective=nan_obj)
  with pytest.raises(NotImplementedError):
    gp.infer_parameters(mean.constant, kernel.squared_exponential, p, ds, objective=lambda **kw: 0.0)


def test_objective_algebra_composes_value_and_grad():
  # objectives.py:221-247 add / mul and the nll_reg* aliases, with stand-in companions (no device needed)
  from hyperbo_amd.gp_utils import objectives as obj
  def f(**kw):
    return 2.0
  f.value_and_grad = lambda **kw: (2.0, {'a': np.array([1.0, 2.0]), 'b': {'c': np.array(3.0)}})
  def g(**kw):
    return 5.0
  g.value_and_grad = lambda **kw: (5.0, {'a': np.array([10.0, 20.0]), 'b': {'c': np.array(30.0)}})
  h = obj.add(f, obj.mul(0.1, g))
  assert h() == 2.5
  v, gr = h.value_and_grad()
  assert v == 2.5
  np.testing.assert_allclose(gr['a'], [2.0, 4.0]); np.testing.assert_allclose(gr['b']['c'], 6.0)
  plain = obj.mul(3.0, lambda **kw: 1.0)
  assert plain() == 3.0 and not hasattr(plain, 'value_and_grad')
  assert obj.kl is obj.multivariate_normal_divergence and obj.ekl is obj.kl and obj.regkl is obj.kl
  assert obj.euc is obj.multivariate_normal_euc_distance and obj.regeuc is obj.euc
  for name in ('nll_regkl1', 'nll_regeuc1', 'nll_regkl01', 'nll_regeuc01', 'nll_regkl10', 'nll_regeuc10'):
    composed = getattr(obj, name)
    assert callable(composed) and composed.value_and_grad.accepts_device_batch
  assert gp._value_and_grad_of(obj.nll) is obj.nll_value_and_grad
  assert gp._value_and_grad_of(obj.ekl) is obj.ekl_value_and_grad
  assert gp._value_and_grad_of(obj.euc) is obj.euc_value_and_grad
  with pytest.raises(NotImplementedError):
    gp._value_and_grad_of(lambda **kw: 0.0)


def test_divergence_selection_rule_and_malformed_y():
  from hyperbo_amd.gp_utils import objectives as obj
  ds = {'iid': defs.SubDataset(np.zeros((3, 1)), np.zeros((3, 1))),
        'al': defs.SubDataset(np.zeros((4, 1)), np.zeros((4, 2)), aligned=1),
        'empty': defs.SubDataset(np.zeros((0, 1)), np.zeros((0, 2)), aligned=2)}
  assert [k for k, _ in obj.included_sub_datasets(ds, only_aligned=True)] == ['al']
  assert [k for k, _ in obj.included_sub_datasets(ds, exclude_aligned=True)] == ['iid']
  assert [k for k, _ in obj.included_sub_datasets(ds, exclude_aligned=False)] == ['iid', 'al']
