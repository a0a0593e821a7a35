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


# ---- the oracle-side training driver (oracle/train_oracle.py) and trajectory parity of the host driver against it --------
from oracle import train_oracle as to


def _rosen(p):
  x = np.asarray(p['x']['a'], dtype=np.float64)
  z = np.asarray(p['z'], dtype=np.float64)
  v = np.concatenate([x, z])
  val = float(np.sum(100 * (v[1:] - v[:-1]**2)**2 + (1 - v[:-1])**2))
  g = np.zeros_like(v)
  g[:-1] = -400 * v[:-1] * (v[1:] - v[:-1]**2) - 2 * (1 - v[:-1])
  g[1:] += 200 * (v[1:] - v[:-1]**2)
  return val, {'x': {'a': g[:x.size]}, 'z': g[x.size:]}


def test_oracle_adam_matches_torch_optim_adam():
  """The optax-style Adam of the oracle against an independent third-party Adam (torch.optim.Adam, CPU): the two libraries'
  update rules are algebraically the same (bias correction by the incremented count, eps outside the square root)."""
  import torch
  rng = np.random.default_rng(0)
  p0 = {'x': {'a': rng.normal(size=3)}, 'z': rng.normal(size=2)}
  tp = [torch.tensor(p0['x']['a'].copy(), dtype=torch.float64, requires_grad=True),
        torch.tensor(p0['z'].copy(), dtype=torch.float64, requires_grad=True)]
  opt = torch.optim.Adam(tp, lr=3e-2)
  state = to.AdamState(p0)
  p = p0
  for _ in range(25):
    _, g = _rosen(p)
    upd, state = to.adam_update(g, state, 3e-2)
    p = to.apply_updates(p, upd)
    tp[0].grad = torch.tensor(g['x']['a']); tp[1].grad = torch.tensor(g['z'])
    opt.step()
    _, g = None, None
    # feed torch the gradient of ITS OWN point next round: keep the two trajectories tied through p
    with torch.no_grad():
      np.testing.assert_allclose(tp[0].numpy(), p['x']['a'], rtol=1e-12, atol=1e-14)
      np.testing.assert_allclose(tp[1].numpy(), p['z'], rtol=1e-12, atol=1e-14)


def test_oracle_lbfgs_direction_equals_dense_bfgs_recursion():
  """Two-loop recursion of the oracle (lbfgs.py:142-183) == -H g with H the dense inverse-BFGS matrix built from the same
  (s, y) pairs starting at gamma I -- the textbook identity, independent of both implementations."""
  rng = np.random.default_rng(2)
  n, mem = 6, 4
  a = rng.normal(size=(n, n)); a = a @ a.T + n * np.eye(n)
  s = [rng.normal(size=n) for _ in range(mem)]
  y = [a @ si for si in s]
  g = rng.normal(size=n)
  split = lambda v: {'u': v[:2], 'w': {'k': v[2:].reshape(2, 2)}}
  d = to.lbfgs_descent_dir_nocedal(split(g), [split(v) for v in s], [split(v) for v in y])
  h = (s[-1] @ y[-1]) / (y[-1] @ y[-1]) * np.eye(n)
  for si, yi in zip(s, y):
    rho = 1. / (yi @ si)
    v = np.eye(n) - rho * np.outer(si, yi)
    h = v @ h @ v.T + rho * np.outer(si, si)
  np.testing.assert_allclose(np.concatenate([d['u'], d['w']['k'].ravel()]), -h @ g, rtol=1e-10)
  from hyperbo_amd.basics import lbfgs as host
  np.testing.assert_allclose(host.lbfgs_descent_dir_nocedal(g, s, y), -h @ g, rtol=1e-10)


def test_oracle_lbfgs_and_host_lbfgs_take_the_same_path_on_rosenbrock():
  p0 = {'x': {'a': np.array([-1.2, 1.0, 0.7])}, 'z': np.array([0.3, -0.4])}
  tr_o, tr_h = [], []
  def vg_o(p):
    v, g = _rosen(p); tr_o.append((helpers.flatten(p), v)); return v, g
  def vg_h(p):
    v, g = _rosen(p); tr_h.append((helpers.flatten(p), v)); return v, g
  trace = []
  vo, po, so = to.lbfgs(vg_o, p0, steps=25, tol=1e-14, trace=trace)
  vh, ph, sh = lbfgs.lbfgs(None, p0, steps=25, tol=1e-14, val_and_grad_fn=vg_h)
  assert len(tr_o) == len(tr_h) > 30
  for (xo, fo), (xh, fh) in zip(tr_o, tr_h):
    np.testing.assert_allclose(xh, xo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fh, fo, rtol=1e-8, atol=1e-14)
  _, g_end = _rosen(po)
  assert float(np.sum(helpers.flatten(g_end)**2)) < 1e-6 and abs(vh - vo) <= 1e-9 * max(1.0, abs(vo))   # a stationary point
  assert len(so[0]) == len(sh[0]) == 10                       # memory trimmed to 10 pairs (lbfgs.py:309-311)
  steps = [r for r in trace if r[0] == 'step']
  assert any(r[2] < 1.0 for r in steps[1:])                   # the shrinking direction of the line search was exercised
  # a small initial step makes the Armijo test pass and the curvature test fail: the growing direction (x 2.1, lbfgs.py:128-131)
  tr2, ev_o, ev_h = [], [], []
  to.lbfgs(lambda p: (ev_o.append(helpers.flatten(p)), _rosen(p))[1], p0, steps=8, alpha=0.02, tol=1e-14, trace=tr2)
  lbfgs.lbfgs(None, p0, steps=8, alpha=0.02, tol=1e-14, val_and_grad_fn=lambda p: (ev_h.append(helpers.flatten(p)), _rosen(p))[1])
  ls = [r for r in tr2 if r[0] == 'ls']
  assert any(b[1] == a[1] + 1 and abs(b[2] / a[2] - 2.1) < 1e-12 for a, b in zip(ls, ls[1:]))
  assert len(ev_o) == len(ev_h)
  np.testing.assert_allclose(np.array(ev_h), np.array(ev_o), rtol=1e-9, atol=1e-12)
  # resumable state: continuing from the returned state == running longer in one go
  v1, p1, s1 = to.lbfgs(lambda p: _rosen(p), p0, steps=6, tol=1e-14)
  v2, p2, _ = to.lbfgs(lambda p: _rosen(p), p1, steps=4, tol=1e-14, state=s1)
  h1, hp1, hs1 = lbfgs.lbfgs(None, p0, steps=6, tol=1e-14, val_and_grad_fn=_rosen)
  h2, hp2, _ = lbfgs.lbfgs(None, hp1, steps=4, tol=1e-14, val_and_grad_fn=_rosen, state=hs1)
  np.testing.assert_allclose(helpers.flatten(hp2), helpers.flatten(p2), rtol=1e-9)


def test_oracle_linesearch_guards():
  quad = lambda p: (0.5 * float(p['w'] @ p['w']), {'w': p['w']})
  p = {'w': np.ones(3)}
  v, g = quad(p)
  # ascent direction: the reference returns (params, alpha) -- value slot holds the pytree (lbfgs.py:103-106)
  out, a = to.backtracking_linesearch(quad, v, p, g, {'w': np.ones(3)}, alpha=0.5)
  assert out is p and a == 0.5
  nanf = lambda q: (float('nan'), {'w': np.ones(3)})
  out, a = to.backtracking_linesearch(nanf, v, p, g, {'w': -np.ones(3)}, max_steps=4)
  assert out == v and a == 0.           # NaN / inf: stay where we started (lbfgs.py:136-139)
  v0, p0_, s0 = to.lbfgs(quad, {'w': np.zeros(3)})
  assert s0 is None and v0 == 0.0       # converged at start (lbfgs.py:241-244)


def _shared_batches(ds, batch_size, seed):
  """Batches for the oracle loop built from the SAME row draws the host driver makes (its index iterator on the same seed)."""
  for index in data_utils.sub_sample_index_iterator(np.random.default_rng(seed), ds, batch_size):
    batch = {}
    for i, (k, s) in enumerate(ds.items()):
      ix = index[k]
      x, y = (s.x, s.y) if ix is None else (s.x[ix], s.y[ix])
      batch[k] = o.SubDataset(x, y, i if isinstance(s.aligned, str) else s.aligned)
    yield batch


@pytest.mark.parametrize('method,steps', [('adam', 10), ('lbfgs', 3)])
@pytest.mark.parametrize('cov_name,mean_name', [('squared_exponential', 'constant'), ('matern32', 'zero'),
                                                ('matern52_mlp', 'linear_mlp'), ('dot_product_mlp', 'linear')])
def test_host_driver_trajectory_equals_oracle_driver(method, steps, cov_name, mean_name):
  """gp.infer_parameters (flat-vector L-BFGS / Adam of hyperbo_amd) vs oracle/train_oracle.infer_parameters (dict pytrees,
  restating lbfgs.py:51-349 and gp.py:53-195) on the SAME objective (the oracle's value_and_grad) and the same drawn rows:
  every objective evaluation at the same parameters with the same loss -- incl. every line-search probe."""
  rng = np.random.default_rng(5)
  d = 2
  mlp = cov_name.endswith('_mlp')
  model = helpers.make_model(rng, mean_name, mlp, d)
  ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i, n in enumerate((40, 25, 33))}
  ds['al'] = defs.SubDataset(*helpers.synthetic_task(rng, 12, d, m=3), aligned='tag')
  bs = 30 if method == 'adam' else 100
  cfg = {'method': method, 'batch_size': bs, 'max_training_step': steps, 'learning_rate': 2e-2,
         'mlp_features': helpers.MLP_FEATURES}
  evals_h = []
  objective = _oracle_objective(cov_name, mean_name)
  inner = objective.value_and_grad
  def logged(mean_func, cov_func, params, dataset, warp_func=None):
    v, g = inner(mean_func, cov_func, params, dataset, warp_func)
    evals_h.append((helpers.flatten(params.model), float(v)))
    return v, g
  objective.value_and_grad = logged
  cb_h, cb_o = [], []
  copy = lambda m: to.tree_copy(m)
  ph = gp.infer_parameters(getattr(mean, mean_name), getattr(kernel, cov_name), defs.GPParams(model=copy(model), config=dict(cfg)),
                           ds, utils.DEFAULT_WARP_FUNC, objective, key=9,
                           callback=lambda *a, **k: cb_h.append(float(k['loss'] if 'loss' in k else a[2])))
  trace = []
  dso = {k: o.SubDataset(v.x, v.y, v.aligned) for k, v in ds.items()}
  po = to.infer_parameters(getattr(o, mean_name), getattr(o, cov_name), o.GPParams(model=copy(model), config=dict(cfg)), dso,
                           o.DEFAULT_WARP_FUNC, key=None, dataset_iter=_shared_batches(ds, bs, 9), trace=trace,
                           callback=lambda *a, **k: cb_o.append(float(k['loss'] if 'loss' in k else a[2])))
  evals_o = [(helpers.flatten(r[1]), r[2]) for r in trace if r[0] == 'eval']
  assert len(evals_h) == len(evals_o) >= steps + 1
  for (xh, fh), (xo, fo) in zip(evals_h, evals_o):
    np.testing.assert_allclose(xh, xo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(fh, fo, rtol=1e-10)
  np.testing.assert_allclose(cb_h, cb_o, rtol=1e-10)
  np.testing.assert_allclose(helpers.flatten(ph.model), helpers.flatten(po.model), rtol=1e-10, atol=1e-12)
  assert ph.cache == {} and po.cache == {}
  if method == 'lbfgs':
    assert [r for r in trace if r[0] == 'step']     # line-search step sizes were recorded


def test_oracle_driver_guards_match_the_reference_rules():
  """gp.py:135-142: NaN at step 0 raises; a non-finite loss later stops and keeps the last finite parameters -- the same
  outcome from the host driver and the oracle driver."""
  ds_h = {0: defs.SubDataset(np.zeros((3, 1)), np.zeros((3, 1)))}
  ds_o = {0: o.SubDataset(np.zeros((3, 1)), np.zeros((3, 1)))}
  calls = {'n': 0}
  def vg(mean_func, cov_func, params, dataset, warp_func=None):
    calls['n'] += 1
    c = float(np.asarray(params.model['constant']))
    val = float('inf') if calls['n'] >= 4 else (c - 3.0)**2
    return val, {'constant': np.asarray(2 * (c - 3.0))}
  objective = lambda **kw: None
  objective.value_and_grad = vg
  cfg = {'method': 'adam', 'batch_size': 5, 'max_training_step': 8, 'learning_rate': 0.1}
  ph = gp.infer_parameters(mean.constant, kernel.squared_exponential, defs.GPParams(model={'constant': np.zeros(())}, config=dict(cfg)),
                           ds_h, objective=objective)
  calls['n'] = 0
  po = to.infer_parameters(o.constant, o.squared_exponential, o.GPParams(model={'constant': np.zeros(())}, config=dict(cfg)),
                           ds_o, value_and_grad=vg)
  # three finite evaluations -> params.model is the point of the THIRD evaluation (two Adam updates applied)
  np.testing.assert_allclose(ph.model['constant'], po.model['constant'], rtol=1e-12)
  assert abs(float(po.model['constant']) - 0.2) < 1e-3      # ~ two steps of size lr
  calls['n'] = 10
  with pytest.raises(ValueError):
    to.infer_parameters(o.constant, o.squared_exponential, o.GPParams(model={'constant': np.zeros(())}, config=dict(cfg)),
                        ds_o, value_and_grad=lambda *a, **k: (float('nan'), {'constant': np.zeros(())}))


def test_oracle_sub_sample_iterator_rules():
  rng = np.random.default_rng(0)
  ds = {'a': o.SubDataset(np.arange(20.).reshape(10, 2), np.arange(10.)[:, None]),
        'b': o.SubDataset(np.ones((3, 2)), np.ones((3, 1)), aligned='tag')}
  it = to.sub_sample_dataset_iterator(rng, ds, 4)
  b1, b2 = next(it), next(it)
  assert b1['a'].x.shape == (4, 2) and b1['b'].x.shape == (3, 2) and b1['b'].aligned == 1
  assert not np.array_equal(b1['a'].x, b2['a'].x)
  np.testing.assert_allclose(b1['a'].x[:, 0] / 2, b1['a'].y[:, 0])


def test_draw_batch_indices_vectorised_draw():
  """data_utils.draw_batch_indices: one vectorised draw for a whole batch -- distinct in-range rows per sub-dataset, None where a
  sub-dataset is smaller than the batch (kept whole), batch_size == n is a permutation, ragged sizes never index beyond a task's
  own rows, every row equally likely, and the task-by-task fallback for very large sub-datasets obeys the same rules."""
  rng = np.random.default_rng(0)
  sizes = [30, 5, 12, 8, 200]
  for _ in range(20):
    out = data_utils.draw_batch_indices(rng, sizes, 8)
    assert out[1] is None and all(o is not None for i, o in enumerate(out) if i != 1)
    for n, ix in zip(sizes, out):
      if ix is not None:
        assert ix.dtype == np.int32 and ix.shape == (8,) and len(set(ix.tolist())) == 8 and 0 <= ix.min() and ix.max() < n
  assert sorted(data_utils.draw_batch_indices(rng, sizes, 8)[3].tolist()) == list(range(8))       # n == batch_size: a permutation
  counts = np.zeros(12)
  for _ in range(6000):
    counts[data_utils.draw_batch_indices(rng, [12, 40], 3)[0]] += 1
  assert np.all(np.abs(counts / 6000 - 0.25) < 0.03)                                              # 3 of 12 rows: p = 1/4 each
  big = data_utils.draw_batch_indices(rng, [3_000_000, 2_500_000], 16)                            # beyond the key-matrix limit
  assert all(len(set(b.tolist())) == 16 for b in big) and big[1].max() < 2_500_000
  assert data_utils.draw_batch_indices(rng, [3, 4], 10) == [None, None]
