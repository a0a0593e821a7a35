"""Randomised parity sweep (`-m gpu`): sizes around the 128 / 512 / 1024 blocking boundaries, every kernel x mean x
MLP combination, ragged multi-task datasets with empty and multi-column members, all three objectives, posterior and
acquisition (+gradient) -- each case against the CPU oracle on the same seeded inputs.  Complements the structured
tests of test_gpu_parity.py; sizes keep the whole file within ~1 minute."""
import os

import numpy as np
import pytest

import helpers
from oracle import hyperbo_oracle as o

pytestmark = pytest.mark.gpu
WFO = o.DEFAULT_WARP_FUNC
# $HBO_FUZZ_SEEDS widens the two seeded sweeps for a soak run (default: 24 / 16 seeds, about a minute)
FUZZ_SEEDS = int(os.environ.get('HBO_FUZZ_SEEDS', '0'))
SIZES = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 511, 512, 513, 640, 1023, 1024, 1025, 1300]


def _native():
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.bo_utils import acfun
  from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
  return defs, acfun, gp, kernel, mean, objectives, utils


def _case(seed):
  rng = np.random.default_rng(1000 + seed)
  kname = helpers.KERNELS[rng.integers(4)]
  mlp = bool(rng.integers(2))
  mname = helpers.MEANS[rng.integers(4)]
  d = int(rng.integers(1, 7))
  return rng, kname, mlp, mname, d


@pytest.mark.parametrize('seed', range(FUZZ_SEEDS or 24))
def test_fuzz_objectives(gpu_ctx, seed):
  defs, _, _, kernel, mean, objectives, utils = _native()
  rng, kname, mlp, mname, d = _case(seed)
  model = helpers.make_model(rng, mname, mlp, d)
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  po, pn = o.GPParams(model=model, config=dict(cfg)), defs.GPParams(model=model, config=dict(cfg))
  ntask = int(rng.integers(1, 6))
  dso = {}
  for t in range(ntask):
    n = int(SIZES[rng.integers(len(SIZES))]) if t == 0 else int(rng.integers(1, 400))
    m = 1 if rng.integers(3) else int(rng.integers(2, 6))
    x, y = helpers.synthetic_task(rng, n, d, m=m)
    dso[f't{t}'] = o.SubDataset(x, y, aligned=(t if m > 1 else None))
  dso['empty'] = o.SubDataset(np.zeros((0, d)), np.zeros((0, 1)))
  dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  mo, mn = getattr(o, mname), getattr(mean, mname)
  exclude = bool(rng.integers(2))
  vo, go = o.nll_value_and_grad(mo, ko, po, dso, WFO, exclude_aligned=exclude)
  vn, gn = objectives.nll_value_and_grad(mn, kn, pn, dsn, utils.DEFAULT_WARP_FUNC, exclude_aligned=exclude)
  assert abs(vn - vo) <= 1e-9 * max(abs(vo), 1.0)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, 1e-9)
  for kind, fnc in (('ekl', objectives.ekl), ('euc', objectives.euc)):
    vo, go = o.divergence_value_and_grad(kind, mo, ko, po, dso, WFO)
    vn, gn = fnc.value_and_grad(mn, kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
    assert abs(vn - vo) <= 1e-8 * max(abs(vo), 1.0)
    fo, fn = helpers.flatten(go), helpers.flatten(gn)
    helpers.assert_grad_close(gn, go, 1e-8, label=kind)


@pytest.mark.parametrize('seed', range(FUZZ_SEEDS or 16))
def test_fuzz_posterior_and_acquisition(gpu_ctx, seed):
  defs, acfun, gp, kernel, mean, _, utils = _native()
  rng, kname, mlp, mname, d = _case(100 + seed)
  model = helpers.make_model(rng, mname, mlp, d)
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  po, pn = o.GPParams(model=model, config=dict(cfg)), defs.GPParams(model=model, config=dict(cfg))
  n = int(SIZES[rng.integers(len(SIZES))])
  x, y = helpers.synthetic_task(rng, n, d)
  mq = int(rng.integers(1, 300))
  xq = rng.uniform(size=(mq, d))
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  mo, mn = getattr(o, mname), getattr(mean, mname)
  ds = {'a': defs.SubDataset(x, y), 'b': defs.SubDataset(*helpers.synthetic_task(rng, 5, d))}
  m = gp.GP(ds, mn, kn, pn, utils.DEFAULT_WARP_FUNC)
  mu, var = m.predict(xq, 'a', full_cov=False, with_noise=True)
  mu_o, var_o = o.predict(mo, ko, po, x, y, xq, WFO)
  mu_o, var_o = o.gp_predict_postprocess(po, {k: o.SubDataset(v.x, v.y) for k, v in ds.items()}, mu_o, var_o, WFO, False, True, True)
  # conditioning of K + (sigma^2 + 1e-6) I bounds what fp64 can reproduce
  cond = np.linalg.cond(ko(po, x, warp_func=WFO) + np.eye(n) * (o.retrieve_params(po, ['noise_variance'], WFO)[0] + 1e-6))
  tol = 1e-13 * max(cond, 1e3)
  assert helpers.rel_err(mu, mu_o) <= tol and np.max(np.abs(var - var_o)) <= tol * max(np.max(np.abs(var_o)), 1.0)
  acq = ['ei', 'pi', 'ucb'][seed % 3]
  fn = {'ei': acfun.expected_improvement, 'pi': acfun.probability_of_improvement, 'ucb': acfun.ucb}[acq]
  param = {'ei': float(np.max(y)), 'pi': float(np.max(y)) + 0.1, 'ucb': 3.0}[acq]
  noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
  val, grad = fn.value_and_grad(model=m, sub_dataset_key='a', x_queries=xq)
  vo, go = o.acquisition_value_and_grad(acq, mo, ko, po, x, y, xq, param, WFO, add_noise=noise, scale=2.0)
  assert np.max(np.abs(val - vo)) <= 10 * tol * max(np.max(np.abs(vo)), 1.0)
  assert np.max(np.abs(grad - go)) <= 100 * tol * max(np.max(np.abs(go)), 1.0)


def test_edge_shapes(gpu_ctx):
  """Limits of the ABI: m + 1 = 128 aligned columns (the augmented tile-row is full) and one more (the data rows leave the tile:
  TaskDesc::nvec), both against the oracle; the maximum feature dimension (256); eight MLP layers; a single point."""
  from hyperbo_amd import _native as nat
  defs, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(77)
  d = 2
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = o.GPParams(model=model, config={}), defs.GPParams(model=model, config={})
  x = rng.uniform(size=(40, d))
  for m in (127, 128):
    y = rng.normal(size=(40, m))
    dso = {'a': o.SubDataset(x, y, aligned=1)}
    dsn = {'a': defs.SubDataset(x, y, aligned=1)}
    vo, go = o.divergence_value_and_grad('ekl', o.constant, o.matern52, po, dso, WFO)
    vn, gn = objectives.ekl.value_and_grad(mean.constant, kernel.matern52, pn, dsn, utils.DEFAULT_WARP_FUNC)
    assert abs(vn - vo) <= 1e-9 * abs(vo)
    helpers.assert_grad_close(gn, go, 1e-9)
    assert abs(objectives.ekl(mean.constant, kernel.matern52, pn, dsn, utils.DEFAULT_WARP_FUNC) - vo) <= 1e-9 * abs(vo)
  # D = 256 (HBO_MAX_FEATURE_DIM) with a per-dimension lengthscale
  dmax = 256
  big = {'lengthscale': helpers.inv_softplus(np.full(dmax, 6.0)), 'signal_variance': helpers.inv_softplus(1.0),
         'noise_variance': helpers.inv_softplus(0.05), 'constant': np.array(0.1)}
  xb = rng.uniform(size=(200, dmax)); yb = rng.normal(size=(200, 1))
  pob, pnb = o.GPParams(model=big, config={}), defs.GPParams(model=big, config={})
  vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, pob, {0: o.SubDataset(xb, yb)}, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pnb, {0: defs.SubDataset(xb, yb)}, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  helpers.assert_grad_close(gn, go, 1e-10)
  # eight MLP layers (HBO_MAX_MLP_LAYERS), kernel and mean on the features
  feats = (5, 7, 3, 6, 4, 8, 5, 6)
  deep = {'lengthscale': (rng.normal(size=feats[-1]) * 0.3 + 0.5), 'signal_variance': np.array(0.3), 'noise_variance': np.array(-2.0),
          'mlp_params': {}, 'linear_mean': {'kernel': rng.normal(size=(feats[-1], 1)), 'bias': rng.normal(size=1)}}
  fin = d
  for l, f in enumerate(feats):
    deep['mlp_params'][f'Dense_{l}'] = {'kernel': rng.normal(size=(fin, f)) * 0.6, 'bias': rng.normal(size=f) * 0.1}
    fin = f
  cfg = {'mlp_features': feats}
  pod, pnd = o.GPParams(model=deep, config=dict(cfg)), defs.GPParams(model=deep, config=dict(cfg))
  xd, yd = helpers.synthetic_task(rng, 150, d)
  vo, go = o.nll_value_and_grad(o.linear_mlp, o.matern32_mlp, pod, {0: o.SubDataset(xd, yd)}, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.linear_mlp, kernel.matern32_mlp, pnd, {0: defs.SubDataset(xd, yd)}, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  helpers.assert_grad_close(gn, go, 1e-10)
  # a single observation: posterior and acquisition
  m1 = gp.GP({0: defs.SubDataset(xd[:1], yd[:1])}, mean.constant, kernel.matern52, pn, utils.DEFAULT_WARP_FUNC)
  mu, var = m1.predict(xd[1:6], 0)
  mu_o, var_o = o.predict(o.constant, o.matern52, po, xd[:1], yd[:1], xd[1:6], WFO)
  noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
  np.testing.assert_allclose(mu, mu_o, rtol=1e-10); np.testing.assert_allclose(var, var_o + noise, rtol=1e-9)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_every_size_across_the_leaf_and_block_boundaries(gpu_ctx, dtype):
  """NLL + gradient for every n in 1..40 (16x16 leaf boundaries, the two-columns-per-step leaf with odd sizes, skipped
  identity leaves), 120..136 and 250..262 (128-block boundaries, single- vs multi-block path, 3-panel groups) and a few
  sizes around 4 and 5 blocks (progressive inverse with a partial last group), as ONE ragged batch against the oracle."""
  defs, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(77)
  d = 3
  sizes = list(range(1, 41)) + list(range(120, 137)) + list(range(250, 263)) + [500, 511, 513, 600, 641]
  model = helpers.make_model(rng, 'constant', False, d)
  f32 = dtype == 'float32'
  cast = (lambda t: {k: cast(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)) if f32 else (lambda t: t)
  dso, dsn = {}, {}
  for n in sizes:
    x, y = helpers.synthetic_task(rng, n, d)
    if f32: x, y = x.astype(np.float32), y.astype(np.float32)
    dso[n] = o.SubDataset(x.astype(np.float64), y.astype(np.float64)); dsn[n] = defs.SubDataset(x, y)
  po, pn = o.GPParams(model=model), defs.GPParams(model=cast(model))
  vo, k2o = o.neg_log_marginal_likelihood(o.constant, o.squared_exponential, po, dso, WFO, return_key2nll=True)
  vn, k2n = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC,
                                                   return_key2nll=True)
  tol = 2e-4 if f32 else 1e-9
  for n in sizes:
    assert abs(k2n[n] - k2o[n]) <= tol * max(abs(k2o[n]), 1.0), (n, k2n[n], k2o[n])
  _, go = o.nll_value_and_grad(o.constant, o.squared_exponential, po, dso, WFO)
  _, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, 2.5e-3 if f32 else 1e-9)


OPTION_SETS = [
    {'lookahead': 0}, {'lookahead': 2},
    {'overlap_trtri': 0},
    {'potrf_group': 1}, {'potrf_group': 2}, {'potrf_group': 4}, {'potrf_group': 8},
    {'persist_free': 0}, {'persist_free': 128}, {'persist_free': 64, 'small_nblk': 0}, {'persist_free': 200, 'small_nblk': 0},
    {'small_nblk': 0}, {'small_nblk': 4}, {'cu_yield': 0}, {'cu_yield': 1},
    {'lookahead': 0, 'potrf_group': 3, 'small_nblk': 0},
    # persistent co-running inverse products and their start point, streamed posterior
    {'trtri_free': 0}, {'trtri_free': 120, 'small_nblk': 0}, {'trtri_free': 200},
    {'trtri_at': 12}, {'trtri_at': 60, 'small_nblk': 0}, {'trtri_at': 48, 'small_nblk': 0},
    {'post_chunk': 128},
    # round 4: the one-sweep inverse off / everywhere, other group sizes, 128-tiles always / never, the persistent and the yielding batch forms
    {'sweep': 0, 'lookahead': 2}, {'sweep': 2}, {'sweep': 1, 'lookahead': 2}, {'sweep': 2, 'sweep_qs': 2, 'lookahead': 2}, {'sweep': 2, 'sweep_qs': 8, 'small_nblk': 0}, {'sweep': 2, 'sweep_big': 0},
    {'sweep': 2, 'sweep_big': 1 << 30, 'batch_bg': 1}, {'sweep': 2, 'batch_bg': 2, 'lookahead': 2}, {'sweep': 2, 'sweep_qs': 1, 'lookahead': 2},
    # F1 split over two streams: never / always (default: batches)
    {'batch_bg': 0, 'lookahead': 2}, {'lauum_persist': 0, 'small_nblk': 0}, {'small_nblk': 2, 'sweep': 0}, {'sweep': 2, 'small_nblk': 0, 'sweep_side': 0}, {'sweep': 2, 'small_nblk': 4, 'sweep_qs': 2}, {'split_f1': 0, 'lookahead': 2}, {'split_f1': 2}, {'split_f1': 2, 'lookahead': 2, 'potrf_group': 1}, {'split_f1': 2, 'sweep': 0, 'potrf_group': 2},
]
DEFAULTS = {'lookahead': 1, 'overlap_trtri': 1, 'potrf_group': 0, 'persist_free': -1, 'small_nblk': -1, 'cu_yield': 2, 'trtri_free': 48,
            'trtri_at': 0, 'post_chunk': 8192, 'sweep': 1, 'sweep_qs': 0, 'sweep_big': 4000, 'batch_bg': -1, 'split_f1': 1, 'sweep_side': 1, 'lauum_persist': 1}


@pytest.mark.parametrize('opts', OPTION_SETS, ids=lambda o_: ','.join(f'{k}={v}' for k, v in o_.items()))
def test_every_scheduling_option_gives_the_same_answer(gpu_ctx, opts):
  """The options of hbo_set_option and the measurement hooks of hbo_tune only move work between streams, launches and tile sizes: NLL, gradient and posterior of
  a single 2900-point matrix (23 blocks: persistent bulk update, partial groups at every level of the inverse) and of a
  ragged batch must not depend on them."""
  defs, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(5)
  d = 4
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = o.GPParams(model=model), defs.GPParams(model=model)
  single = {0: helpers.synthetic_task(rng, 2900, d)}
  batch = {k: helpers.synthetic_task(rng, n, d) for k, n in enumerate((700, 130, 1, 515, 300, 1290))}
  try:
    for k, v in opts.items():
      gpu_ctx.set_option(k, v)
    for data in (single, batch):
      dso = {k: o.SubDataset(x, y) for k, (x, y) in data.items()}
      dsn = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
      vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, po, dso, WFO)
      vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC)
      assert abs(vn - vo) <= 1e-9 * max(abs(vo), 1.0)
      fo, fn = helpers.flatten(go), helpers.flatten(gn)
      helpers.assert_grad_close(gn, go, 1e-9)
    x, y = single[0]
    xq = rng.uniform(size=(50, d))
    mo, so = o.predict(o.constant, o.squared_exponential, po, x[:1500], y[:1500], xq, WFO)
    m = gp.GP({0: defs.SubDataset(x[:1500], y[:1500])}, mean.constant, kernel.squared_exponential, pn, utils.DEFAULT_WARP_FUNC)
    mn, sn = m.predict(xq, 0, with_noise=False, unbiased=False)
    assert np.max(np.abs(np.asarray(mn) - mo)) <= 1e-8 * max(np.max(np.abs(mo)), 1.0)
    assert np.max(np.abs(np.asarray(sn) - so)) <= 1e-8 * max(np.max(np.abs(so)), 1e-3)
  finally:
    for k, v in DEFAULTS.items():
      gpu_ctx.set_option(k, v)


def test_poisoned_workspaces_change_nothing(gpu_ctx):
  """hbo_tune "poison" (on for the whole GPU tier, tests/conftest.py) fills the lower triangles of A and W, all of S, alpha and d f / d mu with
  NaN before every evaluation.  An evaluation that reads anything it did not recompute would turn NaN; one that is complete gives the
  same bits with and without it -- for a single matrix (block-recursive inverse), one that takes the one-sweep inverse, a ragged batch
  and a factor + posterior."""
  defs, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(123)
  d = 4
  model = helpers.make_model(rng, 'constant', False, d)
  pn = defs.GPParams(model=model)
  cases = [{0: helpers.synthetic_task(rng, 7000, d)}, {0: helpers.synthetic_task(rng, 2900, d)},
           {k: helpers.synthetic_task(rng, n, d) for k, n in enumerate((700, 130, 1, 515, 300, 1290))}]
  try:
    for data in cases:
      dsn = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
      dev = objectives.DeviceDataset(dsn)
      out = []
      for poison in (1, 0, 1):
        gpu_ctx.set_option('poison', poison)
        v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC)
        out.append((v, helpers.flatten(g)))
      assert np.isfinite(out[0][0]) and np.all(np.isfinite(out[0][1]))
      for v, g in out[1:]:
        assert v == out[0][0] and np.array_equal(g, out[0][1])
    x, y = cases[1][0]
    xq = rng.uniform(size=(40, d))
    post = []
    for poison in (1, 0):
      gpu_ctx.set_option('poison', poison)
      m = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.squared_exponential, pn, utils.DEFAULT_WARP_FUNC)
      mu, var = m.predict(xq, 0, with_noise=False, unbiased=False)
      post.append((np.asarray(mu), np.asarray(var)))
    assert np.all(np.isfinite(post[0][0])) and np.array_equal(post[0][0], post[1][0]) and np.array_equal(post[0][1], post[1][1])
  finally:
    gpu_ctx.set_option('poison', 1)


def test_one_device_dataset_many_models_and_objectives(gpu_ctx):
  """The same HBM-resident dataset evaluated in turn with different covariances (scalar and ARD length-scales: the
  packed result block changes size), means, objectives and with / without gradient: descriptors are re-uploaded only when
  they change and results come back through one pinned block -- every answer must still be the oracle's."""
  defs, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(11)
  d = 3
  data = {k: helpers.synthetic_task(rng, n, d, m=3) for k, n in enumerate((150, 70, 260))}
  dso = {k: o.SubDataset(x, y, aligned=k) for k, (x, y) in data.items()}
  dsn = {k: defs.SubDataset(x, y, aligned=k) for k, (x, y) in data.items()}
  batch = objectives.DeviceBatch(dsn)
  plan = [('squared_exponential', 'constant', 'nll', True), ('matern32', 'linear', 'ekl', True),
          ('squared_exponential', 'constant', 'euc', True), ('dot_product', 'zero', 'nll', False),
          ('matern52', 'constant', 'nll', True), ('squared_exponential', 'linear', 'nll', False),
          ('squared_exponential', 'constant', 'ekl', True), ('squared_exponential', 'constant', 'nll', True)]
  for rep, (kname, mname, obj, grad) in enumerate(plan):
    model = helpers.make_model(rng, mname, False, d)
    if rep % 2: model['lengthscale'] = np.asarray(model['lengthscale']).reshape(-1)[:1] * np.ones(())   # scalar length-scale
    po, pn = o.GPParams(model=model), defs.GPParams(model=model)
    ko, kn = getattr(o, kname), getattr(kernel, kname)
    mo, mn = getattr(o, mname), getattr(mean, mname)
    fo = {'nll': lambda: o.nll_value_and_grad(mo, ko, po, dso, WFO, exclude_aligned=False),
          'ekl': lambda: o.divergence_value_and_grad('ekl', mo, ko, po, dso, WFO),
          'euc': lambda: o.divergence_value_and_grad('euc', mo, ko, po, dso, WFO)}[obj]
    if obj == 'nll':
      fn = (lambda: objectives.nll_value_and_grad(mn, kn, pn, batch, utils.DEFAULT_WARP_FUNC, exclude_aligned=False)) if grad else \
           (lambda: (objectives.neg_log_marginal_likelihood(mn, kn, pn, batch, utils.DEFAULT_WARP_FUNC, exclude_aligned=False), None))
    else:
      fn = lambda: getattr(objectives, obj + '_value_and_grad')(mn, kn, pn, batch, utils.DEFAULT_WARP_FUNC)
    vo, go = fo()
    vn, gn = fn()
    assert abs(vn - vo) <= 1e-9 * max(abs(vo), 1.0), (rep, kname, mname, obj, vn, vo)
    if grad and gn is not None:
      a, b = helpers.flatten(go), helpers.flatten(gn)
      assert np.max(np.abs(a - b)) <= 1e-7 * max(np.max(np.abs(a)), 1e-6), (rep, kname, mname, obj)


# ---- the single-workgroup evaluation of batches whose tasks all have n <= 128 (small.hip) --------------------------------------
@pytest.mark.parametrize('dtype,tol_blocked,tol_oracle', [(np.float64, 1e-12, 1e-9), (np.float32, 2e-5, 5e-4)])
@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mname', ['zero', 'constant', 'linear'])
def test_single_workgroup_evaluation_vs_blocked_path_and_oracle(gpu_ctx, dtype, tol_blocked, tol_oracle, kname, mname):
  """One workgroup per task does Gram -> potf2 -> L^-1 -> K^-1 -> contraction in LDS (one launch per batch) when every task of the
  batch has n <= 128 -- the reference's training regime (data_utils.py:72-100 sub-samples to batch_size; gp_test.py:58-148).
  Against the blocked pipeline on the same batch (hbo_tune small_fused = 0; fp64: 1e-12) and against the oracle; ragged batch
  across the leaf boundaries (1, 15, 16, 17, 100, 127, 128 points), a task with several y columns (the (m,m) + scalar broadcast
  quirk of objectives.py:153-155), value-only calls, and a batch with ONE task of 129 points, which must take the blocked path."""
  defs, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(700 + helpers.KERNELS.index(kname) * 3 + len(mname))
  d = 5
  model = helpers.make_model(rng, mname, False, d, dtype=dtype)
  po, pn = o.GPParams(model=model), defs.GPParams(model=model)
  sizes = [1, 15, 16, 17, 100, 127, 128, 64]
  dso = {}
  for t, n in enumerate(sizes):
    x, y = helpers.synthetic_task(rng, n, d, m=(3 if t == 4 else 1), dtype=dtype)
    dso[t] = o.SubDataset(x, y)
  dsn = {k: defs.SubDataset(v.x, v.y) for k, v in dso.items()}
  ko, kn, mo, mn = getattr(o, kname), getattr(kernel, kname), getattr(o, mname), getattr(mean, mname)
  wf = utils.DEFAULT_WARP_FUNC
  dev = objectives.DeviceDataset(dsn)
  try:
    gpu_ctx.set_option('small_fused', 1)
    gpu_ctx.profile_enable(1)
    v1, g1 = objectives.nll_value_and_grad(mn, kn, pn, dev, wf)
    stages = gpu_ctx.profile_get()
    assert 'small_eval' in stages and 'potrf' not in stages, stages      # the fused launch ran, the blocked pipeline did not
    t1, k1 = objectives.neg_log_marginal_likelihood(mn, kn, pn, dev, wf, return_key2nll=True)
    gpu_ctx.set_option('small_fused', 0)
    v0, g0 = objectives.nll_value_and_grad(mn, kn, pn, dev, wf)
    assert 'potrf' in gpu_ctx.profile_get()
    t0, k0 = objectives.neg_log_marginal_likelihood(mn, kn, pn, dev, wf, return_key2nll=True)
  finally:
    gpu_ctx.set_option('small_fused', 1)
    gpu_ctx.profile_enable(0)
    dev.close()
  f1, f0 = helpers.flatten(g1), helpers.flatten(g0)
  assert abs(v1 - v0) <= tol_blocked * abs(v0), (v1, v0)
  assert np.max(np.abs(f1 - f0)) <= tol_blocked * 10 * np.max(np.abs(f0)), np.max(np.abs(f1 - f0)) / np.max(np.abs(f0))
  assert abs(t1 - t0) <= tol_blocked * abs(t0) and abs(t1 - v1) <= tol_blocked * abs(v1)
  for k in k0:
    assert abs(k1[k] - k0[k]) <= tol_blocked * max(abs(k0[k]), 1.0)
  po64 = o.GPParams(model={k_: (np.asarray(v_, np.float64) if not isinstance(v_, dict) else {a: np.asarray(b, np.float64) for a, b in v_.items()})
                           for k_, v_ in model.items()})
  dso64 = {k: o.SubDataset(np.asarray(v.x, np.float64), np.asarray(v.y, np.float64)) for k, v in dso.items()}
  vo, go = o.nll_value_and_grad(mo, ko, po64, dso64, WFO)
  fo = helpers.flatten(go)
  assert abs(v1 - vo) <= tol_oracle * abs(vo)
  assert np.max(np.abs(f1 - fo)) <= tol_oracle * 10 * np.max(np.abs(fo))
  # one task beyond 128 points: the whole batch takes the blocked pipeline
  x, y = helpers.synthetic_task(rng, 129, d, dtype=dtype)
  dsn[99] = defs.SubDataset(x, y); dso64[99] = o.SubDataset(np.asarray(x, np.float64), np.asarray(y, np.float64))
  gpu_ctx.profile_enable(1)
  try:
    v2, g2 = objectives.nll_value_and_grad(mn, kn, pn, dsn, wf)
    assert 'potrf' in gpu_ctx.profile_get() and 'small_eval' not in gpu_ctx.profile_get()
  finally:
    gpu_ctx.profile_enable(0)
  vo2, go2 = o.nll_value_and_grad(mo, ko, po64, dso64, WFO)
  assert abs(v2 - vo2) <= tol_oracle * abs(vo2)
  helpers.assert_grad_close(g2, go2, tol_oracle * 10)


def test_single_workgroup_evaluation_reports_a_matrix_that_is_not_positive_definite(gpu_ctx):
  """A task whose Gram matrix is not positive definite (duplicated rows under a noise-free dot product with eps = 0 would be;
  here: NaN in y poisons only the value, a NaN input poisons the factorisation): NaN for that task's NLL and gradient, the other
  tasks of the batch unaffected -- the same outcome as the blocked path (HBO_NOT_PD, NaN-filled outputs, never an exception)."""
  defs, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(11)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  pn = defs.GPParams(model=model)
  x0, y0 = helpers.synthetic_task(rng, 60, d)
  x1, y1 = helpers.synthetic_task(rng, 90, d)
  x1 = x1.copy(); x1[40, 1] = np.nan
  ds = {0: defs.SubDataset(x0, y0), 1: defs.SubDataset(x1, y1)}
  total, key2nll = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, ds, utils.DEFAULT_WARP_FUNC, return_key2nll=True)
  assert np.isnan(total) and np.isnan(key2nll[1]) and np.isfinite(key2nll[0])
  v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, ds, utils.DEFAULT_WARP_FUNC)
  assert np.isnan(v) and all(np.all(np.isnan(g[k])) for k in ('constant', 'lengthscale', 'noise_variance', 'signal_variance'))
  po = o.GPParams(model=model)
  vo, _ = o.nll_sub_dataset_value_and_grad(o.constant, o.squared_exponential, po, x0, y0, WFO)
  assert abs(key2nll[0] - vo) <= 1e-10 * abs(vo)
