"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI via
hyperbo_amd, against the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (stated per north_star "to a stated fp64 tolerance"):
  fp64: NLL rel 1e-10; gradient PER LEAF (helpers.assert_grad_close): every leaf L within FP64_GRAD_TOL * max(||L||_inf,
        1e-3 max|g|) -- a leaf 1e4 x smaller than the lengthscale gradient is held to its own size; chol/kinvy/mu/var 1e-9;
  fp32: 2e-4 relative on values, FP32_GRAD_TOL per leaf on gradients, 5e-3 on variances (fp32 Cholesky of a jittered Gram).
"""
import os

import numpy as np
import pytest
import scipy.linalg as spla

import helpers
from oracle import hyperbo_oracle as o

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
WFO = o.DEFAULT_WARP_FUNC
FP64_GRAD_TOL = 1e-10   # per leaf (helpers.assert_grad_close); worst measured over the suite: 4.4e-12 of its bound (round 6, gpurun_out/grad_log.txt)
FP32_GRAD_TOL = 2.5e-3   # worst measured: 2.3e-4
FP32_REGISTRY_LEAF_TOL = 2e-3   # worst leaf of the fp32 registry sweep, relative to max(leaf norm, 1e-2 max|g|)


def _native():
  from hyperbo_amd.basics import definitions as defs, linalg
  from hyperbo_amd.bo_utils import acfun
  from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
  return defs, linalg, acfun, gp, kernel, mean, objectives, utils


def _pair(model, cfg=None):
  defs = _native()[0]
  cfg = cfg or {'mlp_features': helpers.MLP_FEATURES}
  return o.GPParams(model=model, config=dict(cfg)), defs.GPParams(model=model, config=dict(cfg))


# ---- dense building blocks vs LAPACK ---------------------------------------------------------
@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize('n', [1, 5, 127, 128, 129, 300, 1000])
def test_spd_solve_vs_lapack(gpu_ctx, dtype, tol, n):
  _, linalg, *_ = _native()
  rng = np.random.default_rng(n)
  m_ = rng.normal(size=(n, n))
  a = (m_ @ m_.T / n + np.eye(n)).astype(dtype)
  b = rng.normal(size=(n, 3)).astype(dtype)
  chol, x = linalg.solve_linear_system(a, b)
  inv, ldh = linalg.spd_inverse(a)
  a64 = a.astype(np.float64)
  cref = spla.cholesky(a64, lower=True)
  assert chol.dtype == dtype and np.array_equal(np.triu(chol, 1), np.zeros_like(chol))
  assert helpers.rel_err(chol, cref) < tol * 10
  assert helpers.rel_err(x, spla.cho_solve((cref, True), b.astype(np.float64))) < tol * 100
  assert helpers.rel_err(inv, np.linalg.inv(a64)) < tol * 100
  assert abs(ldh - np.sum(np.log(np.diag(cref)))) < tol * 100 * n


def test_not_positive_definite_gives_nan_not_exception(gpu_ctx):
  defs, linalg, _, gp, kernel, mean, objectives, utils = _native()
  chol, x = linalg.solve_linear_system(-np.eye(6), np.ones((6, 1)))
  assert np.isnan(chol).all() and np.isnan(x).all()
  a = np.eye(200); a[150, 150] = -1.0   # fails in the second diagonal block
  chol, x = linalg.solve_linear_system(a, np.ones((200, 1)))
  assert np.isnan(chol).all()
  # duplicated inputs, zero noise, dot-product kernel, eps=... -> NLL NaN like the reference
  rng = np.random.default_rng(0)
  x1 = np.repeat(rng.uniform(size=(4, 2)), 40, axis=0)
  model = {'dot_prod_sigma': np.array(1.0), 'dot_prod_bias': np.array(0.0), 'noise_variance': np.array(-1e-6)}
  p = defs.GPParams(model=model)
  v = objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, p, {0: defs.SubDataset(x1, x1[:, :1])})
  assert np.isnan(v)


# ---- Gram / mean ------------------------------------------------------------------------------
@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mlp', [False, True])
@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-13), (np.float32, 2e-5)])
def test_gram_vs_oracle(gpu_ctx, kname, mlp, dtype, tol):
  defs, _, _, _, kernel, _, _, utils = _native()
  rng = np.random.default_rng(1)
  d = 3
  model = helpers.make_model(rng, 'zero', mlp, d, dtype)
  po, pn = _pair(model)
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  x1 = rng.uniform(size=(150, d)).astype(dtype); x2 = rng.uniform(size=(37, d)).astype(dtype)
  model64 = helpers.unflatten_like(model, helpers.flatten(model))
  po64 = o.GPParams(model=model64, config=po.config)
  g = kn(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC)
  assert g.shape == (150, 150) and g.dtype == dtype
  assert helpers.rel_err(g, ko(po64, x1.astype(np.float64), warp_func=WFO)) < tol
  c = kn(pn, x1, x2, warp_func=utils.DEFAULT_WARP_FUNC)
  assert c.shape == (150, 37)
  assert helpers.rel_err(c, ko(po64, x1.astype(np.float64), x2.astype(np.float64), warp_func=WFO)) < tol
  dg = kn(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC, diag=True)
  assert dg.shape == (150,)
  assert helpers.rel_err(dg, np.diag(ko(po64, x1.astype(np.float64), warp_func=WFO))) < tol
  # symmetric PSD (kernel_test.py:146-150)
  np.testing.assert_allclose(g, g.T, atol=tol)
  assert np.linalg.eigvalsh(g.astype(np.float64)).min() > -1e-5
  assert kn(pn, np.zeros((0, d), dtype), warp_func=utils.DEFAULT_WARP_FUNC).shape == (0, 0)


@pytest.mark.parametrize('kname', ['squared_exponential', 'matern32', 'matern52'])
@pytest.mark.parametrize('d', [32, 40, 64])
def test_fp32_gram_on_the_matrix_cores_vs_oracle(gpu_ctx, kname, d):
  """fp32 Gram matrices of the stationary covariances with >= 32 features take gram_mfma_kernel (csrc/gram.hip, round 6):
  u = |a|^2 + |b|^2 - 2 a.b with the dot product from the exact three-way bf16 split on the matrix cores.  Symmetric and cross
  Gram, sizes off the 128-tile grid, a feature count that is not a multiple of the 32-feature chunk, duplicated rows (u ~ 0, where
  the expansion loses what the direct form keeps): against the fp64 oracle at the fp32 Gram's 2e-5, and against the direct-form
  kernel (hbo_tune gram_mfma = 0) -- whose own error is recorded beside it."""
  defs, _, _, _, kernel, _, _, utils = _native()
  rng = np.random.default_rng(100 + d)
  model = {'lengthscale': helpers.inv_softplus(rng.uniform(1.5, 3.0, size=d)).astype(np.float32), 'signal_variance': np.float32(helpers.inv_softplus(0.8)),
           'noise_variance': np.float32(-3.0)}
  x1 = rng.uniform(-1, 1, size=(300, d)).astype(np.float32)
  x2 = rng.uniform(-1, 1, size=(150, d)).astype(np.float32)
  x2[:20] = x1[:20]                                                   # exact duplicates across the two sets
  x2[20:40] = x1[20:40] + np.float32(1e-3) * rng.normal(size=(20, d)).astype(np.float32)   # near-duplicates
  x1[280:] = x1[:20]                                                  # duplicates inside the symmetric Gram
  pn = defs.GPParams(model=model)
  po = o.GPParams(model=helpers.unflatten_like(model, helpers.flatten(model)))
  kn, ko = getattr(kernel, kname), getattr(o, kname)
  ref_s = ko(po, x1.astype(np.float64), warp_func=WFO)
  ref_c = ko(po, x1.astype(np.float64), x2.astype(np.float64), warp_func=WFO)
  errs = {}
  try:
    for name, minf in (('mfma', 32), ('direct', 0)):
      gpu_ctx.set_option('gram_mfma', minf)
      gs = kn(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC)
      gc = kn(pn, x1, x2, warp_func=utils.DEFAULT_WARP_FUNC)
      assert gs.dtype == np.float32 and gs.shape == (300, 300) and gc.shape == (300, 150)
      errs[name] = (helpers.rel_err(gs, ref_s), helpers.rel_err(gc, ref_c))
      assert errs[name][0] < 2e-5 and errs[name][1] < 2e-5, errs
      np.testing.assert_allclose(gs, gs.T, atol=2e-5)
      if name == 'mfma':
        sv = float(np.log1p(np.exp(float(model['signal_variance']))) + 1e-10)
        assert np.max(np.abs(np.diag(gs) - sv)) <= 2e-5 * sv               # a point and itself (hbo_gram's direct mode has no exact-zero rule: that is the symmetric factor path)
        assert np.max(np.abs(gc[:20, :20].diagonal() - sv)) <= 2e-5 * sv     # exact duplicates across the sets: u ~ 1e-7 |a|^2
  finally:
    gpu_ctx.set_option('gram_mfma', 32)
  if os.environ.get('HBO_GRAD_LOG'):
    with open(os.environ['HBO_GRAD_LOG'], 'a') as f_:
      f_.write('0 tol=gram32 %s d=%d (symmetric, cross) rel. error vs fp64: mfma %.3e %.3e direct %.3e %.3e\n' % ((kname, d) + errs['mfma'] + errs['direct']))


def test_device_exp_against_numpy(gpu_ctx):
  """The pair kernels' own fp64 exp (csrc/kernfun.h: hbo_exp) through the one place it is observable in isolation: the squared
  exponential with unit length-scale and amplitude on 1-D inputs against the origin is exp(-x^2 / 2), and -x^2 / 2 is formed
  identically on host and device.  <= 2 ulp over the whole range down to the underflow, exact 1 at 0, 0 beyond it, NaN stays NaN."""
  defs, _, _, _, kernel, *_ = _native()
  rng = np.random.default_rng(77)
  t = np.concatenate([[0.0, 1e-300, 1e-17, 0.5 * np.log(2.0), 1.0, 700.0, 708.0, 744.0, 745.13, 746.5, 800.0, 1e6], rng.uniform(0, 40, 4000),
                      rng.uniform(0, 745, 4000), 10.0 ** rng.uniform(-12, 2.8, 2000)])
  x1 = np.sqrt(2.0 * t)[:, None]
  pn = defs.GPParams(model={'lengthscale': np.array(1.0), 'signal_variance': np.array(1.0)})
  got = kernel.squared_exponential(pn, x1, np.zeros((1, 1)), None)[:, 0]
  arg = -0.5 * (x1[:, 0] * x1[:, 0])
  want = np.exp(arg)
  normal = want > 1e-290
  ulp = np.abs(got[normal] - want[normal]) / np.spacing(want[normal])
  assert ulp.max() <= 2.0, ulp.max()
  assert got[0] == 1.0 and np.all(got[arg < -746.0] == 0.0)
  np.testing.assert_allclose(got[~normal], want[~normal], rtol=1e-10, atol=1e-320)
  xn = np.array([[np.nan], [1.0]])
  out = kernel.squared_exponential(pn, xn, np.zeros((1, 1)), None)[:, 0]
  assert np.isnan(out[0]) and out[1] == np.exp(-0.5)


@pytest.mark.parametrize('mname', helpers.MEANS)
def test_mean_vs_oracle(gpu_ctx, mname):
  defs, _, _, _, _, mean, _, utils = _native()
  rng = np.random.default_rng(2)
  model = helpers.make_model(rng, mname, False, 3)
  po, pn = _pair(model)
  x = rng.uniform(size=(50, 3))
  mu = getattr(mean, mname)(pn, x, warp_func=utils.DEFAULT_WARP_FUNC)
  assert mu.shape == (50, 1)
  np.testing.assert_allclose(mu, getattr(o, mname)(po, x, warp_func=WFO), rtol=1e-13, atol=1e-15)


# ---- NLL and its gradient ----------------------------------------------------------------------
def _ragged_dataset(rng, d):
  dso = {0: o.SubDataset(*helpers.synthetic_task(rng, 140, d)), 1: o.SubDataset(*helpers.synthetic_task(rng, 300, d)),
         2: o.SubDataset(*helpers.synthetic_task(rng, 20, d, m=3), aligned=1), 3: o.SubDataset(np.zeros((0, d)), np.zeros((0, 1))),
         4: o.SubDataset(*helpers.synthetic_task(rng, 128, d))}
  return dso


@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mlp', [False, True])
@pytest.mark.parametrize('mname', helpers.MEANS)
@pytest.mark.parametrize('exclude_aligned', [True, False])
def test_nll_value_and_grad_vs_oracle_fp64(gpu_ctx, kname, mlp, mname, exclude_aligned):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(3)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  dso = _ragged_dataset(rng, d)
  dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  vo, go = o.nll_value_and_grad(getattr(o, mname), ko, po, dso, WFO, exclude_aligned=exclude_aligned)
  vn, gn = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, dsn,
                                         utils.DEFAULT_WARP_FUNC, exclude_aligned=exclude_aligned)
  assert set(gn) == set(go)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
  v2, k2 = objectives.neg_log_marginal_likelihood(getattr(mean, mname), kn, pn, dsn,
                                                  utils.DEFAULT_WARP_FUNC, exclude_aligned=exclude_aligned,
                                                  return_key2nll=True)
  _, k2o = o.neg_log_marginal_likelihood(getattr(o, mname), ko, po, dso, WFO,
                                         exclude_aligned=exclude_aligned, return_key2nll=True)
  assert abs(v2 - vo) <= 1e-10 * abs(vo) and set(k2) == set(k2o)
  for k in k2o:
    assert abs(k2[k] - k2o[k]) <= 1e-10 * abs(k2o[k])


@pytest.mark.parametrize('kname', ['matern32', 'matern52'])
@pytest.mark.parametrize('mlp,mname', [(False, 'constant'), (True, 'linear_mlp')])
def test_matern_with_duplicated_training_rows(gpu_ctx, kname, mlp, mname):
  """Duplicated training inputs put u == 0 OFF the diagonal of a Matern Gram matrix, where the reference's safe square root has
  gradient 0 (hyperbo/basics/linalg.py:183-188; device: kernfun.h).  NLL + every gradient leaf across a ragged batch whose
  tasks repeat rows (also across the 128-row block edge), and the acquisition value + d acq / d x at queries that sit ON
  training points, all against the oracle."""
  defs, _, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(41)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  def with_dups(n, ndup):
    x, y = helpers.synthetic_task(rng, n, d)
    src = rng.integers(0, n - ndup, size=ndup)
    x[n - ndup:] = x[src]          # later rows repeat earlier ones (different y: the noise term keeps K PD)
    return x, y
  tasks = {'a': with_dups(150, 12), 'b': with_dups(40, 40 // 2), 'c': with_dups(260, 30)}
  dso = {k: o.SubDataset(*v) for k, v in tasks.items()}
  dsn = {k: defs.SubDataset(*v) for k, v in tasks.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  vo, go = o.nll_value_and_grad(getattr(o, mname), ko, po, dso, WFO)
  vn, gn = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert np.isfinite(vo) and abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  assert np.all(np.isfinite(fn)); helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
  # acquisition at queries ON training points (two of them on a duplicated pair) and off them
  x, y = tasks['a']
  xq = np.concatenate([x[[0, 149, 75]], rng.uniform(size=(4, d))])
  m = gp.GP({'t': defs.SubDataset(x, y), 'other': defs.SubDataset(*tasks['b'])}, getattr(mean, mname), kn, pn, utils.DEFAULT_WARP_FUNC)
  noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
  for acq, fn_, param in (('ei', acfun.expected_improvement, float(np.max(y))), ('ucb', acfun.ucb, 3.0)):
    val, grad = fn_.value_and_grad(model=m, sub_dataset_key='t', x_queries=xq)
    vo_, go_ = o.acquisition_value_and_grad(acq, getattr(o, mname), ko, po, x, y, xq, param, WFO, add_noise=noise, scale=2.0)
    assert np.all(np.isfinite(grad))
    np.testing.assert_allclose(val, vo_, rtol=1e-8, atol=1e-10)
    assert np.max(np.abs(grad - go_)) <= 1e-7 * max(np.max(np.abs(go_)), 1e-3)


def test_nll_with_priors_and_scalar_lengthscale(gpu_ctx):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  from hyperbo_amd.gp_utils import priors
  rng = np.random.default_rng(5)
  model = helpers.make_model(rng, 'constant', False, 4)
  model['lengthscale'] = np.array([0.3])   # scalar lengthscale broadcast over 4 dims
  cfgo = {'priors': o.DEFAULT_PRIORS}; cfgn = {'priors': priors.DEFAULT_PRIORS}
  po, pn = o.GPParams(model=model, config=cfgo), defs.GPParams(model=model, config=cfgn)
  dso = {i: o.SubDataset(*helpers.synthetic_task(rng, 60 + 30 * i, 4)) for i in range(3)}
  dsn = {k: defs.SubDataset(v.x, v.y) for k, v in dso.items()}
  vo, go = o.nll_value_and_grad(o.constant, o.matern52, po, dso, WFO, priors_grad=o.DEFAULT_PRIORS_GRAD)
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.matern52, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
  assert gn['lengthscale'].shape == (1,)


def test_nll_grad_mlp_fp32_and_feature_dim_gt_chunk(gpu_ctx):
  """MLP-basis gradient with a feature dimension above the 16-wide LDS chunk, fp64 and fp32."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(16)
  d, f = 6, 40
  model = {'lengthscale': rng.uniform(0.5, 1.5, size=f), 'signal_variance': np.array(0.3), 'noise_variance': np.array(-2.0),
           'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': rng.normal(size=f) * 0.1}},
           'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
  cfg = {'mlp_features': (f,)}
  dso = {i: o.SubDataset(*helpers.synthetic_task(rng, 150 + 40 * i, d)) for i in range(2)}
  dsn = {k: defs.SubDataset(v.x, v.y) for k, v in dso.items()}
  vo, go = o.nll_value_and_grad(o.linear_mlp, o.matern52_mlp, o.GPParams(model=model, config=cfg), dso, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=model, config=dict(cfg)),
                                         dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  ds32 = {k: defs.SubDataset(v.x.astype(np.float32), v.y.astype(np.float32)) for k, v in dso.items()}
  v32, g32 = objectives.nll_value_and_grad(mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=to32(model), config=dict(cfg)),
                                           ds32, utils.DEFAULT_WARP_FUNC)
  assert abs(v32 - vo) <= 2e-4 * abs(vo)
  helpers.assert_grad_close(g32, go, FP32_GRAD_TOL)


def test_nll_fp32_vs_fp64_oracle(gpu_ctx):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(6)
  model = helpers.make_model(rng, 'constant', False, 4, np.float32)
  dsn = {i: defs.SubDataset(*helpers.synthetic_task(rng, 200 + 57 * i, 4, dtype=np.float32)) for i in range(3)}
  dso = {k: o.SubDataset(v.x.astype(np.float64), v.y.astype(np.float64)) for k, v in dsn.items()}
  po = o.GPParams(model=helpers.unflatten_like(model, helpers.flatten(model)))
  vo, go = o.nll_value_and_grad(o.constant, o.matern32, po, dso, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.matern32, defs.GPParams(model=model), dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 2e-4 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP32_GRAD_TOL)


# ---- factorisation, posterior, acquisition --------------------------------------------------------
CASES = [('squared_exponential', False, 'constant'), ('matern52', True, 'linear_mlp'), ('matern32', False, 'linear'),
         ('dot_product', True, 'zero')]


@pytest.mark.parametrize('kname,mlp,mname', CASES)
def test_factor_predict_acquisition_vs_oracle(gpu_ctx, kname, mlp, mname):
  defs, linalg, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(7)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  mo, mn = getattr(o, mname), getattr(mean, mname)
  x, y = helpers.synthetic_task(rng, 200, d)
  xq = rng.uniform(size=(70, d))
  cho, kio, ymo = o.solve_gp_linear_system(mo, ko, po, x, y, WFO)
  chn, kin, ymn = linalg.solve_gp_linear_system(mn, kn, pn, x, y, utils.DEFAULT_WARP_FUNC)
  assert chn.shape == (200, 200) and kin.shape == (200, 1) and ymn.shape == (200, 1)
  assert helpers.rel_err(chn, cho) < 1e-11 and helpers.rel_err(kin, kio) < 1e-9 and helpers.rel_err(ymn, ymo) < 1e-13
  muo, varo = o.predict(mo, ko, po, x, y, xq, WFO)
  mun, varn = gp.predict(mn, kn, pn, x, y, xq, utils.DEFAULT_WARP_FUNC)
  assert mun.shape == (70, 1) and varn.shape == (70, 1)
  assert helpers.rel_err(mun, muo) < 1e-9 and helpers.rel_err(varn, varo) < 1e-9
  _, covo = o.predict(mo, ko, po, x, y, xq, WFO, full_cov=True)
  mun2, covn = gp.predict(mn, kn, pn, x, y, xq, utils.DEFAULT_WARP_FUNC, full_cov=True)
  assert covn.shape == (70, 70) and helpers.rel_err(covn, covo) < 1e-9
  np.testing.assert_allclose(np.diag(covn), varn[:, 0], rtol=1e-8, atol=1e-12)   # gp_test.py:201-203
  # GP object: cache, +noise, T/(T-1), acquisition functions
  ds = {0: defs.SubDataset(x, y), 1: defs.SubDataset(x[:50], y[:50])}
  model_n = gp.GP(ds, mn, kn, pn, utils.DEFAULT_WARP_FUNC)
  mu_g, var_g = model_n.predict(xq, 0)
  assert 0 in model_n.params.cache and not model_n.params.cache[0].needs_update   # gp_test.py:186-189
  np.testing.assert_allclose(model_n.params.cache[0].chol, cho, rtol=1e-9, atol=1e-12)
  dso = {k: o.SubDataset(v.x, v.y) for k, v in ds.items()}
  mu_o, var_o = o.gp_predict_postprocess(po, dso, muo, varo, WFO, False, True, True)
  assert helpers.rel_err(var_g, var_o) < 1e-9 and helpers.rel_err(mu_g, mu_o) < 1e-9
  mu_nn, var_nn = model_n.predict(xq, 0, with_noise=False, unbiased=False)
  assert helpers.rel_err(var_nn, varo) < 1e-9
  for name, sub_o, par in [('expected_improvement', o.expected_improvement_sub, float(np.max(y))),
                           ('probability_of_improvement', o.probability_of_improvement_sub, float(np.max(y)) + 0.1),
                           ('pi2', o.probability_of_improvement_sub, float(np.max(y) + 0.1 * np.std(y))),
                           ('pi3', o.probability_of_improvement_sub, float(np.max(y)) + 0.05),
                           ('ucb', o.ucb_sub, 3.0), ('ucb2', o.ucb_sub, 2.0), ('ucb4', o.ucb_sub, 4.0)]:
    an = getattr(acfun, name)(model=model_n, sub_dataset_key=0, x_queries=xq)
    assert an.shape == (70, 1)
    ao = sub_o(mu_o, np.sqrt(var_o), par)
    assert helpers.rel_err(an, ao) < 1e-8, name
  # prior path: key not in dataset (gp.py:584-593), EI target 0.0 (acfun.py:146-147)
  mu_p, var_p = model_n.predict(xq, 'missing')
  mu_po, var_po = o.predict(mo, ko, po, None, None, xq, WFO)
  mu_po, var_po = o.gp_predict_postprocess(po, dso, mu_po, var_po, WFO, False, True, True)
  np.testing.assert_allclose(mu_p, mu_po, rtol=1e-12, atol=1e-14)
  np.testing.assert_allclose(var_p, var_po, rtol=1e-12)
  ei_p = acfun.expected_improvement(model=model_n, sub_dataset_key='missing', x_queries=xq)
  assert helpers.rel_err(ei_p, o.expected_improvement_sub(mu_po, np.sqrt(var_po), 0.0)) < 1e-9
  # cache invalidation: append -> refactor from scratch (bayesopt.py:186-190)
  model_n.update_sub_dataset((xq[:3], np.ones((3, 1))), 0, is_append=True)
  assert model_n.params.cache[0].needs_update
  mu_a, _ = model_n.predict(xq, 0)
  mu_ao, _ = o.predict(mo, ko, po, np.vstack([x, xq[:3]]), np.vstack([y, np.ones((3, 1))]), xq, WFO)
  assert helpers.rel_err(mu_a, mu_ao) < 1e-8


def test_multi_column_y_factor(gpu_ctx):
  defs, linalg, _, _, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(8)
  model = helpers.make_model(rng, 'constant', False, 2)
  po, pn = _pair(model)
  x, y = helpers.synthetic_task(rng, 150, 2, m=4)
  cho, kio, ymo = o.solve_gp_linear_system(o.constant, o.matern52, po, x, y, WFO)
  chn, kin, ymn = linalg.solve_gp_linear_system(mean.constant, kernel.matern52, pn, x, y, utils.DEFAULT_WARP_FUNC)
  assert kin.shape == (150, 4)
  assert helpers.rel_err(kin, kio) < 1e-9 and helpers.rel_err(ymn, ymo) < 1e-13


# ---- committed golden fixtures ---------------------------------------------------------------------
def _golden_cases():
  import importlib.util
  spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
  mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
  return mg.CASES


@pytest.mark.parametrize('case', _golden_cases(), ids=lambda c: c[0])
def test_against_golden_fixtures(gpu_ctx, case):
  defs, linalg, acfun, gp, kernel, mean, objectives, utils = _native()
  name, kname, mlp, mname, n, d, nq, seed = case
  ref = np.load(os.path.join(GOLDEN, name + '.npz'))
  rng = np.random.Generator(np.random.PCG64(seed))
  model = helpers.unflatten_like(helpers.make_model(rng, mname, mlp, d), ref['model_flat'])
  pn = defs.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES})
  kn = getattr(kernel, kname + ('_mlp' if mlp else '')); mn = getattr(mean, mname)
  wf = utils.DEFAULT_WARP_FUNC
  x, y, xq = ref['x'], ref['y'], ref['xq']
  np.testing.assert_allclose(kn(pn, x, warp_func=wf), ref['gram'], rtol=1e-12, atol=1e-14)
  np.testing.assert_allclose(kn(pn, x, xq, warp_func=wf), ref['cross'], rtol=1e-12, atol=1e-14)
  np.testing.assert_allclose(mn(pn, x, warp_func=wf), ref['mean_x'], rtol=1e-12, atol=1e-14)
  ds = {0: defs.SubDataset(x, y), 1: defs.SubDataset(ref['x2'], ref['y2'])}
  nll, k2 = objectives.neg_log_marginal_likelihood(mn, kn, pn, ds, wf, return_key2nll=True)
  assert abs(nll - float(ref['nll'])) <= 1e-10 * abs(float(ref['nll']))
  assert abs(k2[0] - float(ref['nll0'])) <= 1e-10 * abs(float(ref['nll0']))
  assert abs(k2[1] - float(ref['nll1'])) <= 1e-10 * abs(float(ref['nll1']))
  assert abs(nll / float(ref['nll_svd']) - 1) < 1e-6          # objectives_test.py:168
  v, g = objectives.nll_value_and_grad(mn, kn, pn, ds, wf)
  gf = helpers.flatten(g)
  helpers.assert_grad_close(g, ref['grad_flat'], FP64_GRAD_TOL, label=name)
  chol, kinvy, ymu = linalg.solve_gp_linear_system(mn, kn, pn, x, y, wf)
  assert helpers.rel_err(chol, ref['chol']) < 1e-10 and helpers.rel_err(kinvy, ref['kinvy']) < 1e-8
  mu, var = gp.predict(mn, kn, pn, x, y, xq, wf)
  _, cov = gp.predict(mn, kn, pn, x, y, xq, wf, full_cov=True)
  assert helpers.rel_err(mu, ref['mu']) < 1e-9 and helpers.rel_err(var, ref['var']) < 1e-8
  assert helpers.rel_err(cov, ref['cov']) < 1e-8
  m = gp.GP(ds, mn, kn, pn, wf)
  for nm, key in (('expected_improvement', 'ei'), ('probability_of_improvement', 'pi'), ('ucb', 'ucb')):
    a = getattr(acfun, nm)(model=m, sub_dataset_key=0, x_queries=xq)
    assert helpers.rel_err(a, ref[key]) < 1e-7, nm
  # divergence objectives and acquisition gradients (fixtures of round 1's later rows)
  al = {'al': defs.SubDataset(x, ref['y_aligned'], aligned='al'), 0: defs.SubDataset(x, y)}
  for fnc, key in ((objectives.ekl, 'ekl'), (objectives.euc, 'euc')):
    v, g = fnc.value_and_grad(mn, kn, pn, al, wf)
    assert abs(v - float(ref[key])) <= 1e-9 * max(abs(float(ref[key])), 1.0), key
    gr = ref[key + '_grad_flat']
    helpers.assert_grad_close(g, gr, 10 * FP64_GRAD_TOL, label=key)
  for fnc, key in ((acfun.expected_improvement, 'ei'), (acfun.ucb, 'ucb')):
    v, dx = fnc.value_and_grad(model=m, sub_dataset_key=0, x_queries=xq)
    assert helpers.rel_err(v, ref[key + '_value']) < 1e-7, key
    assert np.max(np.abs(dx - ref[key + '_dx'])) <= 1e-6 * max(np.max(np.abs(ref[key + '_dx'])), 1e-6), key


# ---- BASELINE sizes: oracle where it still finishes in seconds, size-independent properties above --
def _cfg2_like(rng, n, d=16):
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
  model = {'lengthscale': helpers.inv_softplus(np.full(d, np.sqrt(d) * 0.3)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(1e-2), 'constant': np.array(0.0)}
  return x, y, model


def test_cfg2_shape_n2048_vs_oracle(gpu_ctx):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  x, y, model = _cfg2_like(np.random.default_rng(2), 2048)
  vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, o.GPParams(model=model), {0: o.SubDataset(x, y)}, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model),
                                         {0: defs.SubDataset(x, y)}, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)


def test_cfg2_full_size_value_and_full_gradient(gpu_ctx):
  """BASELINE.json configs[1] at full size, N=8192, D=16 fp64 (seed 2 of SURVEY.md 8(d) = bench.cfg2_inputs()):
  (a) the NLL agrees with LAPACK's Cholesky of the same Gram matrix (oracle value path);
  (b) ALL 19 gradient leaves agree with oracle/cpu_baseline.nll_and_grad_se_ard_constant_omp -- the C/OpenMP +
      LAPACK potrf/potri port that test_oracle_pins.py pins to oracle/hyperbo_oracle.py -- to 1e-8 of max|g|;
  (c) a directional central difference of the GPU NLL itself as an oracle-free cross-check."""
  import bench
  from oracle import cpu_baseline
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  x, y, model = bench.cfg2_inputs()
  assert x.shape == (8192, 16)
  pn = defs.GPParams(model=model)
  dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
  v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC)
  vo = o.neg_log_marginal_likelihood(o.constant, o.squared_exponential, o.GPParams(model=model), {0: o.SubDataset(x, y)}, WFO)
  assert abs(v - vo) <= 1e-10 * abs(vo)
  vc, gc = cpu_baseline.nll_and_grad_se_ard_constant_omp(x, y, model)
  assert abs(v - vc) <= 1e-10 * abs(vc)
  assert set(g) == set(gc) and sum(np.size(a) for a in g.values()) == 19
  fo, fn = helpers.flatten(gc), helpers.flatten(g)
  helpers.assert_grad_close(g, gc, FP64_GRAD_TOL, label='cfg2 vs C/LAPACK port')
  rng = np.random.default_rng(2)
  x0 = helpers.flatten(model)
  direction = rng.normal(size=x0.size); direction /= np.linalg.norm(direction)
  h = 1e-5
  vals = []
  for sgn in (+1, -1):
    pm = defs.GPParams(model=helpers.unflatten_like(model, x0 + sgn * h * direction))
    vals.append(objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pm, dev, utils.DEFAULT_WARP_FUNC))
  num = (vals[0] - vals[1]) / (2 * h)
  assert abs(num - fn @ direction) <= 1e-5 * abs(num) + 1e-6
  # the 64-tile tail of the persistent bulk update (only launches of >= 600 128-tiles have one) with other numbers of reserved CUs
  # (other remainders of the last round) and as a plain launch -- same value and gradient
  try:
    for opts in ({'persist_free': 0}, {'persist_free': 48}, {'persist_free': 64}, {'persist_free': 100}):
      for k_, v_ in opts.items():
        gpu_ctx.set_option(k_, v_)
      v2, g2 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC)
      assert abs(v2 - v) <= 1e-11 * abs(v), opts
      helpers.assert_grad_close(g2, g, 1e-9, label=str(opts))
  finally:
    gpu_ctx.set_option('persist_free', -1)
  dev.close()


def test_cfg1_golden_fixture(gpu_ctx):
  """BASELINE.json configs[0] (SURVEY.md 8(d) cfg 1: seed 1, N=256, D=4, SE, fp64 NLL): the committed fixture
  (oracle, cross-checked against LAPACK by its generator) vs the device NLL, gradient, factor diagonal and K^-1 y."""
  defs, linalg, _, _, kernel, mean, objectives, utils = _native()
  fx = np.load(os.path.join(GOLDEN, 'cfg1_se_n256_d4.npz'))
  x, y = fx['x'], fx['y']
  assert x.shape == (256, 4)
  model = helpers.unflatten_like({'lengthscale': np.zeros(4), 'signal_variance': np.array(0.), 'noise_variance': np.array(0.),
                                  'constant': np.array(0.)}, fx['model_flat'])
  pn = defs.GPParams(model=model)
  wf = utils.DEFAULT_WARP_FUNC
  v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, {0: defs.SubDataset(x, y)}, wf)
  assert abs(v - float(fx['nll'])) <= 1e-10 * abs(float(fx['nll']))
  assert abs(v - float(fx['nll_lapack'])) <= 1e-10 * abs(float(fx['nll_lapack']))
  helpers.assert_grad_close(g, fx['grad_flat'], FP64_GRAD_TOL, label='cfg1')
  chol, kinvy, _ = linalg.solve_gp_linear_system(mean.constant, kernel.squared_exponential, pn, x, y, wf)
  assert helpers.rel_err(np.diag(chol), fx['chol_diag']) < 1e-11 and helpers.rel_err(kinvy, fx['kinvy']) < 1e-8
  vs = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, {0: defs.SubDataset(x, y)}, wf,
                                              use_cholesky=False)
  assert abs(vs - float(fx['nll_svd'])) <= 1e-9 * abs(float(fx['nll_svd']))


def test_cfg4_all_64_tasks_vs_oracle_fixture(gpu_ctx):
  """BASELINE.json configs[3]: the 64 ragged sub-datasets of bench.cfg4_inputs() in ONE batched evaluation (T=64, the
  z-rotated tile rows of the batched kernels) -- per-task NLL, mean NLL and mean gradient against the oracle's
  committed outputs (tests/golden/make_golden_configs.py), plus the oracle live on four of the tasks."""
  import bench
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  fx = np.load(os.path.join(GOLDEN, 'cfg4_t64_oracle.npz'))
  data, raw = bench.cfg4_inputs()
  assert len(data) == 64 and np.array_equal(fx['sizes'], [len(data[k][0]) for k in sorted(data)])
  assert np.array_equal(fx['model_flat'], helpers.flatten(raw))
  ds = {k: defs.SubDataset(xx, yy) for k, (xx, yy) in data.items()}
  pn = defs.GPParams(model=raw)
  wf = utils.DEFAULT_WARP_FUNC
  dev = objectives.DeviceDataset(ds)
  total, key2nll = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, dev, wf, return_key2nll=True)
  per = np.array([key2nll[k] for k in sorted(data)])
  assert np.max(np.abs(per - fx['nll_per_task']) / np.abs(fx['nll_per_task'])) <= 1e-10
  assert abs(total - float(fx['nll_mean'])) <= 1e-10 * abs(float(fx['nll_mean']))
  v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, wf)
  assert abs(v - float(fx['nll_mean'])) <= 1e-10 * abs(float(fx['nll_mean']))
  gf = helpers.flatten(g)
  helpers.assert_grad_close(g, fx['grad_mean_flat'], FP64_GRAD_TOL, label='cfg4')
  dev.close()
  po = o.GPParams(model=raw)
  for k in (0, 21, 42, 63):   # the fixture is not stale: live oracle on a sample
    vo, _ = o.nll_sub_dataset_value_and_grad(o.constant, o.squared_exponential, po, data[k][0], data[k][1], WFO)
    assert abs(vo - fx['nll_per_task'][k]) <= 1e-11 * abs(vo)
    assert abs(per[k] - vo) <= 1e-10 * abs(vo)


def test_cfg4_like_ragged_multitask_vs_oracle(gpu_ctx):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(4)
  d = 4
  sizes = rng.integers(1600, 2401, size=6)
  model = {'lengthscale': helpers.inv_softplus(np.full(d, 0.4)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(1e-2), 'constant': np.array(0.1)}
  dso = {i: o.SubDataset(*helpers.synthetic_task(rng, int(n), d)) for i, n in enumerate(sizes)}
  dsn = {k: defs.SubDataset(v.x, v.y) for k, v in dso.items()}
  vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, o.GPParams(model=model), dso, WFO)
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model), dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)


def test_cfg3_shape_fp32_ei_against_fp64(gpu_ctx):
  """Matern-5/2 on tanh-MLP(32->64) features + linear_mlp mean, N=16384, fp32 factor + EI over 65536
  candidates; checked against the same computation in fp64 on the GPU for a 4096-candidate subset
  (and that against the oracle on 256 of them at N=2048)."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(3)
  d, f = 32, 64
  model = {'lengthscale': helpers.inv_softplus(np.ones(f)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(1e-2),
           'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
           'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
  cfg = {'mlp_features': (f,)}
  n, M = 16384, 65536
  x = rng.uniform(size=(n, d)); y = np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))
  xq = rng.uniform(size=(M, d))
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  g32 = gp.GP({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}, mean.linear_mlp, kernel.matern52_mlp,
              defs.GPParams(model=to32(model), config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  ei32 = acfun.expected_improvement(model=g32, sub_dataset_key=0, x_queries=xq.astype(np.float32))
  assert ei32.shape == (M, 1) and ei32.dtype == np.float32 and np.isfinite(ei32).all() and (ei32 >= 0).all()
  g64 = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp,
              defs.GPParams(model=model, config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  sub = slice(0, 4096)
  ei64 = acfun.expected_improvement(model=g64, sub_dataset_key=0, x_queries=xq[sub])
  mu32, var32 = g32.predict(xq[sub].astype(np.float32), 0)
  mu64, var64 = g64.predict(xq[sub], 0)
  assert np.max(np.abs(mu32 - mu64)) < 5e-3 * (1 + np.max(np.abs(mu64)))
  assert np.max(np.abs(var32 - var64)) < 5e-3 * np.max(np.abs(var64))
  assert np.max(np.abs(ei32[sub] - ei64)) < 5e-3 * (np.max(np.abs(ei64)) + 1e-3)
  # fp64 path vs the oracle at a size it finishes in seconds
  ns = 2048
  gs = gp.GP({0: defs.SubDataset(x[:ns], y[:ns])}, mean.linear_mlp, kernel.matern52_mlp,
             defs.GPParams(model=model, config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  ei_s = acfun.expected_improvement(model=gs, sub_dataset_key=0, x_queries=xq[:256])
  po = o.GPParams(model=model, config=dict(cfg))
  mu_o, var_o = o.predict(o.linear_mlp, o.matern52_mlp, po, x[:ns], y[:ns], xq[:256], WFO)
  mu_o, var_o = o.gp_predict_postprocess(po, {0: o.SubDataset(x[:ns], y[:ns])}, mu_o, var_o, WFO, False, True, True)
  assert helpers.rel_err(ei_s, o.expected_improvement_sub(mu_o, np.sqrt(var_o), float(np.max(y[:ns])))) < 1e-7


def test_cfg3_full_size_against_host_lapack(gpu_ctx):
  """cfg 3 at its FULL size against something that is not this library: Gram, mean and cross-Gram come from the device
  (hbo_gram / hbo_mean, each tested against the oracle elsewhere), then host LAPACK in fp64 does what the reference does --
  dpotrf, cho_solve, solve_triangular (gp.py:286-305), EI (acfun.py:96-110).  The GPU fp64 posterior has to match that to
  1e-8, and the fp32 cache (bf16x3 product on the matrix cores, explicit W = L^-1 instead of a triangular solve) to the
  tolerance the fp32 tests state.  N = 16384, 4096 of the 65 536 candidates (the host solve is N^2 M flops)."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(3)
  d, f = 32, 64
  model = {'lengthscale': helpers.inv_softplus(np.ones(f)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(1e-2),
           'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
           'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
  cfg = {'mlp_features': (f,)}
  n, M = 16384, 4096
  x = rng.uniform(size=(n, d)); y = np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))
  xq = rng.uniform(size=(M, d))
  p64 = defs.GPParams(model=model, config=dict(cfg))
  # ---- host LAPACK, fp64
  k = np.asarray(kernel.matern52_mlp(p64, x, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64)
  noise = float(np.log1p(np.exp(model['noise_variance'])) + 1e-10)
  k[np.diag_indices(n)] += noise + 1e-6
  r = y - np.asarray(mean.linear_mlp(p64, x, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64)
  chol = spla.cholesky(k, lower=True, overwrite_a=True, check_finite=False)
  del k
  alpha = spla.cho_solve((chol, True), r, check_finite=False)
  kxq = np.asarray(kernel.matern52_mlp(p64, x, xq, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64)      # [n, M]
  mu_ref = np.asarray(mean.linear_mlp(p64, xq, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64) + kxq.T @ alpha
  v = spla.solve_triangular(chol, kxq, lower=True, overwrite_b=True, check_finite=False)
  del chol
  var_ref = (np.asarray(kernel.matern52_mlp(p64, xq, warp_func=utils.DEFAULT_WARP_FUNC, diag=True), dtype=np.float64).reshape(-1, 1)
             - np.sum(v * v, axis=0)[:, None])
  del v, kxq
  var_ref_n = var_ref + noise                                   # GP.predict adds the noise (gp.py:607-613)
  ei_ref = o.expected_improvement_sub(mu_ref, np.sqrt(var_ref_n), float(np.max(y)))
  # ---- GPU fp64
  g64 = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp, p64, utils.DEFAULT_WARP_FUNC)
  mu64, var64 = g64.predict(xq, 0)
  ei64 = acfun.expected_improvement(model=g64, sub_dataset_key=0, x_queries=xq)
  assert np.max(np.abs(mu64 - mu_ref)) < 1e-8 * (1 + np.max(np.abs(mu_ref)))
  assert np.max(np.abs(var64 - var_ref_n)) < 1e-8 * np.max(np.abs(var_ref_n))
  assert np.max(np.abs(ei64 - ei_ref)) < 1e-7 * (np.max(np.abs(ei_ref)) + 1e-3)
  del g64
  # ---- GPU fp32 (the reference's default dtype): bf16x3 product and the fp32-MFMA product
  to32 = lambda t: {k_: to32(v_) for k_, v_ in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  g32 = gp.GP({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}, mean.linear_mlp, kernel.matern52_mlp,
              defs.GPParams(model=to32(model), config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  # three fp32 forms of the posterior product + factorisation: the default two-way fp16 split (f16x2: 3 MFMAs per product), the exact
  # three-way bf16 split (bf16x3: hbo_tune post_f16x2 = chol_f16x2 = 0, 6 MFMAs = strict fp32 products) and the fp32-MFMA product.
  # Bounds = 10 x the errors measured at full size in round 6 (bench.py cfg3.paths, vs fp64: |d mu| 1.7e-5 on a scale of 3.5,
  # |d var| 2e-6 on 0.86, |d EI| 1.1e-5 on 1.3 -- the same for f16x2 and bf16x3): mean 5e-5, variance 3e-5, EI 1e-4, relative as below
  # (round 5 asserted 5e-3 for all three).
  TOL_MU, TOL_VAR, TOL_EI = 5e-5, 3e-5, 1e-4
  errs = {}
  forms = {'f16x2': {}, 'bf16x3': {'post_f16x2': 0, 'chol_f16x2': 0}, 'fp32_mfma': {'post_bf16x3': 0}}
  defaults = {'post_f16x2': 1, 'chol_f16x2': 1, 'post_bf16x3': 1}
  try:
    for name, opts in forms.items():
      for k_, v_ in dict(defaults, **opts).items():
        gpu_ctx.set_option(k_, v_)
      g32.update_model_params(g32.params.model)     # drops the cache: the factorisation runs in this form too
      mu32, var32 = g32.predict(xq.astype(np.float32), 0)
      ei32 = acfun.expected_improvement(model=g32, sub_dataset_key=0, x_queries=xq.astype(np.float32))
      errs[name] = (np.max(np.abs(mu32 - mu_ref)) / (1 + np.max(np.abs(mu_ref))), np.max(np.abs(var32 - var_ref_n)) / np.max(np.abs(var_ref_n)),
                    np.max(np.abs(ei32 - ei_ref)) / (np.max(np.abs(ei_ref)) + 1e-3))
      assert errs[name][0] < TOL_MU and errs[name][1] < TOL_VAR and errs[name][2] < TOL_EI, errs
  finally:
    for k_, v_ in defaults.items():
      gpu_ctx.set_option(k_, v_)
  if os.environ.get('HBO_GRAD_LOG'):
    with open(os.environ['HBO_GRAD_LOG'], 'a') as f_:
      f_.write('0 tol=cfg3_full_size (mu, var, ei) relative errors vs host LAPACK fp64: %r\n' % ({k_: tuple(float('%.3e' % e) for e in v_) for k_, v_ in errs.items()},))
  # the split products are as accurate as the fp32-MFMA one against the INDEPENDENT reference too
  for name in ('f16x2', 'bf16x3'):
    assert errs[name][1] <= 1.5 * errs['fp32_mfma'][1] + 1e-6 and errs[name][0] <= 1.5 * errs['fp32_mfma'][0] + 1e-6, errs


@pytest.mark.parametrize('noise_target', [1e-3, 1e-6])
def test_ill_conditioned_fp32_cache_explicit_inverse_vs_triangular_solve(gpu_ctx, noise_target):
  """The posterior uses V = W Kxq with the explicit W = L^-1 where the reference calls solve_triangular (gp.py:297).  On
  ill-conditioned fp32 problems (SE kernel, N = 8192, noise 1e-3: kappa ~ 1e7, at the edge of what fp32 resolves; noise 1e-6:
  beyond it) the explicit-inverse route must not be materially worse than the backward-stable route IN THE SAME PRECISION --
  host LAPACK spotrf + strtrs in fp32 -- both measured against host LAPACK in fp64; and where LAPACK's fp32 factorisation
  itself breaks down (not positive definite to working precision) the library reports NaN like jax.scipy.linalg.cholesky
  (linalg.py:29-33), it does not return numbers."""
  defs, _, _, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(19)
  n, M, d = 8192, 512, 6
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.01 * rng.normal(size=(n, 1))
  xq = rng.uniform(size=(M, d))
  model = {'lengthscale': helpers.inv_softplus(np.full(d, 0.6)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(noise_target), 'constant': np.array(0.0)}
  p64 = defs.GPParams(model=model)
  noise = float(np.log1p(np.exp(model['noise_variance'])) + 1e-10)
  sv = float(np.log1p(np.exp(model['signal_variance'])) + 1e-10)      # k(x, x) of the SE kernel
  k = np.asarray(kernel.squared_exponential(p64, x, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64)
  kxq = np.asarray(kernel.squared_exponential(p64, x, xq, warp_func=utils.DEFAULT_WARP_FUNC), dtype=np.float64)
  kd = k + (noise + 1e-6) * np.eye(n)
  def host(dtype):
    c = spla.cholesky(kd.astype(dtype), lower=True, check_finite=False)
    a = spla.cho_solve((c, True), y.astype(dtype), check_finite=False)
    v = spla.solve_triangular(c, kxq.astype(dtype), lower=True, check_finite=False)
    return (kxq.astype(dtype).T @ a).astype(np.float64), (sv - np.sum(v.astype(np.float64)**2, axis=0))[:, None]
  mu_ref, var_ref = host(np.float64)
  to32 = lambda t: {k_: to32(v_) for k_, v_ in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  mu_g, var_g = gp.predict(mean.constant, kernel.squared_exponential, defs.GPParams(model=to32(model)), x.astype(np.float32),
                           y.astype(np.float32), xq.astype(np.float32), warp_func=utils.DEFAULT_WARP_FUNC)
  mu_g, var_g = np.asarray(mu_g, dtype=np.float64), np.asarray(var_g, dtype=np.float64)
  try:
    mu_h32, var_h32 = host(np.float32)
  except np.linalg.LinAlgError:
    # fp32 cannot factor this matrix: NaN from the library as from the reference's jnp cholesky -- or, if the blocked
    # algorithm's different rounding got through, at least nothing worse than a finite estimate of the right magnitude
    assert np.isnan(mu_g).all() or np.max(np.abs(mu_g - mu_ref)) < 1.0
    return
  e_mu_h, e_mu_g = np.max(np.abs(mu_h32 - mu_ref)), np.max(np.abs(mu_g - mu_ref))
  e_var_h, e_var_g = np.max(np.abs(var_h32 - var_ref)), np.max(np.abs(var_g - var_ref))
  assert np.isfinite(e_mu_g) and e_mu_g <= 10 * e_mu_h + 1e-3, (e_mu_g, e_mu_h)
  assert e_var_g <= 10 * e_var_h + 1e-3, (e_var_g, e_var_h)
  print(f'noise {noise_target}: |mu - fp64| GPU fp32 {e_mu_g:.3e} vs LAPACK fp32 {e_mu_h:.3e}; |var - fp64| {e_var_g:.3e} vs {e_var_h:.3e}')


@pytest.mark.parametrize('n', [1500, 4600])
def test_fp32_factorisation_on_bf16_matrix_cores_is_as_accurate_as_fp32_mfma(gpu_ctx, n):
  """The fp32 trailing updates (and, above 32 blocks, the products of the inverse) run on the bf16 matrix cores from exact
  three-way splits of their operands (post3.hip: syrk3_kernel), and so does K^-1 = W^T W.  On a matrix of condition number ~1e6 the factor, the solve
  and the inverse must stay in the accuracy class of the fp32-MFMA kernels against fp64 LAPACK (linalg.py:29-33 in the
  reference's default dtype).  Measured: well-conditioned matrices come out 3-4x CLOSER to fp64 than with fp32 MFMA (fewer
  accumulation roundings: 3.7e-6 -> 8.4e-7 at n = 4224, tools/dbg32.py); at kappa = 1e6 the three dropped cross products
  (weight 2^-24 each, the size of an fp32 rounding) show and the factor is ~2x further (5.7e-5 against 3.0e-5 at n = 1500) --
  both two orders below kappa * eps.  The bound here is 3x."""
  _, linalg, *_ = _native()
  rng = np.random.default_rng(n)
  q_, _ = np.linalg.qr(rng.normal(size=(n, n)))
  ev = np.logspace(0, -6, n)                       # kappa = 1e6
  a64 = (q_ * ev) @ q_.T
  a64 = 0.5 * (a64 + a64.T)
  a = a64.astype(np.float32)
  b = rng.normal(size=(n, 2)).astype(np.float32)
  a64 = a.astype(np.float64)
  cref = spla.cholesky(a64, lower=True)
  xref = spla.cho_solve((cref, True), b.astype(np.float64))
  iref = np.linalg.inv(a64)
  err = {}
  try:
    for name, on in (('mfma', 0), ('bf16x3', 1)):
      gpu_ctx.set_option('bf16x3', on)
      chol, x = linalg.solve_linear_system(a, b)
      inv, _ = linalg.spd_inverse(a)
      assert np.isfinite(chol).all()
      err[name] = (helpers.rel_err(chol, cref), helpers.rel_err(x, xref), helpers.rel_err(inv, iref))
  finally:
    gpu_ctx.set_option('bf16x3', 1)
  for k in range(3):
    assert err['bf16x3'][k] <= 3.0 * err['mfma'][k] + 1e-7, err
  assert err['bf16x3'][0] < 1e-3        # kappa * eps_fp32 ~ 6e-2 bounds the solve; the factor itself stays accurate


@pytest.mark.parametrize('n', [4225, 5633, 7300])
def test_fp32_bf16x3_paths_at_odd_block_counts(gpu_ctx, n):
  """34, 45 and 58 blocks (the last: panel groups of six, the last one cut): partial groups at several levels of the inverse, an odd number of row tiles in K^-1 = W^T W and in the
  trailing updates' trapezoids (syrk3_kernel modes 0-3, split3_* edge handling).  Factor, solve and inverse of a
  well-conditioned fp32 matrix against fp64 LAPACK (linalg.py:29-33); measured 1e-6 / 5e-6 / 5e-6 (tools/sweep32.py: nine sizes
  from 33 to 66 blocks, 2-3x closer to fp64 than the fp32-MFMA kernels)."""
  _, linalg, *_ = _native()
  rng = np.random.default_rng(n)
  g = rng.normal(size=(n, 64)).astype(np.float32)
  a = (g @ g.T / 64 + 0.5 * np.eye(n, dtype=np.float32)).astype(np.float32)
  a = 0.5 * (a + a.T)
  b = rng.normal(size=(n, 1)).astype(np.float32)
  a64 = a.astype(np.float64)
  cref = spla.cholesky(a64, lower=True)
  iref = spla.cho_solve((cref, True), np.eye(n))
  chol, x = linalg.solve_linear_system(a, b)
  inv, _ = linalg.spd_inverse(a)
  assert helpers.rel_err(chol, cref) <= 2e-5
  assert helpers.rel_err(x, iref @ b.astype(np.float64)) <= 1e-4
  assert helpers.rel_err(inv, iref) <= 1e-4
  assert helpers.rel_err(inv, inv.T) <= 1e-6     # (the diagonal 128-blocks are computed whole: symmetric to rounding)


def test_fp32_objective_beyond_32_blocks_on_bf16_matrix_cores(gpu_ctx):
  """fp32 NLL + gradient of one matrix of 36 blocks (above small_nblk: trailing updates, the upper levels of the inverse AND
  K^-1 = W^T W -- syrk3_kernel modes 0-3 -- on the bf16 matrix cores) against the fp64 path, and against the same evaluation
  with every product on fp32 MFMA (option bf16x3 = 0): the errors must be of the same size (objectives.py:109-210 in the
  reference's default dtype).  Measured at N = 8192 / 16384: identical to three digits (tools/check_lauum3.py)."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(36)
  n, d = 4500, 6
  model = helpers.make_model(rng, 'constant', False, d)
  x, y = helpers.synthetic_task(rng, n, d)
  v64, g64 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model),
                                           {0: defs.SubDataset(x, y)}, utils.DEFAULT_WARP_FUNC)
  f64 = helpers.flatten(g64)
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  ds32 = {0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}
  err = {}
  try:
    for on in (0, 1):
      gpu_ctx.set_option('bf16x3', on)
      v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=to32(model)), ds32,
                                           utils.DEFAULT_WARP_FUNC)
      err[on] = (abs(v - v64) / max(abs(v64), 1.0), np.max(np.abs(helpers.flatten(g) - f64)) / np.max(np.abs(f64)))
  finally:
    gpu_ctx.set_option('bf16x3', 1)
  assert err[1][0] <= 2e-4 and err[1][1] <= 5e-3, err           # the full-size fp32 tolerances of this file
  assert err[1][0] <= 3 * err[0][0] + 1e-6 and err[1][1] <= 3 * err[0][1] + 1e-6, err


@pytest.mark.parametrize('case', ['plain', 'large_targets', 'small_noise', 'matern_mlp', 'batch', 'ekl', 'dot'])
def test_fp32_factorisation_on_two_way_fp16_splits(gpu_ctx, case):
  """fp32 evaluations of the stationary covariances run the trailing updates, the upper levels of the inverse and K^-1 = W^T W
  on the fp16 matrix cores from two-way splits scaled by powers of two (hbo_tune chol_f16x2, default on; post3.hip:
  syrk3_kernel<true>): the factor's entries by the a-priori bound sqrt(max A_ii), W / S21 by measured maxima, the augmented
  tile-row z = L^-1 r per 16 rows x 64 columns.  Against the fp64 evaluation (objectives.py:109-210 in the reference's default
  dtype) the result must be as close as the exact three-way bf16 form (chol_f16x2 = 0) -- with targets of magnitude 1e4 (the
  augmented rows' range), a noise variance of 1e-4 on a smooth kernel (large entries of L^-1; at 1e-5 the fp32 matrix is no longer positive definite on any path), MLP features, a batch (one set of
  augmented-row scales per task), the divergence objective (several augmented rows) -- and the dot-product kernel, whose
  diagonal is not constant, must not take the path at all."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(21)
  d = 5
  n = 4500 if case != 'batch' else 0
  mean_f, kern_f, mlp = mean.constant, kernel.squared_exponential, False
  if case == 'matern_mlp': mean_f, kern_f, mlp = mean.linear_mlp, kernel.matern52_mlp, True
  if case == 'dot': kern_f = kernel.dot_product
  model = helpers.make_model(rng, 'linear_mlp' if mlp else 'constant', mlp, d)
  if case == 'small_noise':
    model['noise_variance'] = np.array(helpers.inv_softplus(1e-4)); model['lengthscale'] = model['lengthscale'] * 0 + helpers.inv_softplus(0.8)
  if case == 'batch':
    data = {k: defs.SubDataset(*helpers.synthetic_task(rng, m, d)) for k, m in enumerate((700, 1300, 515))}
  elif case == 'ekl':
    x, y = helpers.synthetic_task(rng, 1800, d, m=5)
    data = {0: defs.SubDataset(x, y, aligned=0)}
  else:
    x, y = helpers.synthetic_task(rng, n, d)
    if case == 'large_targets': y = 1e4 * y + 3e4
    data = {0: defs.SubDataset(x, y)}
  def evaluate(dt, params):
    ds = {k: defs.SubDataset(np.asarray(v.x, dtype=dt), np.asarray(v.y, dtype=dt), aligned=v.aligned) for k, v in data.items()}
    if case == 'ekl':
      return objectives.ekl.value_and_grad(mean_f, kern_f, defs.GPParams(model=params, config={'mlp_features': helpers.MLP_FEATURES}), ds, utils.DEFAULT_WARP_FUNC)
    return objectives.nll_value_and_grad(mean_f, kern_f, defs.GPParams(model=params, config={'mlp_features': helpers.MLP_FEATURES}), ds, utils.DEFAULT_WARP_FUNC)
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  v64, g64 = evaluate(np.float64, model)
  f64 = helpers.flatten(g64)
  out, err = {}, {}
  try:
    for on in (0, 1):
      gpu_ctx.set_option('chol_f16x2', on)
      v, g = evaluate(np.float32, to32(model))
      out[on] = (v, helpers.flatten(g))
      err[on] = (abs(v - v64) / max(abs(v64), 1.0), np.max(np.abs(out[on][1] - f64)) / np.max(np.abs(f64)))
  finally:
    gpu_ctx.set_option('chol_f16x2', 1)
  assert np.isfinite(out[1][0]) and np.isfinite(out[1][1]).all()
  if case == 'dot':
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])      # not a constant diagonal: the option does not apply
    return
  assert out[0][0] != out[1][0] or not np.array_equal(out[0][1], out[1][1])     # the other path did run
  assert err[1][0] <= 3 * err[0][0] + 2e-6 and err[1][1] <= 3 * err[0][1] + 2e-6, err
  print(case, 'fp32 vs fp64 (value, gradient / max): bf16x3', err[0], 'f16x2', err[1])


def test_largest_single_matrix_closed_form(gpu_ctx):
  """N = 131072 fp64: the Gram matrix (137 GB) is factorised in place on one GPU's 288 GB -- the largest single matrix the path
  takes without tiling over devices (1024 blocks: every tile / block index at its maximum).  Same closed form as the cfg-5 test
  below.  Measured: relative error 1.5e-15, 11.8 s for the factorisation = 63 TFLOP/s = 81 % of the fp64 MFMA peak
  (tools/big_n.py).  Skipped on a device with less than 200 GB."""
  import ctypes as C
  from hyperbo_amd import _native as nat
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  name = C.create_string_buffer(128); cus = C.c_int32(0); mem = C.c_int64(0)
  if nat.lib().hbo_device_info(gpu_ctx.device, name, 128, C.byref(cus), C.byref(mem)) != 0 or mem.value < 200e9:
    pytest.skip('needs a device with at least 200 GB')
  rng = np.random.default_rng(7)
  n = 131072
  x = rng.uniform(-1, 1, size=(n, 1)); y = rng.normal(size=(n, 1))
  sigma, bias, noise = 0.7, 0.3, 0.1
  model = {'dot_prod_sigma': np.array(sigma), 'dot_prod_bias': np.array(bias), 'noise_variance': np.array(noise)}
  v = objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, defs.GPParams(model=model), {0: defs.SubDataset(x, y)})
  c = noise + 1e-6
  U = np.hstack([x / sigma, np.full((n, 1), bias)])
  cap = np.eye(2) + U.T @ U / c
  logdet = n * np.log(c) + np.linalg.slogdet(cap)[1]
  uty = U.T @ y
  quad = ((y.T @ y).item() - (uty.T @ np.linalg.solve(cap, uty)).item() / c) / c
  expect = 0.5 * quad + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi)
  assert abs(v - expect) <= 1e-9 * abs(expect)


def test_cfg5_full_size_closed_form(gpu_ctx):
  """N=65536 fp64 blocked Cholesky (32 GiB Gram, HBM-bound panels).  The dot-product kernel on 1-D
  inputs gives K = x x^T / s^2 + b^2 + c I, whose log-determinant and quadratic form have closed
  forms (matrix determinant lemma / Woodbury): a size-independent check of the whole pipeline."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(5)
  n = 65536
  x = rng.uniform(-1, 1, size=(n, 1)); y = rng.normal(size=(n, 1))
  sigma, bias, noise = 0.7, 0.3, 0.1
  model = {'dot_prod_sigma': np.array(sigma), 'dot_prod_bias': np.array(bias), 'noise_variance': np.array(noise)}
  v = objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, defs.GPParams(model=model), {0: defs.SubDataset(x, y)})
  c = noise + 1e-6
  U = np.hstack([x / sigma, np.full((n, 1), bias)])
  cap = np.eye(2) + U.T @ U / c
  logdet = n * np.log(c) + np.linalg.slogdet(cap)[1]
  uty = U.T @ y
  quad = ((y.T @ y).item() - (uty.T @ np.linalg.solve(cap, uty)).item() / c) / c
  expect = 0.5 * quad + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi)
  assert abs(v - expect) <= 1e-9 * abs(expect)


def test_cfg5_se_ard_full_size_residual_and_schedules(gpu_ctx):
  """BASELINE.json configs[4] with its STATED workload: SE-ARD, D=16, N=65536 fp64 (seed 5, ls = 0.3 sqrt(D), sv = 1,
  noise 1e-1; a full-rank 32 GiB Gram matrix).  The oracle cannot run at this size, so size-independent properties:
  (a) alpha = K^-1 (y - mu) from the blocked factorisation solves the system: on eight sampled 1024-row chunks,
      |K[rows,:] alpha - (y - mu)[rows]| <= 1e-10 * (|K[rows,:]| |alpha|)  with K rows rebuilt through hbo_gram;
  (b) the NLL out of hbo_nll is identical (1e-11) under a different blocking and schedule (three panels per update,
      no look-ahead): every tile is accumulated in another order, the result may not move;
  (c) the same pipeline against LAPACK on the leading N=16384 sub-problem (value incl. log-determinant)."""
  defs, linalg, _, _, kernel, mean, objectives, utils = _native()
  from hyperbo_amd import _native as nat
  rng = np.random.Generator(np.random.PCG64(5))
  n, d = 65536, 16
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
  model = {'lengthscale': helpers.inv_softplus(np.full(d, np.sqrt(d) * 0.3)), 'signal_variance': helpers.inv_softplus(1.0),
           'noise_variance': helpers.inv_softplus(1e-1), 'constant': np.array(0.0)}
  pn = defs.GPParams(model=model)
  wf = utils.DEFAULT_WARP_FUNC
  ctx = nat.default_context()
  dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
  v = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, dev, wf)
  assert np.isfinite(v)
  ctx.set_option('potrf_group', 3); ctx.set_option('lookahead', 0)
  try:
    v2 = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, dev, wf)
  finally:
    ctx.set_option('potrf_group', 0); ctx.set_option('lookahead', 1)
  assert abs(v - v2) <= 1e-11 * abs(v)
  dev.close()
  h = linalg.factor(mean.constant, kernel.squared_exponential, pn, x, y, wf)
  try:
    assert h.status == 0
    kinvy = np.empty((n, 1)); ymu = np.empty((n, 1))
    ctx.check(nat.lib().hbo_cache_export(ctx.handle, h.handle, None, nat.ptr(kinvy), nat.ptr(ymu)))
  finally:
    h.close()
  noise = 1e-1 + 1e-10 + 1e-6
  for c in rng.choice(n // 1024, size=8, replace=False):
    rows = slice(int(c) * 1024, int(c) * 1024 + 1024)
    kr = kernel.squared_exponential(pn, x[rows], x, warp_func=wf)
    resid = kr @ kinvy + noise * kinvy[rows] - ymu[rows]
    bound = np.abs(kr) @ np.abs(kinvy) + noise * np.abs(kinvy[rows])
    assert np.max(np.abs(resid) / bound) <= 1e-10, float(np.max(np.abs(resid) / bound))
  quad = float(ymu[:, 0] @ kinvy[:, 0])
  assert quad > 0
  # (c) leading sub-problem against LAPACK
  ns = 16384
  vs = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, {0: defs.SubDataset(x[:ns], y[:ns])}, wf)
  ks = kernel.squared_exponential(pn, x[:ns], warp_func=wf)
  ks[np.diag_indices(ns)] += noise
  c, info = spla.lapack.dpotrf(ks, lower=1, overwrite_a=1)
  assert info == 0
  al = spla.cho_solve((c, True), y[:ns])
  ref = float(0.5 * (y[:ns].T @ al)[0, 0] + np.sum(np.log(np.diag(c))) + 0.5 * ns * np.log(2 * np.pi))
  assert abs(vs - ref) <= 1e-10 * abs(ref)


# ---- SVD NLL (objectives.py:157-176), GP.stats (gp.py:487-533), HGP (gp.py:623-682, acfun.py:72-82) ------------
def test_svd_nll_and_gp_stats_vs_oracle(gpu_ctx):
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  import functools
  rng = np.random.default_rng(21)
  d = 3
  model = helpers.make_model(rng, 'linear', False, d)
  po, pn = _pair(model)
  dso = {'a': o.SubDataset(*helpers.synthetic_task(rng, 70, d)), 'b': o.SubDataset(*helpers.synthetic_task(rng, 33, d, m=3)),
         'al': o.SubDataset(*helpers.synthetic_task(rng, 40, d, m=6), aligned=1), 'e': o.SubDataset(np.zeros((0, d)), np.zeros((0, 1)))}
  dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
  wf = utils.DEFAULT_WARP_FUNC
  for exclude in (True, False):
    vo, ko = o.neg_log_marginal_likelihood(o.linear, o.matern32, po, dso, WFO, exclude_aligned=exclude, return_key2nll=True, use_cholesky=False)
    vn, kn = objectives.neg_log_marginal_likelihood(mean.linear, kernel.matern32, pn, dsn, wf, exclude_aligned=exclude,
                                                    return_key2nll=True, use_cholesky=False)
    assert set(kn) == set(ko) and abs(vn - vo) <= 1e-9 * abs(vo)
    for k in ko:
      assert abs(kn[k] - ko[k]) <= 1e-9 * abs(ko[k])
    vc = objectives.neg_log_marginal_likelihood(mean.linear, kernel.matern32, pn, dsn, wf, exclude_aligned=exclude)
    assert abs(vn / vc - 1) < 1e-6                      # objectives_test.py:168: SVD and Cholesky variants agree
  m = gp.GP(dsn, mean.linear, kernel.matern32, pn, wf)
  nll, key2nll = m.neg_log_marginal_likelihood()       # the reference's SVD variant (gp.py:487-497)
  vo, _ = o.neg_log_marginal_likelihood(o.linear, o.matern32, po, dso, WFO, return_key2nll=True, use_cholesky=False)
  assert abs(nll - vo) <= 1e-9 * abs(vo) and set(key2nll) == {'a', 'b'}
  nll_s, ekl, ekl_partial, euc, k2 = m.stats(verbose=False)
  assert nll_s == nll and k2 == key2nll
  kl_full = functools.partial(o.kl_multivariate_normal, eps=1e-6, partial=False)
  kl_part = functools.partial(o.kl_multivariate_normal, eps=1e-6, partial=True)
  assert abs(ekl - o.multivariate_normal_divergence(o.linear, o.matern32, po, dso, WFO, distance=kl_full)) <= 1e-7 * abs(ekl)
  assert abs(ekl_partial - o.multivariate_normal_divergence(o.linear, o.matern32, po, dso, WFO, distance=kl_part)) <= 1e-8 * abs(ekl_partial)
  assert abs(euc - o.multivariate_normal_divergence(o.linear, o.matern32, po, dso, WFO, distance=o.euclidean_multivariate_normal)) <= 1e-9 * abs(euc)
  # a numerically singular covariance: Cholesky reports NaN, the SVD variant stays finite (why the reference has it)
  xr = np.repeat(rng.uniform(size=(5, d)), 30, axis=0)
  sing = {0: defs.SubDataset(xr, np.sin(xr[:, :1]))}
  pz = defs.GPParams(model={'dot_prod_sigma': np.array(1.0), 'dot_prod_bias': np.array(0.0), 'noise_variance': np.array(-1e-6)})
  assert np.isnan(objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, pz, sing))
  vs = objectives.neg_log_marginal_likelihood(mean.zero, kernel.dot_product, pz, sing, use_cholesky=False)
  vso = o.neg_log_marginal_likelihood(o.zero, o.dot_product, o.GPParams(model=pz.model), {0: o.SubDataset(xr, np.sin(xr[:, :1]))}, use_cholesky=False)
  assert (np.isfinite(vs) and abs(vs - vso) <= 1e-6 * abs(vso)) or (np.isnan(vs) and np.isnan(vso)) or (np.isinf(vs) and np.isinf(vso))


def test_hgp_acquisition_is_mean_over_parameter_samples(gpu_ctx):
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(22)
  d = 3
  samples = [helpers.make_model(np.random.default_rng(100 + i), 'constant', False, d) for i in range(3)]
  x, y = helpers.synthetic_task(rng, 90, d)
  x2, y2 = helpers.synthetic_task(rng, 50, d)
  xq = rng.uniform(size=(37, d))
  wf = utils.DEFAULT_WARP_FUNC
  dsn = {0: defs.SubDataset(x, y), 1: defs.SubDataset(x2, y2)}
  dso = {0: o.SubDataset(x, y), 1: o.SubDataset(x2, y2)}
  hgp = gp.HGP(dsn, mean.constant, kernel.matern52, defs.GPParams(model=samples[0], samples=samples), wf)
  assert hgp.get_model_params_samples() is samples
  preds = hgp.predict(xq, sub_dataset_key=0)
  assert len(preds) == 3
  refs = []
  for smp, (mu_n, var_n) in zip(samples, preds):
    po = o.GPParams(model=smp)
    mu, var = o.predict(o.constant, o.matern52, po, x, y, xq, WFO)
    mu, var = o.gp_predict_postprocess(po, dso, mu, var, WFO, False, True, True)
    assert helpers.rel_err(mu_n, mu) < 1e-9 and helpers.rel_err(var_n, var) < 1e-8
    refs.append((mu, var))
  target = float(np.max(y))
  for fn, sub, prm in ((acfun.expected_improvement, o.expected_improvement_sub, target),
                       (acfun.probability_of_improvement, o.probability_of_improvement_sub, target + 0.1),
                       (acfun.ucb, o.ucb_sub, 3.0)):
    got = fn(model=hgp, sub_dataset_key=0, x_queries=xq)
    want = np.mean([sub(mu, np.sqrt(var), prm) for mu, var in refs], axis=0)      # acfun.py:72-82
    assert got.shape == (37, 1) and helpers.rel_err(got, want) < 1e-7
  # no samples: HGP degenerates to the single model (gp.py:627-631)
  single = gp.HGP(dsn, mean.constant, kernel.matern52, defs.GPParams(model=samples[1]), wf)
  plain = gp.GP(dsn, mean.constant, kernel.matern52, defs.GPParams(model=samples[1]), wf)
  np.testing.assert_allclose(acfun.ucb(model=single, sub_dataset_key=0, x_queries=xq), acfun.ucb(model=plain, sub_dataset_key=0, x_queries=xq), rtol=1e-12)
  st = hgp.stats(verbose=False)
  assert np.isfinite(st[0]) and set(st[4]) == {0, 1}


@pytest.mark.parametrize('kname,mlp,mname,n', [('squared_exponential', False, 'constant', 300), ('matern52', True, 'linear_mlp', 260),
                                                 ('matern32', False, 'linear', 129), ('dot_product', True, 'zero', 70)])
def test_hgp_samples_as_one_batch_match_the_loop_over_samples(gpu_ctx, kname, mlp, mname, n):
  """hbo_acq_samples (S parameter samples factorised as one batch, one ModelDev and one set of MLP weights per task) against
  (a) the same acquisition evaluated sample by sample on plain GPs and (b) the oracle's mean over samples (acfun.py:72-82)."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(31)
  d = 3
  S = 5
  samples = [helpers.make_model(np.random.default_rng(200 + i), mname, mlp, d) for i in range(S)]
  x, y = helpers.synthetic_task(rng, n, d)
  x2, y2 = helpers.synthetic_task(rng, 40, d)
  xq = rng.uniform(size=(53, d))
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  cov_n = getattr(kernel, kname + ('_mlp' if mlp else '')); cov_o = getattr(o, kname + ('_mlp' if mlp else ''))
  dsn = {0: defs.SubDataset(x, y), 1: defs.SubDataset(x2, y2)}
  dso = {0: o.SubDataset(x, y), 1: o.SubDataset(x2, y2)}
  hgp = gp.HGP(dsn, getattr(mean, mname), cov_n, defs.GPParams(model=samples[0], samples=samples, config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  target = float(np.max(y))
  for fn, acq_id, sub, prm in ((acfun.expected_improvement, 0, o.expected_improvement_sub, target), (acfun.ucb, 2, o.ucb_sub, 3.0)):
    per_sample = acfun.hgp_sample_values(hgp, 0, xq, acq_id, prm)
    assert per_sample.shape == (S, 53, 1)
    loop, refs = [], []
    for smp in samples:
      plain = gp.GP(dsn, getattr(mean, mname), cov_n, defs.GPParams(model=smp, config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
      loop.append(fn(model=plain, sub_dataset_key=0, x_queries=xq))
      po = o.GPParams(model=smp, config=dict(cfg))
      mu, var = o.predict(getattr(o, mname), cov_o, po, x, y, xq, WFO)
      mu, var = o.gp_predict_postprocess(po, dso, mu, var, WFO, False, True, True)
      refs.append(sub(mu, np.sqrt(var), prm))
    np.testing.assert_allclose(per_sample, np.asarray(loop), rtol=1e-9, atol=1e-11)
    got = fn(model=hgp, sub_dataset_key=0, x_queries=xq)
    assert got.shape == (53, 1) and helpers.rel_err(got, np.mean(refs, axis=0)) < 1e-7


@pytest.mark.parametrize('acq', ['ei', 'pi', 'ucb'])
@pytest.mark.parametrize('kname,mlp,mname,n_obs', [('squared_exponential', False, 'constant', 150), ('matern52', True, 'linear_mlp', 131),
                                                     ('matern32', False, 'linear', 0)])
def test_hgp_acquisition_value_and_grad_is_the_mean_over_samples(gpu_ctx, acq, kname, mlp, mname, n_obs):
  """d acquisition / d x on an HGP (acfun.py:72-82 under bayesopt.py:116-125: jax differentiates the MEAN over the parameter
  samples): value and gradient against the oracle's mean over samples; the per-sample factorisations are cached across calls
  (an L-BFGS-B inner loop calls this dozens of times), the samples' fingerprint sees in-place edits, and -- like the reference's
  loop over samples -- the call leaves the LAST sample in params.model."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(41)
  d, S = 3, 4
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  samples = [helpers.make_model(np.random.default_rng(300 + i), mname, mlp, d) for i in range(S)]
  x, y = helpers.synthetic_task(rng, max(n_obs, 1), d)
  x2, y2 = helpers.synthetic_task(rng, 20, d)
  key = 'test'
  ds = {'other': defs.SubDataset(x2, y2), 'third': defs.SubDataset(x2[:5], y2[:5])}
  if n_obs:
    ds[key] = defs.SubDataset(x, y)
  xq = rng.uniform(size=(9, d))
  cov_n = getattr(kernel, kname + ('_mlp' if mlp else '')); cov_o = getattr(o, kname + ('_mlp' if mlp else ''))
  hgp = gp.HGP(ds, getattr(mean, mname), cov_n, defs.GPParams(model=samples[0], samples=samples, config=dict(cfg)), utils.DEFAULT_WARP_FUNC)
  fn = {'ei': acfun.expected_improvement, 'pi': acfun.probability_of_improvement, 'ucb': acfun.ucb}[acq]
  param = {'ei': float(np.max(y)) if n_obs else 0.0, 'pi': float(np.max(y)) + 0.1 if n_obs else 0.0, 'ucb': 3.0}[acq]
  n_iid = len(ds)

  def oracle_mean(smps):
    vs, gs = [], []
    for smp in smps:
      po = o.GPParams(model=smp, config=dict(cfg))
      noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
      v, g = o.acquisition_value_and_grad(acq, getattr(o, mname), cov_o, po, x if n_obs else None, y if n_obs else None, xq, param, WFO,
                                          add_noise=noise, scale=n_iid / (n_iid - 1.0))
      vs.append(v); gs.append(g)
    return np.mean(vs, axis=0), np.mean(gs, axis=0)

  vo, go = oracle_mean(samples)
  val, grad = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
  assert val.shape == (9, 1) and grad.shape == (9, d)
  np.testing.assert_allclose(val, vo, rtol=1e-8, atol=1e-10)
  assert np.max(np.abs(grad - go)) <= 1e-7 * max(np.max(np.abs(go)), 1e-3)
  assert hgp.params.model is samples[-1] and hgp.params.cache == {}          # gp.py:674-678
  np.testing.assert_allclose(val, fn(model=hgp, sub_dataset_key=key, x_queries=xq), rtol=1e-8, atol=1e-10)
  if n_obs:
    handles = hgp._hbo_sample_caches[1]
    val2, grad2 = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
    assert hgp._hbo_sample_caches[1] is handles and np.array_equal(val2, val) and np.array_equal(grad2, grad)   # cached factors
    # an in-place edit that keeps every leaf's sum (swap two lengthscales) must be seen
    ls = samples[1]['lengthscale']
    if ls.size >= 2 and ls[0] != ls[1]:
      ls[[0, 1]] = ls[[1, 0]]
      vo3, go3 = oracle_mean(samples)
      val3, grad3 = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
      assert hgp._hbo_sample_caches[1] is not handles
      np.testing.assert_allclose(val3, vo3, rtol=1e-8, atol=1e-10)
      assert np.max(np.abs(grad3 - go3)) <= 1e-7 * max(np.max(np.abs(go3)), 1e-3)
      np.testing.assert_allclose(fn(model=hgp, sub_dataset_key=key, x_queries=xq), vo3, rtol=1e-8, atol=1e-10)   # the batched values too
    # the cache is bounded by the device budget (round-5 advisor finding): with room for ONE sample the other factors are built,
    # used and released per call -- same numbers; and the kept handles are closed when the observations change
    vref, gref = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
    import hyperbo_amd.bo_utils.acfun as acfun_mod
    real = acfun_mod._samples_per_call
    try:
      acfun_mod._samples_per_call = lambda n, dtype, ns: 1
      acfun_mod.drop_sample_caches(hgp)
      vb, gb = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
      assert len(hgp._hbo_sample_caches[1]) == 1 and np.array_equal(vb, vref) and np.array_equal(gb, gref)
    finally:
      acfun_mod._samples_per_call = real
    kept = hgp._hbo_sample_caches[1]
    # an in-place edit of the observations (same array objects) must not be served the stale factors
    hgp.dataset[key].y[0, 0] += 0.25
    vy, gy = fn.value_and_grad(model=hgp, sub_dataset_key=key, x_queries=xq)
    assert hgp._hbo_sample_caches[1] is not kept and not np.array_equal(vy, vref)
    hgp.dataset[key].y[0, 0] -= 0.25
    hgp.update_sub_dataset(hgp.dataset[key], key)
    assert hgp._hbo_sample_caches is None


def test_hgp_samples_go_in_chunks_that_fit_the_device(gpu_ctx, monkeypatch):
  """acfun.hgp_sample_values: the S caches of one hbo_acq_samples call are bounded by the device's memory -- more samples go in
  several calls with the same result; here the bound is forced down to 3 samples per call, and the entry point's own limit
  (S <= 4096) can no longer be hit from Python."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(43)
  d, S = 2, 11
  samples = [helpers.make_model(np.random.default_rng(400 + i), 'constant', False, d) for i in range(S)]
  x, y = helpers.synthetic_task(rng, 140, d)
  xq = rng.uniform(size=(17, d))
  hgp = gp.HGP({0: defs.SubDataset(x, y)}, mean.constant, kernel.squared_exponential, defs.GPParams(model=samples[0], samples=samples),
               utils.DEFAULT_WARP_FUNC)
  whole = acfun.hgp_sample_values(hgp, 0, xq, 0, float(np.max(y)))
  assert acfun._samples_per_call(140, np.float64, S) == S
  assert acfun._samples_per_call(60000, np.float64, 4096) < 10            # 3 x 29 GB per sample
  monkeypatch.setattr(acfun, '_samples_per_call', lambda n, dtype, s: 3)
  parts = acfun.hgp_sample_values(hgp, 0, xq, 0, float(np.max(y)))
  assert np.array_equal(parts, whole)


def test_hgp_samples_batch_fp32_and_a_sample_that_is_not_positive_definite(gpu_ctx):
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(32)
  d, S = 2, 4
  samples = [helpers.make_model(np.random.default_rng(300 + i), 'constant', False, d, dtype=np.float32) for i in range(S)]
  x, y = helpers.synthetic_task(rng, 150, d)
  x32, y32 = x.astype(np.float32), y.astype(np.float32)
  xq = rng.uniform(size=(20, d)).astype(np.float32)
  hgp = gp.HGP({0: defs.SubDataset(x32, y32)}, mean.constant, kernel.squared_exponential, defs.GPParams(model=samples[0], samples=samples), utils.DEFAULT_WARP_FUNC)
  vals = acfun.hgp_sample_values(hgp, 0, xq, 2, 3.0)
  assert vals.dtype == np.float32 and np.all(np.isfinite(vals))
  for s_, smp in enumerate(samples):
    po = o.GPParams(model={k_: np.asarray(v_, dtype=np.float64) for k_, v_ in smp.items()})
    mu, var = o.predict(o.constant, o.squared_exponential, po, x32.astype(np.float64), y32.astype(np.float64), xq.astype(np.float64), WFO)
    mu, var = o.gp_predict_postprocess(po, {0: o.SubDataset(x, y)}, mu, var, WFO, False, True, True)
    assert helpers.rel_err(vals[s_], o.ucb_sub(mu, np.sqrt(var), 3.0)) < 5e-3
  # more candidates than one posterior chunk: the passes run one after the other, streamed in chunks, same values
  xq_many = rng.uniform(size=(300, d)).astype(np.float32)
  ref_many = acfun.hgp_sample_values(hgp, 0, xq_many, 2, 3.0)
  try:
    gpu_ctx.set_option('post_chunk', 128)
    np.testing.assert_array_equal(acfun.hgp_sample_values(hgp, 0, xq_many, 2, 3.0), ref_many)
  finally:
    gpu_ctx.set_option('post_chunk', 8192)
  # ONE of the samples has a Gram matrix that is not positive definite (un-warped negative noise): its row is NaN, like the
  # reference's NaN Cholesky, and the other samples of the batch are untouched
  bad = [{'dot_prod_sigma': np.array(1.0), 'dot_prod_bias': np.array(0.1), 'noise_variance': np.array(nv)} for nv in (0.1, -1.0, 0.2)]
  h2 = gp.HGP({0: defs.SubDataset(x, y)}, mean.zero, kernel.dot_product, defs.GPParams(model=bad[0], samples=bad), None)
  v2 = acfun.hgp_sample_values(h2, 0, xq.astype(np.float64), 2, 3.0)
  assert np.all(np.isfinite(v2[0])) and np.all(np.isfinite(v2[2]))
  assert np.all(np.isnan(v2[1]))
  single = gp.GP({0: defs.SubDataset(x, y)}, mean.zero, kernel.dot_product, defs.GPParams(model=bad[2]), None)
  np.testing.assert_allclose(v2[2], acfun.ucb(model=single, sub_dataset_key=0, x_queries=xq.astype(np.float64)), rtol=1e-9)


def test_empty_task_shard_builds_the_same_model_and_contributes_zeros(gpu_ctx):
  """A rank beyond the task count holds an empty shard (parallel.shard_dataset): it must build the peers' model (ARD
  lengthscale / linear mean / MLP with D > 1) and reach the all-reduce with zeros of the right layout."""
  from hyperbo_amd import parallel
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(23)
  d = 4
  full = {0: defs.SubDataset(*helpers.synthetic_task(rng, 50, d))}
  assert parallel.shard_dataset(full, 1, 2) == {}

  class Recorder(parallel.LocalComm):
    def allreduce_sum(self, buf):
      self.buf = np.array(buf)
      return buf
  for mname, kname, mlp in (('constant', 'squared_exponential', False), ('linear', 'matern32', False), ('linear_mlp', 'matern52', True)):
    model = helpers.make_model(rng, mname, mlp, d)
    pn = defs.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES})
    kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
    comm = Recorder()
    v, g = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, {}, utils.DEFAULT_WARP_FUNC, comm=comm)
    _, gfull = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, full, utils.DEFAULT_WARP_FUNC, comm=Recorder())
    assert v == 0.0 and np.all(comm.buf == 0) and helpers.flatten(g).size == helpers.flatten(gfull).size


@pytest.mark.parametrize('n,m', [(60, 1), (64, 2), (65, 3), (200, 1), (1000, 4)])
def test_cached_cholesky_path_of_inverse_spdmatrix_vector_product(gpu_ctx, n, m):
  """linalg.py:139-145 with a factor handed in as an array: two substitution sweeps on the device (hbo_chol_solve), checked
  against the factorising path, the defining equation and host LAPACK across the 64-row block edges and several right-hand sides."""
  import scipy.linalg as spla
  _, linalg, *_ = _native()
  rng = np.random.default_rng(24 + n)
  a = rng.normal(size=(n, n)); a = a @ a.T + n * np.eye(n)
  v = rng.normal(size=(n, m))
  chol = np.linalg.cholesky(a)
  x1 = linalg.inverse_spdmatrix_vector_product(a, v, cached_cholesky=chol)
  assert x1.shape == v.shape and x1.dtype == np.float64
  np.testing.assert_allclose(x1, spla.cho_solve((chol, True), v), rtol=1e-10, atol=1e-13)
  np.testing.assert_allclose(a @ x1, v, rtol=1e-9, atol=1e-10)
  if m == 1:
    _, x0 = linalg.solve_linear_system(a, v)
    np.testing.assert_allclose(x1, x0, rtol=1e-9, atol=1e-12)
  # the upper triangle of the array is never read; fp32 in -> fp32 out
  junk = chol + np.triu(np.full((n, n), 7.0), 1)
  np.testing.assert_array_equal(linalg.inverse_spdmatrix_vector_product(a, v, cached_cholesky=junk), x1)
  x32 = linalg.inverse_spdmatrix_vector_product(a.astype(np.float32), v.astype(np.float32), cached_cholesky=chol.astype(np.float32))
  assert x32.dtype == np.float32
  np.testing.assert_allclose(x32, x1, rtol=2e-3, atol=2e-5)


# ---- divergence objectives (objectives.py:29-106): EKL / Euclid on the device vs the oracle ------------------
def _aligned_datasets(rng, d):
  dso = {'a': o.SubDataset(*helpers.synthetic_task(rng, 150, d, m=6), aligned=1),
         'b': o.SubDataset(*helpers.synthetic_task(rng, 20, d, m=10), aligned='x'),
         'iid': o.SubDataset(*helpers.synthetic_task(rng, 40, d)),
         'one': o.SubDataset(*helpers.synthetic_task(rng, 130, d, m=1), aligned=2),
         'empty': o.SubDataset(np.zeros((0, d)), np.zeros((0, 2)), aligned=3)}
  return dso


@pytest.mark.parametrize('kind', ['ekl', 'euc'])
@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mlp', [False, True])
@pytest.mark.parametrize('mname', helpers.MEANS)
def test_divergence_value_and_grad_vs_oracle_fp64(gpu_ctx, kind, kname, mlp, mname):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(13)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  dso = _aligned_datasets(rng, d)
  dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  vo, go = o.divergence_value_and_grad(kind, getattr(o, mname), ko, po, dso, WFO)
  fn = objectives.ekl if kind == 'ekl' else objectives.euc
  vn, gn = fn.value_and_grad(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert set(gn) == set(go)
  # the partial KL adds tr(K1^-1 C0) + logdet of opposite signs: scale the tolerance by the terms' size
  assert abs(vn - vo) <= 1e-10 * max(abs(vo), 1.0) * (50 if kind == 'ekl' else 1)
  fo, fng = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL, label=kind)
  v_only = fn(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(v_only - vn) <= 1e-12 * max(abs(vn), 1.0)
  if kind == 'euc':
    v_dist = objectives.multivariate_normal_divergence(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC,
                                                       distance=utils.euclidean_multivariate_normal)
    assert v_dist == v_only


@pytest.mark.parametrize('kind', ['ekl', 'euc'])
@pytest.mark.parametrize('kname,mlp,mname,dtype', [('squared_exponential', False, 'constant', np.float64), ('matern52', True, 'linear_mlp', np.float64),
                                                    ('matern32', False, 'linear', np.float64), ('squared_exponential', False, 'zero', np.float32)])
def test_divergence_with_more_than_127_aligned_columns(gpu_ctx, kind, kname, mlp, mname, dtype):
  """objectives.py:29-106 puts no limit on the number m of aligned columns; the augmented tile-row holds 128 rows.  Beyond that the
  data rows become outer-product vectors of their own (EKL: through the explicit inverse, EUC: as they are): value and gradient
  against the oracle for m = 128 (the first size that does not fit), m = 300 with n = 200 (two blocks), next to a sub-dataset
  that still takes the tile path, value-only calls, and the same batch sub-sampled on the device."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(57)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d, dtype=dtype)
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  po = o.GPParams(model=_to64(model), config=dict(cfg)); pn = defs.GPParams(model=model, config=dict(cfg))
  dso = {}
  for key, (n, m) in {'tile': (90, 40), 'first': (70, 128), 'big': (200, 300)}.items():
    x = rng.uniform(size=(n, d))
    y = np.sin(3 * x.sum(axis=1, keepdims=True) + rng.normal(size=(1, m))) + 0.3 * rng.normal(size=(n, m))
    dso[key] = o.SubDataset(x, y, aligned=key)
  dsn = {k: defs.SubDataset(v.x.astype(dtype), v.y.astype(dtype), v.aligned) for k, v in dso.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  vo, go = o.divergence_value_and_grad(kind, getattr(o, mname), ko, po, dso, WFO)
  fn = objectives.ekl if kind == 'ekl' else objectives.euc
  vn, gn = fn.value_and_grad(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
  vtol, gtol = (1e-10 * (50 if kind == 'ekl' else 1), FP64_GRAD_TOL) if dtype == np.float64 else (2e-3, 2e-3)
  assert abs(vn - vo) <= vtol * max(abs(vo), 1.0), (vn, vo)
  fo, fng = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, gtol, label=kind)
  v_only = fn(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(v_only - vn) <= (1e-12 if dtype == np.float64 else 1e-5) * max(abs(vn), 1.0)
  if dtype == np.float64:
    # each sub-dataset alone (the value-only path of a batch that holds ONLY big-m tasks runs the inverse for the extra rows)
    for key in ('first', 'big'):
      v1 = fn(getattr(mean, mname), kn, pn, {key: dsn[key]}, utils.DEFAULT_WARP_FUNC)
      v1o, _ = o.divergence_value_and_grad(kind, getattr(o, mname), ko, po, {key: dso[key]}, WFO)
      assert abs(v1 - v1o) <= vtol * max(abs(v1o), 1.0), (key, v1, v1o)
    # rows gathered on the device (hbo_dataset_subsample carries the divergence rows of every column along)
    dev = objectives.DeviceBatch(dsn)
    idx = {'tile': None, 'first': np.arange(0, 70, 2, dtype=np.int32), 'big': rng.permutation(200)[:150].astype(np.int32)}
    sub = dev.subsample(idx)
    vs, gs = fn.value_and_grad(getattr(mean, mname), kn, pn, sub, utils.DEFAULT_WARP_FUNC)
    dso_s = {k: (v if idx[k] is None else o.SubDataset(v.x[idx[k]], v.y[idx[k]], v.aligned)) for k, v in dso.items()}
    vso, gso = o.divergence_value_and_grad(kind, getattr(o, mname), ko, po, dso_s, WFO)
    assert abs(vs - vso) <= vtol * max(abs(vso), 1.0)
    helpers.assert_grad_close(gs, gso, gtol, label=kind + ' subsample')
    sub.close(); dev.close()


def test_value_only_ekl_small_m_task_of_two_blocks_beside_a_big_m_task(gpu_ctx):
  """A value-only EKL call on a FRESH dataset whose batch holds a task with more than 127 aligned columns runs the inverse over all
  tasks; a task of the same batch with few columns but two or more blocks (n >= 129) receives the S21 products of GEMM_TRTRI_A
  too and needs the S buffer (round-5 advisor finding: it was only allocated for the big-m task).  Values against the oracle, then
  the gradient call on the same dataset."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(61)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = _pair(model)
  dso = {}
  for key, (n, m) in {'small_m_two_blocks': (150, 5), 'small_m_three_blocks': (300, 2), 'big': (140, 130)}.items():
    x = rng.uniform(size=(n, d))
    y = np.sin(3 * x.sum(axis=1, keepdims=True) + rng.normal(size=(1, m))) + 0.3 * rng.normal(size=(n, m))
    dso[key] = o.SubDataset(x, y, aligned=key)
  dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
  vo, go = o.divergence_value_and_grad('ekl', o.constant, o.matern52, po, dso, WFO)
  v_only = objectives.ekl(mean.constant, kernel.matern52, pn, dsn, utils.DEFAULT_WARP_FUNC)   # first call on this dataset: value only
  assert abs(v_only - vo) <= 5e-9 * max(abs(vo), 1.0), (v_only, vo)
  dev = objectives.DeviceDataset(dsn, only_aligned=True)
  v1 = objectives.ekl(mean.constant, kernel.matern52, pn, dev, utils.DEFAULT_WARP_FUNC)
  vn, gn = objectives.ekl.value_and_grad(mean.constant, kernel.matern52, pn, dev, utils.DEFAULT_WARP_FUNC)
  dev.close()
  assert abs(v1 - vo) <= 5e-9 * max(abs(vo), 1.0) and abs(vn - vo) <= 5e-9 * max(abs(vo), 1.0)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)


def _to64(tree):
  return {k: _to64(v) for k, v in tree.items()} if isinstance(tree, dict) else np.asarray(tree, dtype=np.float64)


@pytest.mark.parametrize('qs', [0, 1, 2, 8])
@pytest.mark.parametrize('kname,mlp,mname', [('squared_exponential', False, 'constant'), ('matern52', True, 'linear_mlp'), ('dot_product', False, 'linear'),
                                             ('matern32', True, 'zero')])
def test_one_sweep_inverse_across_the_registry_and_objectives(gpu_ctx, kname, mlp, mname, qs):
  """The one-sweep inverse (sched.hip:sweep_advance -- W = L^-1 and K^-1 = W^T W row group by row group behind the panel chain;
  default only for batches large enough for look-ahead) forced on for small ragged batches and a single matrix: NLL and EKL value
  + every gradient leaf (incl. the MLP backward, which reads K^-1 tile by tile), fp64 against the oracle and fp32 against fp64,
  for every row-group size -- groups cut by the end of a task, tasks shorter than one group, an empty and a one-point task."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(91)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  sizes = (700, 130, 1, 515, 0, 1290, 1153)
  dso = {k: o.SubDataset(*helpers.synthetic_task(rng, n, d)) if n else o.SubDataset(np.zeros((0, d)), np.zeros((0, 1))) for k, n in enumerate(sizes)}
  dsn = {k: defs.SubDataset(v.x, v.y) for k, v in dso.items()}
  al_o = {k: o.SubDataset(*helpers.synthetic_task(rng, n, d, m=4), aligned=k) for k, n in enumerate((900, 260, 1100))}
  al_n = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in al_o.items()}
  try:
    for k_, v_ in {'lookahead': 2, 'sweep': 2, 'sweep_qs': qs}.items():
      gpu_ctx.set_option(k_, v_)
    vo, go = o.nll_value_and_grad(getattr(o, mname), ko, po, dso, WFO)
    vn, gn = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, dsn, utils.DEFAULT_WARP_FUNC)
    assert abs(vn - vo) <= 1e-10 * abs(vo)
    fo, fn = helpers.flatten(go), helpers.flatten(gn)
    helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
    single = {0: dso[5]}
    vo1, go1 = o.nll_value_and_grad(getattr(o, mname), ko, po, single, WFO)
    vn1, gn1 = objectives.nll_value_and_grad(getattr(mean, mname), kn, pn, {0: dsn[5]}, utils.DEFAULT_WARP_FUNC)
    assert abs(vn1 - vo1) <= 1e-10 * abs(vo1)
    helpers.assert_grad_close(gn1, go1, FP64_GRAD_TOL)
    ve, ge = o.divergence_value_and_grad('ekl', getattr(o, mname), ko, po, al_o, WFO)
    vne, gne = objectives.ekl.value_and_grad(getattr(mean, mname), kn, pn, al_n, utils.DEFAULT_WARP_FUNC)
    assert abs(vne - ve) <= 5e-9 * max(abs(ve), 1.0)
    helpers.assert_grad_close(gne, ge, FP64_GRAD_TOL)
    if qs in (0, 2):
      to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
      p32 = defs.GPParams(model=to32(model), config={'mlp_features': helpers.MLP_FEATURES})
      ds32 = {k: defs.SubDataset(v.x.astype(np.float32), v.y.astype(np.float32)) for k, v in dsn.items()}
      v32, g32 = objectives.nll_value_and_grad(getattr(mean, mname), kn, p32, ds32, utils.DEFAULT_WARP_FUNC)
      assert abs(v32 - vo) <= 2e-4 * abs(vo)
      helpers.assert_grad_close(g32, go, FP32_GRAD_TOL)
  finally:
    for k_, v_ in {'lookahead': 1, 'sweep': 1, 'sweep_qs': 0}.items():
      gpu_ctx.set_option(k_, v_)


@pytest.mark.parametrize('n', [2200, 3750, 5300])
def test_one_matrix_in_the_size_range_where_the_sweep_is_the_default(gpu_ctx, n):
  """One fp64 matrix of 17-48 blocks takes the one-sweep inverse by default (sched.hip:use_sweep; row groups of 4 blocks up to 28
  blocks, of 8 above; above 40 blocks on 128-tiles with the K^-1 updates on a stream of their own): against the oracle, and against
  the block-recursive inverse + W^T W (sweep = 0) to rounding."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(n)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = _pair(model)
  x, y = helpers.synthetic_task(rng, n, d)
  vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, po, {0: o.SubDataset(x, y)}, WFO)
  dsn = {0: defs.SubDataset(x, y)}
  vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert abs(vn - vo) <= 1e-10 * abs(vo)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  helpers.assert_grad_close(gn, go, FP64_GRAD_TOL)
  try:
    gpu_ctx.set_option('sweep', 0)
    v0, g0 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC)
  finally:
    gpu_ctx.set_option('sweep', 1)
  assert v0 == vn                                    # the factorisation is the same
  helpers.assert_grad_close(g0, gn, 1e-12)


def test_divergence_fp32_and_no_aligned_data(gpu_ctx):
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(14)
  d = 2
  model = helpers.make_model(rng, 'constant', False, d)
  po, _ = _pair(model)
  model32 = {k: np.asarray(v, dtype=np.float32) for k, v in model.items()}
  pn = defs.GPParams(model=model32, config={'mlp_features': helpers.MLP_FEATURES})
  dso = {'a': o.SubDataset(*helpers.synthetic_task(rng, 200, d, m=8), aligned=1)}
  dsn = {k: defs.SubDataset(v.x.astype(np.float32), v.y.astype(np.float32), v.aligned) for k, v in dso.items()}
  for kind, fn in (('ekl', objectives.ekl), ('euc', objectives.euc)):
    vo, go = o.divergence_value_and_grad(kind, o.constant, o.matern52, po, dso, WFO)
    vn, gn = fn.value_and_grad(mean.constant, kernel.matern52, pn, dsn, utils.DEFAULT_WARP_FUNC)
    assert abs(vn - vo) <= 2e-3 * max(abs(vo), 1.0)
    fo, fng = helpers.flatten(go), helpers.flatten(gn)
    helpers.assert_grad_close(gn, go, 2e-3, label=kind)
  only_iid = {'iid': defs.SubDataset(*helpers.synthetic_task(rng, 10, d))}
  p64 = defs.GPParams(model=model, config={})
  assert objectives.ekl(mean.constant, kernel.matern52, p64, only_iid, utils.DEFAULT_WARP_FUNC) == 0.
  v, g = objectives.euc.value_and_grad(mean.constant, kernel.matern52, p64, only_iid, utils.DEFAULT_WARP_FUNC)
  assert v == 0. and all(np.all(np.asarray(x) == 0) for x in g.values())
  bad = {'bad': defs.SubDataset(np.zeros((4, d)), np.zeros((3, 2)), aligned=1)}
  with pytest.raises(ValueError):
    objectives.ekl(mean.constant, kernel.matern52, p64, bad, utils.DEFAULT_WARP_FUNC)


def test_kl_of_model_samples_is_small_and_array_level_distances(gpu_ctx):
  # objectives_test.py:60-100 flavour: data drawn from the model itself; utils_test.py:26-55 inputs
  defs, linalg, _, _, kernel, mean, objectives, utils = _native()
  np.random.seed(1)
  mu0 = np.random.uniform(-5, 5, (10,)); mu1 = np.random.uniform(-5, 5, (10,))
  cov0 = np.random.uniform(-5, 5, (10, 100)); cov0 = cov0 @ cov0.T
  cov1 = np.random.uniform(-5, 5, (10, 100)); cov1 = cov1 @ cov1.T
  kl_01 = utils.kl_multivariate_normal(mu0, cov0, mu1, cov1, partial=False)
  assert kl_01 > 0 and abs(kl_01 - o.kl_multivariate_normal(mu0, cov0, mu1, cov1, partial=False)) <= 1e-9 * kl_01
  assert abs(utils.kl_multivariate_normal(mu0, cov0, mu0, cov0, partial=False)) <= 1e-5
  pk, pko = utils.partial_kl_mvn(mu0, cov0, mu1, cov1), o.partial_kl_mvn(mu0, cov0, mu1, cov1)
  assert abs(pk - pko) <= 1e-10 * abs(pko)
  assert abs(utils.kl_multivariate_normal(mu0, cov0, mu1, cov1, weight=2., eps=1e-3)
             - o.kl_multivariate_normal(mu0, cov0, mu1, cov1, weight=2., eps=1e-3)) <= 1e-10 * abs(pko)
  np.random.seed(1)
  mu0 = np.random.uniform(-5, 5, (100,)); mu1 = np.random.uniform(-5, 5, (100,))
  feat0 = np.random.uniform(-5, 5, (100, 5)); cov0 = feat0 @ feat0.T
  cov1 = np.random.uniform(-5, 5, (100, 1000)); cov1 = cov1 @ cov1.T
  kl = utils.kl_multivariate_normal(mu0, cov0, mu1, cov1, partial=False)
  assert 0 < kl < np.inf
  assert abs(utils.euclidean_multivariate_normal(mu0, cov0, mu1, cov1, mean_weight=2., cov_weight=.5)
             - o.euclidean_multivariate_normal(mu0, cov0, mu1, cov1, mean_weight=2., cov_weight=.5)) <= 1e-9
  v = np.random.randn(100)
  np.testing.assert_allclose(cov1 @ linalg.inverse_spdmatrix_vector_product(cov1, v), v, rtol=1e-8, atol=1e-8)
  # data whose sample mean / covariance ARE the model's: Euclid distance 0, partial KL = n + logdet K1 and the
  # model is a stationary point of the KL (zero gradient)
  rng = np.random.default_rng(0)
  n, d, m = 40, 2, 120
  x = rng.uniform(size=(n, d))
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = _pair(model)
  k1 = o.squared_exponential(po, x, warp_func=WFO) + np.eye(n) * o.retrieve_params(po, ['noise_variance'], WFO)[0]
  mu = o.constant(po, x, warp_func=WFO)
  basis = np.linalg.qr(np.concatenate([np.ones((m, 1)), rng.normal(size=(m, n))], axis=1))[0][:, 1:]   # m x n, cols _|_ 1
  y = mu + np.sqrt(m) * np.linalg.cholesky(k1) @ basis.T
  dsn = {'s': defs.SubDataset(x, y, aligned=1)}
  args = (mean.constant, kernel.squared_exponential, pn, dsn, utils.DEFAULT_WARP_FUNC)
  assert objectives.euc(*args) <= 1e-10
  v, g = objectives.ekl.value_and_grad(*args)
  assert abs(v - (n + np.linalg.slogdet(k1)[1])) <= 1e-9 * n
  assert np.max(np.abs(helpers.flatten(g))) <= 1e-8 * n


@pytest.mark.parametrize('objective_name,method,cov_name', [
    ('euc', 'lbfgs', 'squared_exponential'), ('kl', 'adam', 'dot_product_mlp'), ('kl', 'adam', 'squared_exponential_mlp'),
    ('nll_regkl1', 'adam', 'matern32'), ('nll_regeuc1', 'lbfgs', 'matern52')])
def test_infer_parameters_on_divergence_objectives(gpu_ctx, objective_name, method, cov_name):
  # objectives_test.py:60-130: constant mean, n=20, d=2, 10 aligned samples, a couple of training steps
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(2)
  n, d, m = 20, 2, 10
  mlp = cov_name.endswith('_mlp')
  model = helpers.make_model(rng, 'constant', mlp, d)
  x = rng.uniform(size=(n, d))
  ds = {'all_data': defs.SubDataset(x, np.sin(3 * x[:, :1]) + 0.3 * rng.normal(size=(n, m)), aligned='all_data'),
        'iid0': defs.SubDataset(*helpers.synthetic_task(rng, 30, d)), 'iid1': defs.SubDataset(*helpers.synthetic_task(rng, 25, d))}
  objective = getattr(objectives, objective_name)
  cfg = {'method': method, 'batch_size': 100, 'max_training_step': 5, 'learning_rate': 1e-3,
         'mlp_features': helpers.MLP_FEATURES, 'objective': objective}
  params = defs.GPParams(model=model, config=cfg)
  cov, mu = getattr(kernel, cov_name), mean.constant
  init = objective(mean_func=mu, cov_func=cov, params=params, dataset=ds, warp_func=utils.DEFAULT_WARP_FUNC)
  # the composed value equals the sum of its native parts (oracle for the pieces)
  dso = {k: o.SubDataset(v.x, v.y, v.aligned) for k, v in ds.items()}
  po = o.GPParams(model=model, config=cfg)
  ko = getattr(o, cov_name)
  parts = {'euc': lambda: o.divergence_value_and_grad('euc', o.constant, ko, po, dso, WFO)[0],
           'kl': lambda: o.divergence_value_and_grad('ekl', o.constant, ko, po, dso, WFO)[0],
           'nll': lambda: o.neg_log_marginal_likelihood(o.constant, ko, po, dso, WFO)}
  expect = {'euc': lambda: parts['euc'](), 'kl': lambda: parts['kl'](),
            'nll_regkl1': lambda: parts['nll']() + parts['kl'](),
            'nll_regeuc1': lambda: parts['nll']() + parts['euc']()}[objective_name]()
  assert abs(init - expect) <= 1e-9 * max(abs(expect), 1.0)
  out = gp.infer_parameters(mu, cov, params, ds, warp_func=utils.DEFAULT_WARP_FUNC, objective=objective, key=0)
  final = objective(mean_func=mu, cov_func=cov, params=out, dataset=ds, warp_func=utils.DEFAULT_WARP_FUNC)
  assert np.isfinite(final) and final < init


# ---- training driver on the native objective (gp_test.py:58-148, objectives_test.py:206-324) -------
@pytest.mark.parametrize('method,steps,lr', [('adam', 10, 1e-2), ('lbfgs', 3, None)])
@pytest.mark.parametrize('kname,mname', [('squared_exponential', 'constant'), ('matern32', 'zero'),
                                         ('matern52_mlp', 'linear_mlp'), ('dot_product_mlp', 'linear')])
def test_gp_train_reduces_nll(gpu_ctx, method, steps, lr, kname, mname):
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(21)
  d = 1
  mlp = kname.endswith('_mlp')
  model = helpers.make_model(rng, mname, mlp, d)
  ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, 100, d)) for i in range(10)}   # 10 tasks x n=100, d=1
  ds['aligned'] = defs.SubDataset(*helpers.synthetic_task(rng, 30, d, m=5), aligned=1)  # ignored by the NLL
  cfg = {'method': method, 'batch_size': 50 if method == 'adam' else 300, 'max_training_step': steps,
         'learning_rate': lr, 'mlp_features': helpers.MLP_FEATURES}
  model_n = gp.GP(ds, getattr(mean, mname), getattr(kernel, kname), defs.GPParams(model=model, config=cfg),
                  utils.DEFAULT_WARP_FUNC)
  init_nll, _ = model_n.neg_log_marginal_likelihood()
  seen = []
  model_n.train(key=0, callback=lambda *a, **k: seen.append(1))
  nll, key2nll = model_n.neg_log_marginal_likelihood()
  assert np.isfinite(nll) and nll < init_nll
  assert 'aligned' not in key2nll and len(key2nll) == 10 and seen
  assert model_n.params.cache == {}


def _shared_batches_for_oracle(ds, batch_size, seed):
  """Batches for the oracle's training loop from the SAME row draws GP.train makes (its index iterator, same seed)."""
  from hyperbo_amd.basics import data_utils
  for index in data_utils.sub_sample_index_iterator(np.random.default_rng(seed), ds, batch_size):
    batch = {}
    for i, (k, s) in enumerate(ds.items()):
      ix = index[k]
      x, y = (s.x, s.y) if ix is None else (s.x[ix], s.y[ix])
      batch[k] = o.SubDataset(x, y, i if isinstance(s.aligned, str) else s.aligned)
    yield batch


@pytest.mark.parametrize('method,steps,lr', [('adam', 10, 2e-2), ('lbfgs', 3, None)])
@pytest.mark.parametrize('kname,mname', [('squared_exponential', 'constant'), ('matern32', 'zero'),
                                         ('matern52_mlp', 'linear_mlp'), ('dot_product_mlp', 'linear')])
def test_gp_train_trajectory_vs_oracle_driver(gpu_ctx, method, steps, lr, kname, mname):
  """Row f1: GP.train() on the device (native NLL+grad behind hyperbo_amd's flat-vector Adam / L-BFGS, batches gathered in HBM)
  against the oracle-side driver (oracle/train_oracle.py: dict-pytree restatement of lbfgs.py:51-349 and gp.py:53-195 around the
  oracle's value_and_grad) on the same drawn rows.  EVERY objective evaluation -- each Adam step, each L-BFGS step and each
  line-search probe (so: every step size) -- must happen at the same parameters (1e-8) with the same loss (1e-8)."""
  from oracle import train_oracle as to
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(33)
  d = 2
  mlp = kname.endswith('_mlp')
  model = helpers.make_model(rng, mname, mlp, d)
  ds = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i, n in enumerate((100, 64, 130, 80, 45))}
  ds['aligned'] = defs.SubDataset(*helpers.synthetic_task(rng, 30, d, m=5), aligned='tag')   # ignored by the NLL
  bs = 50 if method == 'adam' else 300
  cfg = {'method': method, 'batch_size': bs, 'max_training_step': steps, 'learning_rate': lr, 'mlp_features': helpers.MLP_FEATURES}
  evals_d = []
  inner = objectives.nll.value_and_grad
  def logged(**kw):
    return objectives.nll(**kw)
  def logged_vg(**kw):
    v, g = inner(**kw)
    evals_d.append((helpers.flatten(kw['params'].model), float(v)))
    return v, g
  logged_vg.accepts_device_batch = getattr(inner, 'accepts_device_batch', False)
  assert logged_vg.accepts_device_batch                     # the resident-dataset path is the one under test
  logged.value_and_grad = logged_vg
  cfg_d = dict(cfg, objective=logged)
  cb_d, cb_o = [], []
  model_d = gp.GP(ds, getattr(mean, mname), getattr(kernel, kname), defs.GPParams(model=to.tree_copy(model), config=cfg_d),
                  utils.DEFAULT_WARP_FUNC)
  out_d = model_d.train(key=12, callback=lambda *a, **k: cb_d.append(float(k['loss'] if 'loss' in k else a[2])))
  trace = []
  dso = {k: o.SubDataset(v.x, v.y, v.aligned) for k, v in ds.items()}
  out_o = to.infer_parameters(getattr(o, mname), getattr(o, kname), o.GPParams(model=to.tree_copy(model), config=dict(cfg)), dso,
                              WFO, dataset_iter=_shared_batches_for_oracle(ds, bs, 12), trace=trace,
                              callback=lambda *a, **k: cb_o.append(float(k['loss'] if 'loss' in k else a[2])))
  evals_o = [(helpers.flatten(r[1]), r[2]) for r in trace if r[0] == 'eval']
  assert len(evals_d) == len(evals_o) >= steps + 1, (len(evals_d), len(evals_o))
  for i, ((xd, fd), (xo, fo)) in enumerate(zip(evals_d, evals_o)):
    assert helpers.rel_err(xd, xo) < 1e-8, (i, helpers.rel_err(xd, xo))
    assert abs(fd - fo) <= 1e-8 * max(1.0, abs(fo)), (i, fd, fo)
  assert len(cb_d) == len(cb_o) and helpers.rel_err(cb_d, cb_o) < 1e-8
  assert helpers.rel_err(helpers.flatten(out_d.model), helpers.flatten(out_o.model)) < 1e-8
  assert evals_d[-1][1] < evals_d[0][1] or method == 'adam'
  if method == 'lbfgs':
    sizes = [r[2] for r in trace if r[0] == 'step']
    assert sizes and all(np.isfinite(sizes))


def test_gp_train_guards_on_the_device(gpu_ctx):
  """gp.py:135-142 on the native objective: a NaN loss at step 0 raises; a loss that turns non-finite later stops the loop and
  the last finite parameters are kept -- same result as the oracle-side driver given the same losses."""
  from oracle import train_oracle as to
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(2)
  x = rng.uniform(size=(40, 2))
  ds = {0: defs.SubDataset(x, np.full((40, 1), np.nan))}
  model = helpers.make_model(rng, 'constant', False, 2)
  cfg = {'method': 'adam', 'batch_size': 100, 'max_training_step': 3, 'learning_rate': 1e-2}
  with pytest.raises(ValueError):
    gp.infer_parameters(mean.constant, kernel.squared_exponential, defs.GPParams(model=to.tree_copy(model), config=dict(cfg)), ds,
                        utils.DEFAULT_WARP_FUNC, objectives.nll, key=0)
  with pytest.raises(ValueError):
    to.infer_parameters(o.constant, o.squared_exponential, o.GPParams(model=to.tree_copy(model), config=dict(cfg)),
                        {0: o.SubDataset(x, np.full((40, 1), np.nan))}, WFO)
  # the loss turns non-finite at the 4th evaluation (injected on top of the native objective / the oracle): both drivers stop there
  # and keep the parameters of the 3rd evaluation
  y = np.sin(4 * x[:, :1]) + 0.05 * rng.normal(size=(40, 1))
  ds = {0: defs.SubDataset(x, y)}
  cfg = {'method': 'adam', 'batch_size': 100, 'max_training_step': 12, 'learning_rate': 0.05}
  calls = {'d': 0, 'o': 0}
  def inj(**kw):
    return objectives.nll(**kw)
  def inj_vg(**kw):
    calls['d'] += 1
    v, g = objectives.nll.value_and_grad(**kw)
    return (float('inf') if calls['d'] >= 4 else v), g
  inj_vg.accepts_device_batch = True
  inj.value_and_grad = inj_vg
  def inj_o(*a, **kw):
    calls['o'] += 1
    v, g = o.nll_value_and_grad(*a, **kw)
    return (float('inf') if calls['o'] >= 4 else v), g
  losses_d, losses_o = [], []
  out_d = gp.infer_parameters(mean.constant, kernel.squared_exponential, defs.GPParams(model=to.tree_copy(model), config=dict(cfg)),
                              ds, utils.DEFAULT_WARP_FUNC, inj, key=0, callback=lambda i, m_, l_: losses_d.append(float(l_)))
  out_o = to.infer_parameters(o.constant, o.squared_exponential, o.GPParams(model=to.tree_copy(model), config=dict(cfg)),
                              {0: o.SubDataset(x, y)}, WFO, value_and_grad=inj_o, callback=lambda i, m_, l_: losses_o.append(float(l_)))
  assert len(losses_d) == len(losses_o) == 3 and helpers.rel_err(losses_d, losses_o) < 1e-9
  assert np.all(np.isfinite(helpers.flatten(out_d.model)))
  assert helpers.rel_err(helpers.flatten(out_d.model), helpers.flatten(out_o.model)) < 1e-8
  assert helpers.rel_err(helpers.flatten(out_d.model), helpers.flatten(model)) > 1e-3       # it did move before stopping


def test_simulated_bo_iteration(gpu_ctx):
  """One step of hyperbo/bo_utils/bayesopt.py:164-190: acquisition over all candidates -> argmax ->
  append (cache goes stale) -> next acquisition re-factorises; checked against the oracle."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(22)
  d = 5
  model = helpers.make_model(rng, 'constant', False, d)
  x, y = helpers.synthetic_task(rng, 40, d)
  cand_x, cand_y = helpers.synthetic_task(rng, 30, d)
  ds = {0: defs.SubDataset(x, y), 1: defs.SubDataset(x[:10], y[:10])}
  m = gp.GP(ds, mean.constant, kernel.matern52, defs.GPParams(model=model), utils.DEFAULT_WARP_FUNC)
  po = o.GPParams(model=model)
  for it in range(3):
    ev = acfun.expected_improvement(model=m, sub_dataset_key=0, x_queries=cand_x)
    sd = m.dataset[0]
    mu, var = o.predict(o.constant, o.matern52, po, sd.x, sd.y, cand_x, WFO)
    mu, var = o.gp_predict_postprocess(po, {k: o.SubDataset(v.x, v.y) for k, v in m.dataset.items()}, mu, var, WFO, False, True, True)
    ref = o.expected_improvement_sub(mu, np.sqrt(var), float(np.max(sd.y)))
    assert helpers.rel_err(ev, ref) < 1e-7 and int(np.argmax(ev)) == int(np.argmax(ref))
    sel = int(np.argmax(ev))
    m.update_sub_dataset((cand_x[sel:sel + 1], cand_y[sel:sel + 1]), 0, is_append=True)
    assert m.params.cache[0].needs_update and m.dataset[0].x.shape[0] == 41 + it


# ---- d acquisition / d x_query (bayesopt.py:116-125) and the continuous BO loop ------------------------------
@pytest.mark.parametrize('acq', ['ei', 'pi', 'ucb'])
@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mlp,mname', [(False, 'constant'), (True, 'linear_mlp'), (False, 'linear'), (True, 'zero'),
                                       (False, 'linear_mlp'), (True, 'linear')])
@pytest.mark.parametrize('n_obs', [0, 150])
def test_acquisition_value_and_grad_vs_oracle(gpu_ctx, acq, kname, mlp, mname, n_obs):
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(23)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  x, y = helpers.synthetic_task(rng, 150, d)
  x2, y2 = helpers.synthetic_task(rng, 20, d)
  key = 'test'
  ds = {'other': defs.SubDataset(x2, y2), 'third': defs.SubDataset(x2[:5], y2[:5])}
  if n_obs:
    ds[key] = defs.SubDataset(x, y)
  xq = rng.uniform(size=(7, d))
  cov_n = getattr(kernel, kname + ('_mlp' if mlp else '')); cov_o = getattr(o, kname + ('_mlp' if mlp else ''))
  m = gp.GP(ds, getattr(mean, mname), cov_n, pn, utils.DEFAULT_WARP_FUNC)
  fn = {'ei': acfun.expected_improvement, 'pi': acfun.probability_of_improvement, 'ucb': acfun.ucb}[acq]
  val, grad = fn.value_and_grad(model=m, sub_dataset_key=key, x_queries=xq)
  if acq == 'ei':
    param = float(np.max(y)) if n_obs else 0.0
  elif acq == 'pi':
    param = float(np.max(y)) + 0.1 if n_obs else 0.0
  else:
    param = 3.0
  noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
  n_iid = len(ds)
  vo, go = o.acquisition_value_and_grad(acq, getattr(o, mname), cov_o, po, x if n_obs else None, y if n_obs else None,
                                        xq, param, WFO, add_noise=noise, scale=n_iid / (n_iid - 1.0))
  assert val.shape == (7, 1) and grad.shape == (7, d)
  np.testing.assert_allclose(val, vo, rtol=1e-8, atol=1e-10)
  assert np.max(np.abs(grad - go)) <= 1e-7 * max(np.max(np.abs(go)), 1e-3)
  np.testing.assert_allclose(val, fn(model=m, sub_dataset_key=key, x_queries=xq), rtol=1e-8, atol=1e-10)


def test_acquisition_grad_fp32_and_many_queries(gpu_ctx):
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(24)
  d = 4
  model = helpers.make_model(rng, 'constant', False, d)
  po, _ = _pair(model)
  model32 = {k: np.asarray(v, dtype=np.float32) for k, v in model.items()}
  x, y = helpers.synthetic_task(rng, 300, d)
  xq = rng.uniform(size=(1500, d))      # more than one 1024-query pass
  m = gp.GP({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}, mean.constant, kernel.matern52,
            defs.GPParams(model=model32, config={}), utils.DEFAULT_WARP_FUNC)
  val, grad = acfun.ucb.value_and_grad(model=m, sub_dataset_key=0, x_queries=xq.astype(np.float32))
  noise = float(np.squeeze(o.retrieve_params(po, ['noise_variance'], WFO)[0]))
  vo, go = o.acquisition_value_and_grad('ucb', o.constant, o.matern52, po, x, y, xq, 3.0, WFO, add_noise=noise)
  assert val.dtype == np.float32 and grad.dtype == np.float64
  assert np.max(np.abs(val - vo)) <= 5e-3 * np.max(np.abs(vo))
  assert np.max(np.abs(grad - go)) <= 2e-2 * np.max(np.abs(go))


def test_sample_from_gp_and_random_dataset(gpu_ctx):
  defs, _, _, gp, kernel, mean, _, utils = _native()
  from hyperbo_amd.bo_utils import data
  rng = np.random.default_rng(25)
  d, n = 2, 30
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = _pair(model)
  x = rng.uniform(size=(n, d))
  ys = gp.sample_from_gp(3, mean.constant, kernel.squared_exponential, pn, x, utils.DEFAULT_WARP_FUNC, num_samples=20000)
  assert ys.shape == (n, 20000)
  cov = o.squared_exponential(po, x, warp_func=WFO) + np.eye(n) * (o.retrieve_params(po, ['noise_variance'], WFO)[0] + 1e-6)
  mu = o.constant(po, x, warp_func=WFO)[:, 0]
  assert np.max(np.abs(ys.mean(axis=1) - mu)) < 0.05
  assert np.max(np.abs(np.cov(ys) - cov)) < 0.06 * np.max(np.abs(cov))
  dataset, key, queried = data.random(4, mean.constant, kernel.squared_exponential, pn, d, n_observed=5, n_queries=40,
                                      n_func_historical=3, m_points_historical=25, warp_func=utils.DEFAULT_WARP_FUNC)
  assert key == 3 and set(dataset) == {0, 1, 2, 3} and dataset[0].x.shape == (25, d) and dataset[0].y.shape == (25, 1)
  assert dataset[3].x.shape == (5, d) and queried.x.shape == (40, d) and queried.y.shape == (40, 1)


@pytest.mark.parametrize('method', ['svd', 'eigh', 'cholesky'])
def test_sample_from_gp_methods(gpu_ctx, method):
  """gp_test.py:279-303 (shape for 'svd' and 'cholesky') + the law of the draws: for every factorisation method of
  jax.random.multivariate_normal the draws are mean + F z with F F^T = K + (noise + eps) I, so z recovered through the oracle's
  covariance must be white: F^-1 (y - mu) with F = chol(cov) has identity covariance."""
  defs, _, _, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(5)
  nx, num_samples = 20, 10
  vx = rng.normal(size=(nx, 1))
  params = defs.GPParams(model={'constant': 5., 'lengthscale': 1., 'signal_variance': 1.0, 'noise_variance': 0.01})
  vy = gp.sample_from_gp(0, mean.constant, kernel.squared_exponential, params, vx, num_samples=num_samples, method=method)
  assert vy.shape == (nx, num_samples)
  big = gp.sample_from_gp(1, mean.constant, kernel.squared_exponential, params, vx, num_samples=40000, method=method)
  po = o.GPParams(model=params.model)
  cov = o.squared_exponential(po, vx) + np.eye(nx) * (0.01 + 1e-6)
  white = spla.solve_triangular(np.linalg.cholesky(cov), big - 5.0, lower=True)
  assert np.max(np.abs(np.cov(white) - np.eye(nx))) < 0.04 and np.max(np.abs(white.mean(axis=1))) < 0.03
  with pytest.raises(ValueError):
    gp.sample_from_gp(0, mean.constant, kernel.squared_exponential, params, vx, method='qr')


def test_continuous_bayesopt_loop(gpu_ctx):
  """bayesopt.py:75-133 with the native acquisition gradient driving SciPy's L-BFGS-B: every proposed point is
  at least as good (in acquisition value) as the best random candidate it started from, and the incumbent improves."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  from hyperbo_amd.bo_utils import bayesopt
  rng = np.random.default_rng(26)
  d = 2
  model = helpers.make_model(rng, 'constant', False, d)
  f = lambda xx: -np.sum((np.atleast_2d(xx) - 0.3)**2, axis=1, keepdims=True)
  x0 = rng.uniform(size=(4, d))
  ds = {'hist': defs.SubDataset(*helpers.synthetic_task(rng, 30, d)), 'test': defs.SubDataset(x0, f(x0))}
  m = gp.GP(ds, mean.constant, kernel.matern52, defs.GPParams(model=model, config={}), utils.DEFAULT_WARP_FUNC)
  checks = []
  def sampler(key, dim):
    cands = key.uniform(size=(64, dim))
    checks.append(float(np.max(acfun.ucb(model=m, sub_dataset_key='test', x_queries=cands))))
    return cands
  best0 = float(np.max(ds['test'].y))
  out = bayesopt.bayesopt(5, m, 'test', f, acfun.ucb, iters=6, input_sampler=sampler)
  assert out.x.shape == (10, d) and out.y.shape == (10, 1) and np.all(out.x >= 0) and np.all(out.x <= 1)
  assert float(np.max(out.y)) > best0
  assert bayesopt.get_best_datapoint(out)[1] == np.max(out.y)
  # simulated variant on a candidate pool incl. random search
  pool = defs.SubDataset(rng.uniform(size=(50, d)), None)
  pool = defs.SubDataset(pool.x, f(pool.x))
  out2 = bayesopt.simulated_bayesopt(m, 'test', pool, acfun.expected_improvement, iters=3)
  assert out2.x.shape == (13, d)
  out3 = bayesopt.simulated_bayesopt(m, 'test', pool, acfun.random_search, iters=2, random_key=1)
  assert out3.x.shape == (15, d)
  with pytest.raises(ValueError):
    bayesopt.simulated_bayesopt(m, 'test', pool, acfun.rand, iters=1)


@pytest.mark.parametrize('acname', ['expected_improvement', 'probability_of_improvement', 'ucb3', 'random_search', 'ucb2', 'ucb'])
def test_run_bayesopt_synthetic(gpu_ctx, acname):
  """hyperbo/bo_utils/bayesopt_test.py:45-103 (test_run_synthetic) and data_test.py:45-83 (test_dataset_shape) on the
  native path: data.random -> run_bayesopt -> shapes and the best-query invariant."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  from hyperbo_amd.bo_utils import bayesopt, const, data
  params = defs.GPParams(model={'constant': 5., 'lengthscale': 1., 'signal_variance': 1.0, 'noise_variance': 0.01},
                         config={'method': 'adam', 'learning_rate': 1e-5, 'beta': 0.9, 'max_training_step': 1})
  mean_func, cov_func = mean.constant, kernel.squared_exponential
  dataset, key, queried = data.random(key=0, mean_func=mean_func, cov_func=cov_func, params=params, dim=5, n_observed=0,
                                      n_queries=30, n_func_historical=2, m_points_historical=10)
  assert len(dataset) == 3
  for i in range(2):
    assert dataset[i].x.shape == (10, 5) and dataset[i].y.shape == (10, 1) and dataset[i].aligned is None
  assert dataset[key].x.shape == (0, 5) and queried.x.shape == (30, 5) and queried.y.shape == (30, 1)
  observations, queries, out_params = bayesopt.run_bayesopt(
      dataset=dataset, sub_dataset_key=key, queried_sub_dataset=queried, mean_func=mean_func, cov_func=cov_func,
      init_params=params, ac_func=const.ACFUN[acname], iters=3, init_random_key=0)
  assert observations[0].shape == (3, 5) and observations[1].shape == (3, 1)
  assert queries[0].shape == (5,) and queries[1] == np.max(queried.y)
  assert set(out_params.model) == set(params.model)
  # data_test.py:45-83 with observed points
  ds2, k2, q2 = data.random(key=1, mean_func=mean_func, cov_func=cov_func, params=params, dim=5, n_observed=20, n_queries=10,
                            n_func_historical=3, m_points_historical=7)
  assert k2 == 3 and ds2[3].x.shape == (20, 5) and ds2[3].y.shape == (20, 1) and q2.x.shape == (10, 5)
  assert all(ds2[i].x.shape == (7, 5) and ds2[i].y.shape == (7, 1) for i in range(3))
  with pytest.raises(NotImplementedError):
    bayesopt.run_bayesopt(dataset, key, queried, mean_func, cov_func, params, acfun.ucb, 1, method=const.HBO_SS)


@pytest.mark.parametrize('kname,mlp,mname', CASES)
def test_incremental_cache_append_matches_refactorisation(gpu_ctx, kname, mlp, mname):
  """O(N^2) row append (hbo_cache_append) vs the reference behaviour (re-factorise from scratch)."""
  defs, linalg, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(23)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  po, pn = _pair(model)
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  mo, mn = getattr(o, mname), getattr(mean, mname)
  x, y = helpers.synthetic_task(rng, 120, d)
  xa, ya = helpers.synthetic_task(rng, 11, d)
  xq = rng.uniform(size=(40, d))
  m = gp.GP({0: defs.SubDataset(x, y)}, mn, kn, pn, utils.DEFAULT_WARP_FUNC)
  m.predict(xq, 0)
  h0 = m.params.cache[0].handle
  for i in range(0, 11, 4):                     # appends of 4, 4, 3 rows (crosses the 128-row tile edge)
    m.update_sub_dataset((xa[i:i + 4], ya[i:i + 4]), 0, is_append=True)
    mu, var = m.predict(xq, 0)
    assert m.params.cache[0].handle is h0 or m.dataset[0].x.shape[0] > 128   # updated in place while capacity lasts
    xs, ys = m.dataset[0].x, m.dataset[0].y
    mu_o, var_o = o.predict(mo, ko, po, xs, ys, xq, WFO)
    mu_o, var_o = o.gp_predict_postprocess(po, {0: o.SubDataset(xs, ys)}, mu_o, var_o, WFO, False, True, True)
    assert helpers.rel_err(mu, mu_o) < 1e-8 and helpers.rel_err(var, var_o) < 1e-8
    cho, kio, _ = o.solve_gp_linear_system(mo, ko, po, xs, ys, WFO)
    assert helpers.rel_err(m.params.cache[0].chol, cho) < 1e-9 and helpers.rel_err(m.params.cache[0].kinvy, kio) < 1e-8
  assert m.dataset[0].x.shape[0] == 131          # 120 + 11 > 128: the last append re-factorised
  # a replace (not append) always re-factorises
  m.update_sub_dataset((x[:50], y[:50]), 0)
  mu, _ = m.predict(xq, 0)
  mu_o, _ = o.predict(mo, ko, po, x[:50], y[:50], xq, WFO)
  assert helpers.rel_err(mu, mu_o) < 1e-8


@pytest.mark.parametrize('n,mq', [(1100, 40), (2300, 129), (2300, 700)])
def test_posterior_at_few_candidates_split_along_k(gpu_ctx, n, mq):
  """gp.predict (gp.py:242-305) with few candidates against a cache of >= 8 blocks: the triangular product V = L^-1 Kxq is cut
  into K chunks (one workgroup per (row tile, chunk), partial products summed and squared by a second kernel) instead of one
  workgroup walking a row tile's whole K range -- the BO-step shape.  9 / 18 blocks, 1 / 2 / 6 column tiles (the last one is above
  the split threshold at 18 blocks: the one-pass product), mean and variance against the oracle; EI through the same path."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(n + mq)
  d = 4
  model = helpers.make_model(rng, 'constant', False, d)
  po, pn = _pair(model)
  x, y = helpers.synthetic_task(rng, n, d)
  xq = rng.uniform(size=(mq, d))
  m = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.matern52, pn, utils.DEFAULT_WARP_FUNC)
  mu, var = m.predict(xq, 0)
  mu_o, var_o = o.predict(o.constant, o.matern52, po, x, y, xq, WFO)
  mu_o, var_o = o.gp_predict_postprocess(po, {0: o.SubDataset(x, y)}, mu_o, var_o, WFO, False, True, True)
  assert helpers.rel_err(mu, mu_o) < 1e-9 and helpers.rel_err(var, var_o) < 1e-9
  ei = acfun.expected_improvement(model=m, sub_dataset_key=0, x_queries=xq)
  assert np.isfinite(ei).all() and (ei >= 0).all()
  # the same cache in fp32 (the reference's default dtype): few candidates take the fp32-MFMA split product, not the bf16 planes
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  m32 = gp.GP({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}, mean.constant, kernel.matern52,
              defs.GPParams(model=to32(model)), utils.DEFAULT_WARP_FUNC)
  mu32, var32 = m32.predict(xq.astype(np.float32), 0)
  assert mu32.dtype == np.float32 and helpers.rel_err(mu32, mu_o) < 5e-4 and helpers.rel_err(var32, var_o) < 5e-4


def test_append_stops_at_a_row_that_breaks_the_factorisation(gpu_ctx):
  """hbo_cache_append on the device: a row whose pivot is not positive (here: a NaN input, the deterministic way to get one) raises
  the failure word; the rows before it stay appended, the call reports HBO_NOT_PD, the Python cache re-factorises and ends up
  with the NaN cache the reference would have (gp.py:552-560: jax's Cholesky yields NaN, nothing raises).  A valid dataset
  afterwards works again."""
  import ctypes as C
  from hyperbo_amd import _native as nat
  defs, linalg, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(31)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  x, y = helpers.synthetic_task(rng, 90, d)
  xa, ya = helpers.synthetic_task(rng, 3, d)
  xa[1, 0] = np.nan
  xq = rng.uniform(size=(10, d))
  m = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.squared_exponential, defs.GPParams(model=model), utils.DEFAULT_WARP_FUNC)
  mu0, _ = m.predict(xq, 0)
  assert np.isfinite(mu0).all()
  cache = m.params.cache[0]
  # the C entry point itself: stops at row 1 of 3
  from hyperbo_amd import _model
  bm = _model.BuiltModel(mean.constant, kernel.squared_exponential, m.params, utils.DEFAULT_WARP_FUNC, np.float64, d, eps=1e-6)
  rc = nat.lib().hbo_cache_append(gpu_ctx.handle, bm.ref(), cache.handle.handle, nat.ptr(np.ascontiguousarray(xa)), 3, nat.ptr(np.ascontiguousarray(ya)))
  assert rc == nat.HBO_NOT_PD
  # through the GP object: the same rows, no exception, NaN posterior like a fresh factorisation of the same data
  m2 = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.squared_exponential, defs.GPParams(model=model), utils.DEFAULT_WARP_FUNC)
  m2.predict(xq, 0)
  m2.update_sub_dataset((xa, ya), 0, is_append=True)
  mu, var = m2.predict(xq, 0)
  assert m2.dataset[0].x.shape[0] == 93 and np.isnan(mu).all()
  m2.update_sub_dataset((x, y), 0)
  mu1, _ = m2.predict(xq, 0)
  assert helpers.rel_err(mu1, mu0) < 1e-12


@pytest.mark.parametrize('objective_name', ['nll', 'ekl'])
def test_adam_batches_gathered_on_the_device_match_host_sub_sampling(gpu_ctx, objective_name):
  """infer_parameters' Adam loop keeps the dataset resident in HBM and gathers every step's batch there from freshly drawn row
  indices (hbo_dataset_subsample) -- the reference indexes device arrays with jax.random.permutation (data_utils.py:72-100,
  gp.py:101-111).  Same seed, same draws: the trajectory must equal, to rounding, the one through the host iterator
  (sub_sample_dataset_iterator + one upload per step), taken here by hiding the objective's device-batch capability.
  Ragged tasks, one of them smaller than the batch (kept whole), aligned sub-datasets for the divergence objective."""
  defs, _, _, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(17)
  d = 3
  data = {}
  xal = rng.uniform(size=(60, d))
  for k, n in enumerate((90, 61, 33, 130)):
    x, y = helpers.synthetic_task(rng, n, d)
    data[k] = defs.SubDataset(x, y)
  data['al'] = defs.SubDataset(xal, rng.normal(size=(60, 5)), aligned='g')
  objective = getattr(objectives, objective_name)
  def hidden(**kw):
    return objective(**kw)
  def hidden_vg(**kw):
    return objective.value_and_grad(**kw)
  hidden.value_and_grad = hidden_vg                      # no accepts_device_batch: host batches
  model = helpers.make_model(rng, 'constant', False, d)
  out = []
  for obj_fn in (objective, hidden):
    p = defs.GPParams(model={k: np.array(v, copy=True) for k, v in model.items()},
                      config={'method': 'adam', 'batch_size': 40, 'max_training_step': 6, 'learning_rate': 0.05, 'objective': obj_fn})
    losses = []
    res = gp.infer_parameters(mean.constant, kernel.squared_exponential, p, data, utils.DEFAULT_WARP_FUNC, obj_fn, key=5,
                              callback=lambda i, m_, l_: losses.append(l_))
    out.append((losses, helpers.flatten(res.model)))
  # (the same rows in every batch; the library orders a batch's tasks by size, and a gathered batch inherits the resident order among
  #  equal sizes while an uploaded one keeps the dict's: the sum over tasks is taken in another order -- last-bit differences)
  np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-13)
  np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-11, atol=1e-13)


# ---- multi-GPU plumbing: libhbo's RCCL binding and the self-spawning bench ------------------------------------------
def test_rccl_single_rank_allreduce(gpu_ctx):
  """hbo_comm_* with nranks = 1 (what tools/rccl_smoke.py does): id, init, all-reduce (identity), destroy, re-init."""
  from hyperbo_amd import parallel
  for _ in range(2):
    comm = parallel.RcclComm(gpu_ctx, 0, 1, lambda b: b)
    x = np.arange(37, dtype=np.float64) * 0.5 - 3
    y = comm.allreduce_sum(x)
    assert np.array_equal(x, y)
    for _ in range(3):
      y = comm.allreduce_sum(y)
    assert np.array_equal(x, y)
    comm.close()


def test_sharded_objective_entry_point_reduces_on_the_device(gpu_ctx):
  """hbo_objective_sharded (objectives.py:181-195 sharded over ranks): with no communicator and with a one-rank RCCL
  communicator the device-side [nll, count, grad] reduction must equal hbo_objective's host loop bit for bit -- NLL, EKL and
  an MLP model -- and an EMPTY shard must come back as zeros with count 0 (a rank beyond the task count)."""
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  from hyperbo_amd import parallel
  rng = np.random.default_rng(77)
  d = 3
  for mname, mlp in (('constant', False), ('linear_mlp', True)):
    model = helpers.make_model(rng, mname, mlp, d)
    p = defs.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES})
    full = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i, n in enumerate([300, 170, 260, 90, 410])}
    kn = kernel.squared_exponential_mlp if mlp else kernel.squared_exponential
    mn = getattr(mean, mname)
    v0, g0 = objectives.nll_value_and_grad(mn, kn, p, full, utils.DEFAULT_WARP_FUNC)
    comm = parallel.RcclComm(gpu_ctx, 0, 1, lambda b: b)
    try:
      v1, g1 = objectives.nll_value_and_grad(mn, kn, p, full, utils.DEFAULT_WARP_FUNC, comm=comm)
      assert comm.last_timing is not None and comm.last_timing[0] > 0 and comm.last_timing[1] >= 0
      # (MLP gradients are accumulated with fp64 atomics on the device: two evaluations agree to rounding, not to the bit)
      assert v1 == v0
      if mlp:
        np.testing.assert_allclose(helpers.flatten(g1), helpers.flatten(g0), rtol=1e-11, atol=1e-13)
      else:
        assert np.array_equal(helpers.flatten(g1), helpers.flatten(g0))
      # an empty shard: zeros, count 0 -> the mean over tasks is defined as 0
      ve, ge = objectives.nll_value_and_grad(mn, kn, p, {}, utils.DEFAULT_WARP_FUNC, comm=comm)
      assert ve == 0.0 and not np.any(helpers.flatten(ge))
    finally:
      comm.close()
    # without a communicator the entry point returns the local sums
    dev = objectives.DeviceDataset(full)
    s, cnt, gflat, _ = dev.evaluate_sharded(mn, kn, p, utils.DEFAULT_WARP_FUNC)
    s2, _, gflat2, _ = dev.evaluate(mn, kn, p, utils.DEFAULT_WARP_FUNC, want_grad=True)
    assert cnt == 5.0 and s == s2
    np.testing.assert_allclose(gflat, gflat2, rtol=1e-11 if mlp else 0, atol=1e-13 if mlp else 0)
    dev.close()
  # divergence objective through the same path
  al = {i: defs.SubDataset(*helpers.synthetic_task(rng, 120, d, m=3), aligned=i) for i in range(3)}
  p = defs.GPParams(model=helpers.make_model(rng, 'constant', False, d))
  v0, g0 = objectives.ekl.value_and_grad(mean.constant, kernel.matern52, p, al, utils.DEFAULT_WARP_FUNC)
  comm = parallel.RcclComm(gpu_ctx, 0, 1, lambda b: b)
  try:
    v1, g1 = objectives._divergence(objectives.OBJ_EKL, mean.constant, kernel.matern52, p, al, utils.DEFAULT_WARP_FUNC, True, comm=comm)
  finally:
    comm.close()
  assert v1 == v0 and np.array_equal(helpers.flatten(g1), helpers.flatten(g0))


def test_sharded_objective_rank_with_a_local_failure_still_reaches_the_collective(gpu_ctx):
  """A rank whose local part of hbo_objective_sharded fails (here: a model whose dtype does not match its shard) must not leave
  its peers waiting in the all-reduce: it takes part with NaN in every slot, reports its own error code, and the communicator
  stays usable for the next evaluation (advisor finding, round 3)."""
  import ctypes as C
  defs, _, _, _, kernel, mean, objectives, utils = _native()
  from hyperbo_amd import _model, _native as nat, parallel
  rng = np.random.default_rng(78)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  p = defs.GPParams(model=model)
  full = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i, n in enumerate([120, 70])}
  comm = parallel.RcclComm(gpu_ctx, 0, 1, lambda b: b)
  try:
    dev = objectives.DeviceDataset(full)                      # fp64 shard ...
    bm = _model.BuiltModel(mean.constant, kernel.squared_exponential, defs.GPParams(model={k: np.asarray(v, np.float32) for k, v in model.items()}),
                           utils.DEFAULT_WARP_FUNC, np.float32, d)   # ... evaluated with an fp32 model
    val, cnt = C.c_double(0.0), C.c_double(-1.0)
    g = (C.c_double * bm.layout.total)()
    rc = nat.lib().hbo_objective_sharded(gpu_ctx.handle, bm.ref(), dev._h, objectives.OBJ_NLL, C.byref(val), C.byref(cnt), g, None)
    assert rc not in (nat.HBO_OK, nat.HBO_NOT_PD)             # the rank's own error ...
    assert np.isnan(val.value) and np.isnan(cnt.value) and all(np.isnan(v) for v in g)   # ... and what every rank of the job now sees
    # the communicator took exactly one collective: the next evaluation is in step again
    v1, g1 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC, comm=comm)
    v0, g0 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, full, utils.DEFAULT_WARP_FUNC)
    assert v1 == v0 and np.array_equal(helpers.flatten(g1), helpers.flatten(g0))
    dev.close()
  finally:
    comm.close()


_RCCL_2RANK = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ['HBO_ROOT']); sys.path.insert(0, os.path.join(os.environ['HBO_ROOT'], 'tests'))
import helpers
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
rank, world = int(os.environ['RANK']), 2
group = parallel.SocketGroup(rank, world, int(os.environ['HBO_TEST_PORT']), token='rccl2')
ctx = nat.default_context()
comm = parallel.RcclComm(ctx, rank, world, group.bcast_bytes)
rng = np.random.default_rng(7)
model = helpers.make_model(rng, 'constant', False, 3)
full = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, 3)) for i, n in enumerate([300, 170, 260, 90, 410])}
mine = parallel.shard_dataset(full, rank, world)
v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model), mine, utils.DEFAULT_WARP_FUNC, comm=comm)
out = {'value': v, 'grad': helpers.flatten(g).tolist(), 'torch': 'torch' in sys.modules}
if rank == 0:
  v1, g1 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=model), full, utils.DEFAULT_WARP_FUNC)
  out['single'] = v1; out['single_grad'] = helpers.flatten(g1).tolist()
out['timing'] = list(comm.last_timing or ())
group.barrier(); comm.close(); group.close()
print(json.dumps(out))
"""


def test_rccl_two_rank_sharded_objective(gpu_ctx):
  """Two ranks on two GPUs: RcclComm (unique id over the socket group, ncclAllReduce over xGMI) inside
  nll_value_and_grad(comm=...) must reproduce the single-process mean NLL and gradient.  Needs >= 2 devices."""
  import json, socket, subprocess, sys
  from hyperbo_amd import _native as nat
  if nat.lib().hbo_device_count() < 2:
    pytest.skip('needs two GPUs (the multi-GPU box of the driver)')
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  procs = [subprocess.Popen([sys.executable, '-c', _RCCL_2RANK], stdout=subprocess.PIPE, text=True,
                            env=dict(os.environ, RANK=str(r), HBO_DEVICE=str(r), HBO_TEST_PORT=str(port), HBO_ROOT=root))
           for r in range(2)]
  outs = [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in procs]
  assert all(p.returncode == 0 for p in procs)
  for o_ in outs:
    assert not o_['torch'] and len(o_['timing']) == 2      # the device-resident route (hbo_objective_sharded) was taken
    assert abs(o_['value'] - outs[0]['single']) <= 1e-11 * abs(outs[0]['single'])
    np.testing.assert_allclose(o_['grad'], outs[0]['single_grad'], rtol=1e-9, atol=1e-11)


_FAKE_RCCL_2RANK = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ['HBO_ROOT']); sys.path.insert(0, os.path.join(os.environ['HBO_ROOT'], 'tests'))
import bench, helpers
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
rank, world = int(os.environ['RANK']), 2
group = parallel.SocketGroup(rank, world, int(os.environ['HBO_TEST_PORT']), token='fake2')
ctx = nat.default_context()
comm = parallel.RcclComm(ctx, rank, world, group.bcast_bytes)           # rank 1: hbo_comm_init(rank > 0) with the broadcast id
data, raw = bench.cfg4_inputs()
full = {k: defs.SubDataset(xx, yy) for k, (xx, yy) in data.items()}
mine = parallel.shard_dataset(full, rank, world)
p = defs.GPParams(model=raw)
wf = utils.DEFAULT_WARP_FUNC
dev = objectives.DeviceDataset(mine)
out = {'rank': rank, 'local_tasks': len(mine), 'torch': 'torch' in sys.modules}
# (1) the sharded mean NLL + gradient through the device-buffer all-reduce
v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf, comm=comm)
out['value'] = v; out['grad'] = helpers.flatten(g).tolist(); out['timing'] = list(comm.last_timing or ())
# (2) NaN-contribution path: rank 1's local part fails (one-shot injection) -> it still joins the ONE collective, with NaN in
#     every slot: rank 0's call returns a NaN objective (not a hang), rank 1's its own error code
group.barrier()
if rank == 1:
  ctx.set_option('fault_shard', 1)
try:
  v2, g2 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf, comm=comm)
  out['nan_value'] = bool(np.isnan(v2)); out['nan_grad'] = bool(np.all(np.isnan(helpers.flatten(g2)))); out['nan_raised'] = None
except nat.HboError as e:
  out['nan_raised'] = e.code
# (3) the communicator is still in step: the next evaluation is right again
group.barrier()
v3, g3 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf, comm=comm)
out['value_after'] = v3
# (4) abort path: rank 1 cannot produce even the NaN buffer -> ncclCommAbort; rank 0's all-reduce fails (HBO_ERR_COMM) instead of
#     waiting for ever; on rank 1 later sharded calls report HBO_ERR_COMM until hbo_comm_init builds a new communicator
group.barrier()
if rank == 1:
  ctx.set_option('fault_shard', 2)
codes = []
for _ in range(2):
  try:
    objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf, comm=comm)
    codes.append(0)
  except nat.HboError as e:
    codes.append(e.code)
  if rank == 0:
    break
out['abort_codes'] = codes
group.barrier()
comm.close()
# (5) a new communicator (new id) after the abort: both ranks in step again
comm = parallel.RcclComm(ctx, rank, world, group.bcast_bytes)
v5, _ = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf, comm=comm)
out['value_reinit'] = v5
group.barrier(); comm.close(); dev.close(); group.close()
print(json.dumps(out))
"""


def test_two_ranks_on_one_gpu_through_a_stand_in_collective_library(gpu_ctx):
  """The N > 1 code of the sharded objective, executed on the one GPU of the test box: RCCL refuses two ranks on one device, so
  both ranks load tests/libfake_rccl.so (the five nccl entry points over POSIX shm + hipMemcpy, $HBO_RCCL_LIB) and go through
  hbo_comm_init(rank 1) with the broadcast unique id, hbo_objective_sharded's device-buffer all-reduce on BASELINE cfg 4 (the
  2-rank mean NLL + gradient against tests/golden/cfg4_t64_oracle.npz), the NaN-contribution path, the ncclCommAbort path and
  re-initialisation.  The real RCCL transport (xGMI) is NOT exercised here -- no multi-GPU hardware curve exists yet."""
  import json, socket, subprocess, sys
  import bench
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  fake = os.path.join(root, 'tests', 'libfake_rccl.so')
  if not os.path.exists(fake):
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                           os.path.join(root, 'tests', 'fake_rccl.c'), '-o', fake, '-L/opt/rocm/lib', '-lamdhip64', '-lrt'])
  fx = np.load(os.path.join(GOLDEN, 'cfg4_t64_oracle.npz'))
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'LOCAL_RANK')}
  procs = [subprocess.Popen([sys.executable, '-c', _FAKE_RCCL_2RANK], stdout=subprocess.PIPE, text=True,
                            env=dict(env, RANK=str(r), HBO_DEVICE='0', HBO_TEST_PORT=str(port), HBO_ROOT=root, HBO_RCCL_LIB=fake, HBO_TEST_HOOKS='1'))
           for r in range(2)]
  outs = []
  for p in procs:
    txt = p.communicate(timeout=600)[0]
    assert p.returncode == 0, txt
    outs.append(json.loads(txt.strip().splitlines()[-1]))
  outs.sort(key=lambda d: d['rank'])
  from hyperbo_amd import _native as nat
  assert outs[0]['local_tasks'] + outs[1]['local_tasks'] == 64 and min(o_['local_tasks'] for o_ in outs) >= 24
  nll, gref = float(fx['nll_mean']), fx['grad_mean_flat']
  for o_ in outs:
    assert not o_['torch'] and len(o_['timing']) == 2
    assert abs(o_['value'] - nll) <= 1e-10 * abs(nll)
    helpers.assert_grad_close(np.array(o_['grad']), helpers.unflatten_like(bench.cfg4_inputs()[1], gref), FP64_GRAD_TOL, label='rank %d' % o_['rank'])
    assert abs(o_['value_after'] - nll) <= 1e-10 * abs(nll) and abs(o_['value_reinit'] - nll) <= 1e-10 * abs(nll)
  assert outs[0]['value'] == outs[1]['value'] and outs[0]['grad'] == outs[1]['grad']      # rank-ordered sum: the same bits
  # NaN contribution: rank 0 sees NaN everywhere (or NOT_PD surfaced as NaN), rank 1 its own error
  assert outs[0]['nan_raised'] is None and outs[0]['nan_value'] and outs[0]['nan_grad']
  assert outs[1]['nan_raised'] == nat.HBO_ERR_HIP
  # abort: rank 1 fails locally, then HBO_ERR_COMM until re-init; rank 0's collective fails instead of hanging
  assert outs[1]['abort_codes'] == [nat.HBO_ERR_HIP, nat.HBO_ERR_COMM]
  assert outs[0]['abort_codes'] == [nat.HBO_ERR_COMM]


def test_bench_spawns_its_own_ranks_without_torch(gpu_ctx):
  """`python bench.py --gpus 2` with no launcher: two ranks, n_gpus = 2 in the JSON line, no torch in the ranks.  On a
  one-GPU box both ranks share the device and the all-reduce falls back to host sockets (RCCL refuses duplicate GPUs);
  on the driver's multi-GPU box the same command reports comm = rccl (libhbo, xGMI)."""
  import json, subprocess, sys
  from hyperbo_amd import _native as nat
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'HBO_DEVICE')}
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                      '--no-cpu-baseline', '--no-extra', '--n', '2048'], env=env, stdout=subprocess.PIPE, text=True, timeout=600)
  assert r.returncode == 0, r.stdout
  assert len(r.stdout.strip().splitlines()) == 1, r.stdout      # ONE JSON line on stdout, nothing else
  line = json.loads(r.stdout.strip())
  assert line['n_gpus'] == 2 and line['torch_imported'] is False
  mt = line['multitask']
  assert mt['local_tasks'] == 32 and mt['comm_us'] is not None
  assert abs(mt['nll_unperturbed'] - mt['nll_oracle_fixture']) <= 1e-9 * abs(mt['nll_oracle_fixture'])
  if nat.lib().hbo_device_count() >= 2:
    assert mt['comm'].startswith('rccl (libhbo')


def test_streamed_posterior_chunks_match_single_pass(gpu_ctx):
  """The candidates are processed in chunks (two alternating cross-Gram workspaces, Gram build of chunk i+1 beside the
  triangular product of chunk i): every chunk size -- one pass, ragged last chunk, 128-candidate chunks -- must give
  bit-identical mean / variance / EI, with data and on the prior branch, fp64 and fp32."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(31)
  d = 5
  model = helpers.make_model(rng, 'linear_mlp', True, d)
  x, y = helpers.synthetic_task(rng, 700, d)
  xq = rng.uniform(size=(1000, d))
  for dtype in (np.float64, np.float32):
    cast = lambda t: {k: cast(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=dtype)
    g = gp.GP({0: defs.SubDataset(x.astype(dtype), y.astype(dtype)), 1: defs.SubDataset(np.zeros((0, d), dtype), np.zeros((0, 1), dtype))},
              mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=cast(model), config={'mlp_features': helpers.MLP_FEATURES}),
              utils.DEFAULT_WARP_FUNC)
    ref = None
    try:
      for chunk in (65536, 512, 384, 128):
        gpu_ctx.set_option('post_chunk', chunk)
        cur = (g.predict(xq.astype(dtype), 0), acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq.astype(dtype)),
               g.predict(xq.astype(dtype), 1), acfun.ucb(model=g, sub_dataset_key=1, x_queries=xq.astype(dtype)))
        flat = [cur[0][0], cur[0][1], cur[1], cur[2][0], cur[2][1], cur[3]]
        assert all(np.isfinite(a).all() for a in flat)
        if ref is None:
          ref = flat
        else:
          for a, b in zip(flat, ref):
            assert np.array_equal(a, b), chunk
    finally:
      gpu_ctx.set_option('post_chunk', 8192)
  if dtype == np.float32:
    pass
  # against the oracle (fp64) with small chunks
  gpu_ctx.set_option('post_chunk', 256)
  try:
    g64 = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp,
                defs.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES}), utils.DEFAULT_WARP_FUNC)
    mu, var = g64.predict(xq, 0, with_noise=False, unbiased=False)
  finally:
    gpu_ctx.set_option('post_chunk', 8192)
  po = o.GPParams(model=model, config={'mlp_features': helpers.MLP_FEATURES})
  mu_o, var_o = o.predict(o.linear_mlp, o.matern52_mlp, po, x, y, xq, WFO)
  assert helpers.rel_err(mu, mu_o) < 1e-9 and helpers.rel_err(var, var_o) < 1e-8


def test_pooled_device_buffers_are_safe_to_reuse(gpu_ctx):
  """Datasets and caches hand their device buffers back to a pool on close (GP.train()'s Adam loop re-creates its
  sub-sampled batch every step, gp.py:101-111); the next dataset of the same shape takes them over -- including the
  inverse factor W, whose zeros above the diagonal are only valid for the same size and dtype.  Alternating shapes, dtypes,
  objectives and a posterior cache must all still match the oracle."""
  defs, linalg, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(41)
  d = 3
  model = helpers.make_model(rng, 'constant', False, d)
  po = o.GPParams(model=model)
  wf = utils.DEFAULT_WARP_FUNC
  cast32 = lambda t: {k: cast32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  plan = [(300, np.float64), (300, np.float64), (180, np.float64), (300, np.float32), (180, np.float32), (300, np.float64), (428, np.float64), (300, np.float64)]
  for step, (n, dt) in enumerate(plan):
    data = {k: helpers.synthetic_task(rng, n + 17 * k, d) for k in range(3)}
    dso = {k: o.SubDataset(x, y) for k, (x, y) in data.items()}
    dsn = {k: defs.SubDataset(x.astype(dt), y.astype(dt)) for k, (x, y) in data.items()}
    pn = defs.GPParams(model=cast32(model) if dt == np.float32 else model)
    vo, go = o.nll_value_and_grad(o.constant, o.matern32, po, dso, WFO)
    dev = objectives.DeviceDataset(dsn)
    vn, gn = objectives.nll_value_and_grad(mean.constant, kernel.matern32, pn, dev, wf)
    dev.close()
    tol_v, tol_g = (1e-10, FP64_GRAD_TOL) if dt == np.float64 else (3e-4, 1e-3)
    assert abs(vn - vo) <= tol_v * max(abs(vo), 1.0), (step, n, dt)
    fo, fn = helpers.flatten(go), helpers.flatten(gn)
    helpers.assert_grad_close(gn, go, tol_g, label=str((step, n, dt)))
    if dt == np.float64:   # a posterior cache (its own X / W / S buffers) in between
      x, y = data[0]
      h = linalg.factor(mean.constant, kernel.matern32, pn, x, y, wf)
      chol, kinvy, _ = h.export(); h.close()
      co, ko, _ = o.solve_gp_linear_system(o.constant, o.matern32, po, x, y, WFO)
      assert helpers.rel_err(chol, co) < 1e-10 and helpers.rel_err(kinvy, ko) < 1e-8


# ---- fp32 (the reference's DEFAULT dtype: JAX without x64) across the whole registry -----------------------------------
_FP32_ERR = {}


@pytest.mark.parametrize('kname', helpers.KERNELS)
@pytest.mark.parametrize('mlp', [False, True])
@pytest.mark.parametrize('mname', helpers.MEANS)
def test_fp32_registry_value_grad_posterior_vs_oracle(gpu_ctx, kname, mlp, mname):
  """Every kernel x MLP-or-not x mean of the closed registry in float32 -- the dtype the reference runs in unless
  JAX_ENABLE_X64 is set: NLL (two ragged tasks), ALL gradient leaves, posterior mean / variance and EI against the fp64
  oracle evaluated on the SAME float32-rounded inputs and parameters.  Tolerances: a float32 Cholesky of a jittered Gram
  matrix with noise variance 0.13 (cond ~1e3), about ten times the worst error measured on MI355X (value 7e-7, gradient
  6e-6, mean 4e-5, variance 5e-6): value 1e-5, gradient 1e-4 of max|g|, mean 5e-4, variance 1e-4, EI 2e-3."""
  defs, _, acfun, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(51)
  d = 3
  model = helpers.make_model(rng, mname, mlp, d)
  cast32 = lambda t: {k: cast32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  up64 = lambda t: {k: up64(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float64)
  m32 = cast32(model)
  data = {k: tuple(a.astype(np.float32) for a in helpers.synthetic_task(rng, n, d)) for k, n in enumerate((260, 131))}
  cfg = {'mlp_features': helpers.MLP_FEATURES}
  po = o.GPParams(model=up64(m32), config=dict(cfg)); pn = defs.GPParams(model=m32, config=dict(cfg))
  dso = {k: o.SubDataset(x.astype(np.float64), y.astype(np.float64)) for k, (x, y) in data.items()}
  dsn = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
  ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
  mo, mn = getattr(o, mname), getattr(mean, mname)
  wf = utils.DEFAULT_WARP_FUNC
  vo, go = o.nll_value_and_grad(mo, ko, po, dso, WFO)
  vn, gn = objectives.nll_value_and_grad(mn, kn, pn, dsn, wf)
  fo, fn = helpers.flatten(go), helpers.flatten(gn)
  e_val = abs(vn - vo) / max(abs(vo), 1.0)
  e_grad = max(err / max(norm, 1e-2 * np.max(np.abs(fo))) for _, err, norm in helpers.grad_leaf_errors(gn, go))   # worst leaf, relative to the leaf
  x, y = data[0]
  xq = rng.uniform(size=(64, d)).astype(np.float32)
  g32 = gp.GP(dsn, mn, kn, pn, wf)
  mu, var = g32.predict(xq, 0)
  ei = acfun.expected_improvement(model=g32, sub_dataset_key=0, x_queries=xq)
  assert mu.dtype == np.float32 and var.dtype == np.float32 and ei.dtype == np.float32
  mu_o, var_o = o.predict(mo, ko, po, x.astype(np.float64), y.astype(np.float64), xq.astype(np.float64), WFO)
  mu_o, var_o = o.gp_predict_postprocess(po, dso, mu_o, var_o, WFO, False, True, True)
  ei_o = o.expected_improvement_sub(mu_o, np.sqrt(var_o), float(np.max(y)))
  e_mu = np.max(np.abs(mu - mu_o)) / max(np.max(np.abs(mu_o)), 1.0)
  e_var = np.max(np.abs(var - var_o)) / np.max(np.abs(var_o))
  e_ei = np.max(np.abs(ei - ei_o)) / max(np.max(np.abs(ei_o)), 1e-3)
  _FP32_ERR[(kname, mlp, mname)] = (e_val, e_grad, e_mu, e_var, e_ei)
  if os.environ.get('HBO_GRAD_LOG'):
    with open(os.environ['HBO_GRAD_LOG'], 'a') as f_:
      f_.write('%.3e tol=registry32 value %.3e grad_leaf %.3e mu %.3e var %.3e ei %.3e %s\n' % (e_grad / FP32_REGISTRY_LEAF_TOL, e_val, e_grad, e_mu, e_var, e_ei, (kname, mlp, mname)))
  assert e_val <= 1e-5 and e_grad <= FP32_REGISTRY_LEAF_TOL and e_mu <= 5e-4 and e_var <= 1e-4 and e_ei <= 2e-3, _FP32_ERR[(kname, mlp, mname)]


def test_fp32_registry_error_summary(gpu_ctx):
  """Prints the worst float32 errors of the sweep above (run with -s); fails only if the sweep did not run."""
  if not _FP32_ERR:
    pytest.skip('run together with test_fp32_registry_value_grad_posterior_vs_oracle')
  worst = np.max(np.array(list(_FP32_ERR.values())), axis=0)
  print('fp32 worst relative errors (value, grad, mu, var, ei):', ' '.join('%.2e' % w for w in worst))
  assert len(_FP32_ERR) == 32


@pytest.mark.gpu
@pytest.mark.parametrize('kname', ['squared_exponential', 'matern32', 'matern52', 'dot_product'])
def test_fp32_posterior_on_bf16_matrix_cores_is_as_accurate_as_fp32_mfma(gpu_ctx, kname):
  """fp32 caches run the posterior product V = L^-1 Kxq on the bf16 MFMA from exact three-way splits of both operands
  (csrc/post3.hip; gp.py:295-305 is the reference's solve).  Claim under test: the result is fp32-accurate -- its error
  against the fp64 path is not larger than that of the fp32-MFMA product it replaces (same inputs, option off) -- for every
  kernel of the registry, ragged sizes, several chunks; and the planes of W follow a row append of the cache."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(77)
  d, n, M = 6, 1300, 7000      # (enough candidates for the matrix-core product: below 2 x CUs tiles the split-K fp32 path takes over)
  isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
  model = {'lengthscale': isp(np.full(d, 0.8)), 'signal_variance': isp(1.2), 'noise_variance': isp(3e-2), 'constant': np.array(0.3),
           'dot_prod_sigma': isp(1.5), 'dot_prod_bias': np.array(0.4)}
  x = rng.uniform(size=(n, d)); y = np.sin(3.0 * x[:, :2].sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(n, 1))
  xq = rng.uniform(size=(M, d))
  cov = getattr(kernel, kname)
  out = {}
  try:
    gpu_ctx.set_option('post_chunk', 2048)
    for name, dt, opt, opt2 in (('f64', np.float64, 0, 0), ('mfma', np.float32, 0, 0), ('bf16x3', np.float32, 1, 0), ('f16x2', np.float32, 1, 1)):
      gpu_ctx.set_option('post_bf16x3', opt)
      gpu_ctx.set_option('post_f16x2', opt2)
      cast = lambda t: {k: cast(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=dt)
      g = gp.GP({0: defs.SubDataset(x[:n - 40].astype(dt), y[:n - 40].astype(dt))}, mean.constant, cov, defs.GPParams(model=cast(model)),
                utils.DEFAULT_WARP_FUNC)
      mu, var = g.predict(xq.astype(dt), 0)
      ei = acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq.astype(dt))
      # the cache grows by 40 rows (O(N^2) append): W changes, its bf16 planes must be rebuilt
      g.update_sub_dataset(defs.SubDataset(x[n - 40:].astype(dt), y[n - 40:].astype(dt)), 0, is_append=True)
      mu2, var2 = g.predict(xq.astype(dt), 0)
      out[name] = [np.asarray(a, np.float64).ravel() for a in (mu, var, ei, mu2, var2)]
  finally:
    gpu_ctx.set_option('post_bf16x3', 1)
    gpu_ctx.set_option('post_f16x2', 1)
    gpu_ctx.set_option('post_chunk', 8192)
  for j, what in enumerate(('mean', 'variance', 'EI', 'mean after append', 'variance after append')):
    ref = out['f64'][j]
    scale = max(np.abs(ref).max(), 1e-6)
    e_mfma = np.abs(out['mfma'][j] - ref).max() / scale
    # bf16x3: exact products; f16x2 (the default for the stationary covariances; the dot product keeps bf16x3): two-way fp16 split,
    # 2^-22 per product -- both must be fp32-accurate: no worse than the fp32-MFMA product (1.5x + 2 ulp of slack for the max statistic)
    for path in ('bf16x3', 'f16x2'):
      e_p = np.abs(out[path][j] - ref).max() / scale
      assert np.isfinite(out[path][j]).all(), (path, what)
      assert e_p <= 1.5 * e_mfma + 2.4e-7, (kname, path, what, e_p, e_mfma)
      assert e_p < 2e-3, (kname, path, what, e_p)
  if kname == 'dot_product':
    assert all(np.array_equal(a, b) for a, b in zip(out['f16x2'], out['bf16x3']))      # unbounded kernel: the f16x2 option does not apply
  else:
    assert not np.array_equal(out['f16x2'][1], out['bf16x3'][1])                      # the other path did run


@pytest.mark.gpu
@pytest.mark.parametrize('n,M', [(1, 1), (5, 3), (128, 128), (129, 1), (130, 257), (511, 130), (2049, 129)])
def test_fp32_posterior_bf16x3_edge_sizes(gpu_ctx, n, M):
  """Block-boundary sizes of the bf16 matrix-core product (one training point, exactly one 128-block, one past a block, a single
  candidate, ragged chunks): same mean, variance and EI as the fp32-MFMA product to fp32 rounding."""
  defs, _, acfun, gp, kernel, mean, _, utils = _native()
  rng = np.random.default_rng(1000 * n + M)
  d = 3
  isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
  model = {k: np.asarray(v, np.float32) for k, v in
           {'lengthscale': isp(np.full(d, 0.6)), 'signal_variance': isp(1.0), 'noise_variance': isp(1e-2), 'constant': np.array(0.1)}.items()}
  x = rng.uniform(size=(n, d)).astype(np.float32); y = np.sin(x.sum(1, keepdims=True)).astype(np.float32)
  xq = rng.uniform(size=(M, d)).astype(np.float32)
  out = []
  try:
    gpu_ctx.set_option('post_chunk', 128)
    for opt in (0, 1):
      gpu_ctx.set_option('post_bf16x3', opt)
      g = gp.GP({0: defs.SubDataset(x, y)}, mean.constant, kernel.matern52, defs.GPParams(model=model), utils.DEFAULT_WARP_FUNC)
      mu, var = g.predict(xq, 0)
      ei = acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq)
      out.append([np.asarray(a, np.float64).ravel() for a in (mu, var, ei)])
  finally:
    gpu_ctx.set_option('post_bf16x3', 1)
    gpu_ctx.set_option('post_chunk', 8192)
  for a, b, what in zip(out[0], out[1], ('mean', 'variance', 'EI')):
    assert np.isfinite(b).all(), what
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), (what, np.abs(a - b).max())
