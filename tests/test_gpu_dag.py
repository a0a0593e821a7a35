"""The resident tile-task schedule of the factorisation phase (hyperbo_amd/csrc/dag.hip, option `dag`; off by default, see
profiles/r03_dag.md) against the launch schedule: same tiles, same K order inside a tile, so the NLL and every gradient leaf
have to agree to the last bits -- for a single matrix, for ragged batches, in fp32, through `hbo_factor` / the posterior, and
after a forced wall-clock abort (the context has to repeat the evaluation on the launch schedule and stay there).
Reference for the arithmetic: hyperbo/basics/linalg.py:29-33, hyperbo/gp_utils/objectives.py:109-210."""
import numpy as np
import pytest

import helpers
from oracle import hyperbo_oracle as o

pytestmark = pytest.mark.gpu


def _native():
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
  return defs, gp, kernel, mean, objectives, utils


def _se_model(rng, d, dtype=np.float64):
  return {'lengthscale': helpers.inv_softplus(np.full(d, np.sqrt(d) * 0.3)).astype(dtype),
          'signal_variance': np.array(helpers.inv_softplus(1.0), dtype=dtype),
          'noise_variance': np.array(helpers.inv_softplus(1e-2), dtype=dtype), 'constant': np.array(0.1, dtype=dtype)}


@pytest.fixture
def dag_ctx(gpu_ctx):
  gpu_ctx.set_option('lookahead', 2)     # the resident schedule is a look-ahead schedule: keep it on at every size tested here
  yield gpu_ctx
  gpu_ctx.set_option('lookahead', 1)
  gpu_ctx.set_option('dag', 0)
  gpu_ctx.set_option('dag_timeout_ms', 2000)
  gpu_ctx.set_option('dag_trtri', 64)


def _eval(objectives, mean, kernel, utils, p, dev):
  v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
  return float(v), helpers.flatten(g)


@pytest.mark.parametrize('form', [1, 2])
@pytest.mark.parametrize('sizes', [(1024,), (1500,), (2304,), (1100, 1536, 1290), (1030, 40, 2050, 1024)])
def test_resident_schedule_equals_launch_schedule(dag_ctx, form, sizes):
  defs, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(7 + len(sizes) + sizes[0])
  d = 5
  ds = {f't{i}': defs.SubDataset(*helpers.synthetic_task(rng, n, d)) for i, n in enumerate(sizes)}
  dev = objectives.DeviceDataset(ds)
  p = defs.GPParams(model=_se_model(rng, d))
  dag_ctx.set_option('dag', 0)
  v0, g0 = _eval(objectives, mean, kernel, utils, p, dev)
  for cut in (64, 0, 40):
    dag_ctx.set_option('dag', form)
    dag_ctx.set_option('dag_trtri', cut)
    v1, g1 = _eval(objectives, mean, kernel, utils, p, dev)
    assert abs(v1 - v0) <= 1e-13 * abs(v0), (form, cut)
    assert np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0)), (form, cut)
  # and against the CPU oracle on the first task (the launch schedule is tested against it everywhere else)
  x0, y0 = ds['t0'].x, ds['t0'].y
  vo, _ = o.nll_value_and_grad(o.constant, o.squared_exponential, o.GPParams(model=p.model), {'t0': o.SubDataset(x0, y0)}, o.DEFAULT_WARP_FUNC)
  dev0 = objectives.DeviceDataset({'t0': ds['t0']})
  v2, _ = _eval(objectives, mean, kernel, utils, p, dev0)
  assert abs(v2 - vo) <= 1e-10 * abs(vo)


@pytest.mark.parametrize('form', [1, 2])
def test_resident_schedule_fp32_and_posterior(dag_ctx, form):
  """fp32 factor + posterior through hbo_factor (gp.py:242-305): the cache built by the resident schedule predicts like
  the one built by the launch schedule."""
  defs, gp, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(11)
  d, n = 4, 1400
  x, y = helpers.synthetic_task(rng, n, d, dtype=np.float32)
  xq = rng.uniform(size=(300, d)).astype(np.float32)
  p = defs.GPParams(model=_se_model(rng, d, np.float32))
  out = {}
  for f in (0, form):
    dag_ctx.set_option('dag', f)
    mu, var = gp.predict(mean.constant, kernel.squared_exponential, p, x, y, xq, warp_func=utils.DEFAULT_WARP_FUNC)
    out[f] = (np.asarray(mu, dtype=np.float64), np.asarray(var, dtype=np.float64))
  dag_ctx.set_option('dag', 0)
  p64 = defs.GPParams(model={k: np.asarray(v, dtype=np.float64) for k, v in p.model.items()})
  mu64, var64 = gp.predict(mean.constant, kernel.squared_exponential, p64, x.astype(np.float64), y.astype(np.float64), xq.astype(np.float64),
                           warp_func=utils.DEFAULT_WARP_FUNC)
  # (fp32: the launch schedule runs its trailing updates on the bf16 matrix cores, the tile tasks on fp32 MFMA -- the same accuracy
  #  class, not the same rounding: each against the fp64 path, at the fp32 tolerances of the parity tests)
  for f in (0, form):
    assert np.max(np.abs(out[f][0] - mu64)) <= 5e-4 * (1 + np.max(np.abs(mu64))), f
    assert np.max(np.abs(out[f][1] - var64)) <= 1e-3 * np.max(np.abs(var64)), f


@pytest.mark.parametrize('form', [1, 2])
def test_not_positive_definite_reports_nan_not_a_hang(dag_ctx, form):
  """linalg.py:29-33 semantics under the resident schedule: a pivot that fails in a later diagonal block gives NaN, and the
  counters still advance (the call returns in milliseconds, not at the wall-clock bound)."""
  import time
  from hyperbo_amd.basics import linalg
  n = 1400
  a = np.eye(n); a[1300, 1300] = -1.0   # fails in the eleventh diagonal block
  dag_ctx.set_option('dag', form)
  t0 = time.perf_counter()
  chol, x = linalg.solve_linear_system(a, np.ones((n, 1)))
  assert time.perf_counter() - t0 < 1.5
  assert np.isnan(chol).all() and np.isnan(x).all()
  b = np.eye(n) * 2.0
  chol, x = linalg.solve_linear_system(b, np.ones((n, 1)))
  assert np.allclose(np.diag(chol), np.sqrt(2.0)) and np.allclose(x, 0.5)


def test_wall_clock_abort_falls_back_to_the_launch_schedule(dag_ctx):
  defs, _, kernel, mean, objectives, utils = _native()
  rng = np.random.default_rng(5)
  d, n = 5, 4096
  dev = objectives.DeviceDataset({'t': defs.SubDataset(*helpers.synthetic_task(rng, n, d))})
  p = defs.GPParams(model=_se_model(rng, d))
  dag_ctx.set_option('dag', 0)
  v0, g0 = _eval(objectives, mean, kernel, utils, p, dev)
  dag_ctx.set_option('dag', 2)
  dag_ctx.set_option('dag_timeout_ms', 1)   # the factorisation of 32 blocks takes longer than 1 ms in this form: guaranteed abort
  v1, g1 = _eval(objectives, mean, kernel, utils, p, dev)
  assert abs(v1 - v0) <= 1e-13 * abs(v0)
  assert np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0))
  # the context is back on the launch schedule until the option is set again
  v2, _ = _eval(objectives, mean, kernel, utils, p, dev)
  assert v2 == v0
