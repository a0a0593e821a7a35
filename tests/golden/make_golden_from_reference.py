"""One-shot pin of the golden fixtures against the JAX REFERENCE itself (last row).

Runs the inputs of every committed fixture (tests/golden/*.npz: the five make_golden.py cases, cfg 1 and the 64 tasks of cfg 4 of
make_golden_configs.py) through the reference's own code -- imported from /root/reference, JAX_ENABLE_X64=1 -- and
writes `<case>_ref.npz` next to them:

  kernel.<k>(params, x[, xq])                               hyperbo/gp_utils/kernel.py:29-183
  mean.<m>(params, x)                                       hyperbo/gp_utils/mean.py:30-79
  obj.neg_log_marginal_likelihood (Cholesky and SVD)        hyperbo/gp_utils/objectives.py:109-210
  jax.value_and_grad of it w.r.t. params.model              hyperbo/gp_utils/gp.py:134
  linalg.solve_gp_linear_system                             hyperbo/basics/linalg.py:72-110
  gp.predict (diagonal and full covariance)                 hyperbo/gp_utils/gp.py:242-305
  acfun.expected_improvement / probability_of_improvement / ucb on a gp.GP   hyperbo/bo_utils/acfun.py:36-185
  obj.multivariate_normal_divergence (ekl) / _euc_distance and their jax.grad   objectives.py:29-106
  gp.infer_parameters: every callback (step, params, loss) of Adam and L-BFGS       gp.py:53-195, basics/lbfgs.py:186-349
      for the four row-f1 cases (TRAIN_CASES below; batch_size above every sub-dataset's size, so that no jax.random
      permutation enters and the trajectory is a function of the inputs alone) -> train_<case>_<method>_ref.npz
  acfun.* on a gp.HGP with parameter samples: the mean over samples                  bo_utils/acfun.py:72-82, gp.py:666-682
      -> hgp_<acq>_ref.npz

tests/test_oracle_pins.py::test_golden_fixtures_match_reference compares oracle/hyperbo_oracle.py with every
`*_ref.npz` it finds; with those files committed the parity chain  HIP == oracle == reference  is closed.

Today neither this container nor the GPU box has jax / flax (no network): the script then says so
and exits 0 without writing anything -- "parity unpinned" stays the honest label until it has run once.

  JAX_ENABLE_X64=1 python tests/golden/make_golden_from_reference.py [/path/to/reference]
"""
import importlib.util
import os
import sys

os.environ.setdefault('JAX_ENABLE_X64', '1')
os.environ.setdefault('JAX_PLATFORMS', 'cpu')

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))   # the repo root: bench.cfg4_inputs()
import helpers  # noqa: E402


# Row f1 / a15 inputs of the widened pin: plain NumPy draws, stored INSIDE the *_ref.npz next to the reference's outputs, so that the
# consuming test (tests/test_oracle_pins.py::test_golden_fixtures_match_reference) re-runs the oracle on exactly these numbers.
TRAIN_CASES = [('squared_exponential', 'constant'), ('matern32', 'zero'), ('matern52_mlp', 'linear_mlp'), ('dot_product_mlp', 'linear')]
TRAIN_METHODS = [('adam', 10, 2e-2), ('lbfgs', 3, None)]
TRAIN_SIZES = (100, 64, 130, 80, 45)
TRAIN_BATCH = 500   # above every size: data_utils.py:72-100 leaves the sub-datasets whole


def train_inputs(kname, mname):
  rng = np.random.default_rng(33)
  d = 2
  model = helpers.make_model(rng, mname, kname.endswith('_mlp'), d)
  data = [helpers.synthetic_task(rng, n, d) for n in TRAIN_SIZES]
  return model, data


def hgp_inputs():
  rng = np.random.default_rng(41)
  d, S = 3, 4
  samples = [helpers.make_model(np.random.default_rng(300 + i), 'constant', False, d) for i in range(S)]
  x, y = helpers.synthetic_task(rng, 60, d)
  x2, y2 = helpers.synthetic_task(rng, 20, d)
  xq = rng.uniform(size=(9, d))
  return samples, x, y, x2, y2, xq


def _load(name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + '.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def main(reference_root='/root/reference'):
  try:
    import jax
    import jax.numpy as jnp
    import flax  # noqa: F401
  except ImportError as e:
    print(f'make_golden_from_reference: jax/flax not importable here ({e}); nothing written -- the fixtures stay '
          'pinned to oracle/hyperbo_oracle.py only (parity unpinned against the reference).')
    return 0
  if not os.path.isdir(os.path.join(reference_root, 'hyperbo')):
    print(f'make_golden_from_reference: {reference_root}/hyperbo not found; nothing written.')
    return 0
  jax.config.update('jax_enable_x64', True)
  sys.path.insert(0, reference_root)
  from hyperbo.basics import definitions as defs
  from hyperbo.basics import linalg
  from hyperbo.bo_utils import acfun
  from hyperbo.gp_utils import gp
  from hyperbo.gp_utils import kernel
  from hyperbo.gp_utils import mean
  from hyperbo.gp_utils import objectives as obj
  from hyperbo.gp_utils import utils

  wf = utils.DEFAULT_WARP_FUNC
  to_jnp = lambda t: {k: to_jnp(v) for k, v in t.items()} if isinstance(t, dict) else jnp.asarray(t, dtype=jnp.float64)
  to_np = lambda a: np.asarray(a, dtype=np.float64)

  def run_case(model, cfg, kname, mname, x, y, x2, y2, xq, ya):
    cov_func, mean_func = getattr(kernel, kname), getattr(mean, mname)
    jm = to_jnp(model)
    params = defs.GPParams(model=jm, config=dict(cfg))
    x, y, xq = jnp.asarray(x), jnp.asarray(y), jnp.asarray(xq)
    dataset = {0: defs.SubDataset(x, y)}
    if x2 is not None:
      dataset[1] = defs.SubDataset(jnp.asarray(x2), jnp.asarray(y2))
    out = {}
    out['gram'] = to_np(cov_func(params, x, warp_func=wf))
    out['cross'] = to_np(cov_func(params, x, xq, warp_func=wf))
    out['mean_x'] = to_np(mean_func(params, x, warp_func=wf))
    nll, key2nll = obj.neg_log_marginal_likelihood(mean_func, cov_func, params, dataset, wf, return_key2nll=True)
    out['nll'] = float(nll); out['nll0'] = float(key2nll[0])
    if 1 in key2nll:
      out['nll1'] = float(key2nll[1])
    out['nll_svd'] = float(obj.neg_log_marginal_likelihood(mean_func, cov_func, params, dataset, wf, use_cholesky=False))

    def loss(m, objective, data):
      return objective(mean_func=mean_func, cov_func=cov_func, params=defs.GPParams(model=m, config=dict(cfg)),
                       dataset=data, warp_func=wf)
    _, g = jax.value_and_grad(lambda m: loss(m, obj.neg_log_marginal_likelihood, dataset))(jm)
    out['grad_flat'] = helpers.flatten(jax.tree.map(to_np, g))
    chol, kinvy, ymu = linalg.solve_gp_linear_system(mean_func=mean_func, cov_func=cov_func, params=params, x=x, y=y,
                                                     warp_func=wf)
    out['chol'], out['kinvy'], out['ymu'] = to_np(chol), to_np(kinvy), to_np(ymu)
    mu, var = gp.predict(mean_func, cov_func, params, x, y, xq, warp_func=wf, full_cov=False)
    _, cov = gp.predict(mean_func, cov_func, params, x, y, xq, warp_func=wf, full_cov=True)
    out['mu'], out['var'], out['cov'] = to_np(mu), to_np(var), to_np(cov)
    model_obj = gp.GP(dataset=dataset, mean_func=mean_func, cov_func=cov_func, params=params, warp_func=wf)
    for name, key in (('expected_improvement', 'ei'), ('probability_of_improvement', 'pi'), ('ucb', 'ucb')):
      out[key] = to_np(getattr(acfun, name)(model=model_obj, sub_dataset_key=0, x_queries=xq))
    if ya is not None:
      aligned = {'al': defs.SubDataset(x, jnp.asarray(ya), aligned='al'), 0: defs.SubDataset(x, y)}
      for objective, key in ((obj.ekl, 'ekl'), (obj.euc, 'euc')):
        v, g = jax.value_and_grad(lambda m: loss(m, objective, aligned))(jm)   # pylint: disable=cell-var-from-loop
        out[key] = float(v); out[key + '_grad_flat'] = helpers.flatten(jax.tree.map(to_np, g))
    return out

  mg = _load('make_golden')
  written = []
  for case in mg.CASES:
    name, kname, mlp, mname, n, d, nq, seed = case
    fx = np.load(os.path.join(HERE, name + '.npz'))
    rng = np.random.Generator(np.random.PCG64(seed))
    model = helpers.unflatten_like(helpers.make_model(rng, mname, mlp, d), fx['model_flat'])
    out = run_case(model, {'mlp_features': helpers.MLP_FEATURES}, kname + ('_mlp' if mlp else ''), mname,
                   fx['x'], fx['y'], fx['x2'], fx['y2'], fx['xq'], fx['y_aligned'])
    np.savez_compressed(os.path.join(HERE, name + '_ref.npz'), **out)
    written.append(name)
  mc = _load('make_golden_configs')
  x, y, raw = mc.cfg1_inputs()
  out = run_case(raw, {}, 'squared_exponential', 'constant', x, y, None, None, x[:8], None)
  np.savez_compressed(os.path.join(HERE, 'cfg1_se_n256_d4_ref.npz'), **out)
  written.append('cfg1_se_n256_d4')
  # cfg 4: the 64 ragged sub-datasets of bench.cfg4_inputs(): per-task NLL, their mean and the mean gradient through the
  # reference's own multi-task objective (objectives.py:178-195) -- the same keys as cfg4_t64_oracle.npz
  import bench
  data, raw4 = bench.cfg4_inputs()
  ds4 = {k: defs.SubDataset(jnp.asarray(xx), jnp.asarray(yy)) for k, (xx, yy) in data.items()}
  jm4 = to_jnp(raw4)
  nll4, key2nll4 = obj.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, defs.GPParams(model=jm4), ds4, wf,
                                                   return_key2nll=True)
  _, g4 = jax.value_and_grad(lambda m: obj.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential,
                                                                       defs.GPParams(model=m), ds4, wf))(jm4)
  np.savez_compressed(os.path.join(HERE, 'cfg4_t64_oracle_ref.npz'), sizes=np.asarray([data[k][0].shape[0] for k in sorted(data)]),
                      nll_per_task=np.asarray([float(key2nll4[k]) for k in sorted(data)]), nll_mean=float(nll4),
                      grad_mean_flat=helpers.flatten(jax.tree.map(to_np, g4)))
  written.append('cfg4_t64_oracle')
  # ---- row f1: the training driver's trajectory (every callback of gp.infer_parameters) ----
  for kname, mname in TRAIN_CASES:
    for method, steps, lr in TRAIN_METHODS:
      model, data = train_inputs(kname, mname)
      cfg = {'method': method, 'batch_size': TRAIN_BATCH, 'max_training_step': steps, 'learning_rate': lr,
             'mlp_features': helpers.MLP_FEATURES}
      ds = {i: defs.SubDataset(jnp.asarray(xx), jnp.asarray(yy)) for i, (xx, yy) in enumerate(data)}
      log = []
      def cb(*a, **k):   # Adam: callback(i, params.model, loss); L-BFGS: callback(step, params, loss) -- positional or by name
        step = k.get('step', a[0] if a else None)
        prm = k.get('params', k.get('model_params', a[1] if len(a) > 1 else None))
        loss = k.get('loss', a[2] if len(a) > 2 else None)
        log.append((int(step), helpers.flatten(jax.tree.map(to_np, prm)), float(loss)))
      out = gp.infer_parameters(getattr(mean, mname), getattr(kernel, kname), defs.GPParams(model=to_jnp(model), config=dict(cfg)), ds,
                                warp_func=wf, objective=obj.neg_log_marginal_likelihood, key=jax.random.PRNGKey(12), callback=cb)
      name = f'train_{kname}_{mname}_{method}'
      arrays = {f'x{i}': xx for i, (xx, yy) in enumerate(data)}
      arrays.update({f'y{i}': yy for i, (xx, yy) in enumerate(data)})
      np.savez_compressed(os.path.join(HERE, name + '_ref.npz'), kname=kname, mname=mname, method=method, steps=steps,
                          learning_rate=np.nan if lr is None else lr, batch_size=TRAIN_BATCH, model_flat=helpers.flatten(model),
                          cb_steps=np.asarray([r[0] for r in log]), cb_params=np.asarray([r[1] for r in log]),
                          cb_losses=np.asarray([r[2] for r in log]), final_flat=helpers.flatten(jax.tree.map(to_np, out.model)), **arrays)
      written.append(name)
  # ---- row a15: acquisition on an HGP = mean over the parameter samples ----
  samples, hx, hy, hx2, hy2, hxq = hgp_inputs()
  hds = {'test': defs.SubDataset(jnp.asarray(hx), jnp.asarray(hy)), 'other': defs.SubDataset(jnp.asarray(hx2), jnp.asarray(hy2)),
         'third': defs.SubDataset(jnp.asarray(hx2[:5]), jnp.asarray(hy2[:5]))}
  jsamples = [to_jnp(smp) for smp in samples]
  hgp = gp.HGP(dataset=hds, mean_func=mean.constant, cov_func=kernel.matern52,
               params=defs.GPParams(model=jsamples[0], samples=jsamples, config={}), warp_func=wf)
  hout = {'x': hx, 'y': hy, 'x2': hx2, 'y2': hy2, 'xq': hxq, 'samples_flat': np.asarray([helpers.flatten(smp) for smp in samples])}
  for name_, key_ in (('expected_improvement', 'ei'), ('probability_of_improvement', 'pi'), ('ucb', 'ucb')):
    hout[key_] = to_np(getattr(acfun, name_)(model=hgp, sub_dataset_key='test', x_queries=jnp.asarray(hxq)))
  np.savez_compressed(os.path.join(HERE, 'hgp_matern52_constant_ref.npz'), **hout)
  written.append('hgp_matern52_constant')
  print('make_golden_from_reference: wrote *_ref.npz for', ', '.join(written), f'(jax {jax.__version__})')
  return 0


if __name__ == '__main__':
  sys.exit(main(*sys.argv[1:2]))
