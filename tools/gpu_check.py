"""Ad-hoc GPU bring-up script (prints errors instead of asserting); the formal tests live in tests/."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as spla
from oracle import hyperbo_oracle as o
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import linalg, definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, gp, utils
from hyperbo_amd.bo_utils import acfun

rng = np.random.default_rng(0)
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

print('== spd_solve')
for dt in (np.float64, np.float32):
    for n in (5, 128, 200, 300, 1000):
        m_ = rng.normal(size=(n, n))
        a = (m_ @ m_.T / n + np.eye(n)).astype(dt)
        b = rng.normal(size=(n, 3)).astype(dt)
        chol, x = linalg.solve_linear_system(a, b)
        inv, ldh = linalg.spd_inverse(a)
        cref = spla.cholesky(a.astype(np.float64), lower=True)
        xref = spla.cho_solve((cref, True), b.astype(np.float64))
        iref = np.linalg.inv(a.astype(np.float64))
        print(dt.__name__, n, 'chol', rel(chol, cref), 'solve', rel(x, xref), 'inv', rel(inv, iref), 'logdet', abs(ldh - np.sum(np.log(np.diag(cref)))))
# non PD
a = -np.eye(6); chol, x = linalg.solve_linear_system(a, np.ones((6, 1))); print('nonPD nan:', np.isnan(chol).all(), np.isnan(x).all())

def mk_params(kname, mname, mlp, d, dt=np.float64, F=None):
    F = F or (5 if mlp else d)
    model = {'lengthscale': (rng.normal(size=F) * 0.3 + 0.5), 'signal_variance': np.array(0.3), 'noise_variance': np.array(-2.0),
             'constant': np.array(0.4), 'dot_prod_sigma': np.array(0.7), 'dot_prod_bias': np.array(0.2)}
    if mlp or mname == 'linear_mlp':
        model['mlp_params'] = {'Dense_0': {'kernel': rng.normal(size=(d, 4)) * 0.7, 'bias': rng.normal(size=4) * 0.1},
                               'Dense_1': {'kernel': rng.normal(size=(4, 5)) * 0.7, 'bias': rng.normal(size=5) * 0.1}}
    if mname in ('linear', 'linear_mlp'):
        fin = 5 if mname == 'linear_mlp' else d
        model['linear_mean'] = {'kernel': rng.normal(size=(fin, 1)), 'bias': rng.normal(size=1)}
    return model

print('== gram / mean')
d = 3
for kname in ['squared_exponential', 'matern32', 'matern52', 'dot_product']:
    for mlp in (False, True):
        model = mk_params(kname, 'linear_mlp', mlp, d)
        po = o.GPParams(model=model, config={'mlp_features': (4, 5)}); pn = defs.GPParams(model=model, config={'mlp_features': (4, 5)})
        ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
        x1 = rng.uniform(size=(150, d)); x2 = rng.uniform(size=(37, d))
        e1 = rel(kn(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC), ko(po, x1, warp_func=o.DEFAULT_WARP_FUNC))
        e2 = rel(kn(pn, x1, x2, warp_func=utils.DEFAULT_WARP_FUNC), ko(po, x1, x2, warp_func=o.DEFAULT_WARP_FUNC))
        e3 = rel(kn(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC, diag=True), ko(po, x1, warp_func=o.DEFAULT_WARP_FUNC, diag=True))
        print(kname, mlp, 'gram', e1, 'cross', e2, 'diag', e3)
for mname in ['zero', 'constant', 'linear', 'linear_mlp']:
    model = mk_params('squared_exponential', mname, False, d)
    po = o.GPParams(model=model); pn = defs.GPParams(model=model)
    x1 = rng.uniform(size=(50, d))
    print('mean', mname, rel(getattr(mean, mname)(pn, x1, warp_func=utils.DEFAULT_WARP_FUNC), getattr(o, mname)(po, x1, warp_func=o.DEFAULT_WARP_FUNC)) if mname != 'zero' else np.abs(getattr(mean, mname)(pn, x1)).max())

def flat(tree, out=None):
    out = [] if out is None else out
    if isinstance(tree, dict):
        for k in sorted(tree): flat(tree[k], out)
    else: out.append(np.asarray(tree, dtype=np.float64).ravel())
    return out

print('== nll / grad')
for kname in ['squared_exponential', 'matern32', 'matern52', 'dot_product']:
    for mname in ['zero', 'constant', 'linear']:
        model = mk_params(kname, mname, False, d)
        po = o.GPParams(model=model); pn = defs.GPParams(model=model)
        dso = {0: o.SubDataset(rng.uniform(size=(140, d)), rng.normal(size=(140, 1))), 1: o.SubDataset(rng.uniform(size=(300, d)), rng.normal(size=(300, 1))),
               2: o.SubDataset(rng.uniform(size=(20, d)), rng.normal(size=(20, 3)), aligned=1), 3: o.SubDataset(np.zeros((0, d)), np.zeros((0, 1)))}
        dsn = {k: defs.SubDataset(v.x, v.y, v.aligned) for k, v in dso.items()}
        for ex in (True, False):
            vo, go = o.nll_value_and_grad(getattr(o, mname), getattr(o, kname), po, dso, o.DEFAULT_WARP_FUNC, exclude_aligned=ex)
            vn, gn = objectives.nll_value_and_grad(getattr(mean, mname), getattr(kernel, kname), pn, dsn, utils.DEFAULT_WARP_FUNC, exclude_aligned=ex)
            vn2, k2 = objectives.neg_log_marginal_likelihood(getattr(mean, mname), getattr(kernel, kname), pn, dsn, utils.DEFAULT_WARP_FUNC, exclude_aligned=ex, return_key2nll=True)
            _, k2o = o.neg_log_marginal_likelihood(getattr(o, mname), getattr(o, kname), po, dso, o.DEFAULT_WARP_FUNC, exclude_aligned=ex, return_key2nll=True)
            keys = [k for k in model if k in go and not isinstance(go[k], dict) or k == 'linear_mean' and mname == 'linear']
            gerr = {k: rel(np.concatenate(flat(gn[k])), np.concatenate(flat(go[k]))) for k in keys if np.abs(np.concatenate(flat(go[k]))).max() > 0}
            print(kname, mname, ex, 'val', abs(vn - vo) / abs(vo), abs(vn2 - vo) / abs(vo), 'k2', max(abs(k2[k] - k2o[k]) / abs(k2o[k]) for k in k2o), 'grad', {k: f'{v:.1e}' for k, v in gerr.items()})

print('== predict / acq')
for kname, mlp, mname in [('squared_exponential', False, 'constant'), ('matern52', True, 'linear_mlp'), ('matern32', False, 'linear'), ('dot_product', True, 'zero')]:
    model = mk_params(kname, mname, mlp, d)
    cfg = {'mlp_features': (4, 5)}
    po = o.GPParams(model=model, config=cfg); pn = defs.GPParams(model=model, config=dict(cfg))
    ko = getattr(o, kname + ('_mlp' if mlp else '')); kn = getattr(kernel, kname + ('_mlp' if mlp else ''))
    mo = getattr(o, mname); mn = getattr(mean, mname)
    x = rng.uniform(size=(200, d)); y = rng.normal(size=(200, 1)); xq = rng.uniform(size=(70, d))
    cho, kio, ymo = o.solve_gp_linear_system(mo, ko, po, x, y, o.DEFAULT_WARP_FUNC)
    chn, kin, ymn = linalg.solve_gp_linear_system(mn, kn, pn, x, y, utils.DEFAULT_WARP_FUNC)
    muo, varo = o.predict(mo, ko, po, x, y, xq, o.DEFAULT_WARP_FUNC)
    mun, varn = gp.predict(mn, kn, pn, x, y, xq, utils.DEFAULT_WARP_FUNC)
    muo2, covo = o.predict(mo, ko, po, x, y, xq, o.DEFAULT_WARP_FUNC, full_cov=True)
    mun2, covn = gp.predict(mn, kn, pn, x, y, xq, utils.DEFAULT_WARP_FUNC, full_cov=True)
    print(kname, mlp, mname, 'chol', rel(chn, cho), 'kinvy', rel(kin, kio), 'ymu', rel(ymn, ymo), 'mu', rel(mun, muo), 'var', rel(varn, varo), 'cov', rel(covn, covo))
    ds = {0: defs.SubDataset(x, y), 1: defs.SubDataset(x[:50], y[:50])}
    model_n = gp.GP(ds, mn, kn, pn, utils.DEFAULT_WARP_FUNC)
    mu_g, var_g = model_n.predict(xq, 0)
    mu_o, var_o = o.gp_predict_postprocess(po, {0: 1, 1: 2} and {k: o.SubDataset(v.x, v.y) for k, v in ds.items()}, muo, varo, o.DEFAULT_WARP_FUNC, False, True, True)
    print('   GP.predict var', rel(var_g, var_o))
    for name, sub_o, cb in [('expected_improvement', o.expected_improvement_sub, np.max(y)), ('probability_of_improvement', o.probability_of_improvement_sub, np.max(y) + 0.1), ('ucb', o.ucb_sub, 3.0)]:
        an = getattr(acfun, name)(model=model_n, sub_dataset_key=0, x_queries=xq)
        ao = sub_o(mu_o, np.sqrt(var_o), cb)
        print('  ', name, rel(an, ao))
    # prior path
    mu_p, var_p = model_n.predict(xq, 'missing')
    mu_po, var_po = o.predict(mo, ko, po, None, None, xq, o.DEFAULT_WARP_FUNC)
    print('   prior mu', rel(mu_p, mu_po) if np.abs(mu_po).max() > 0 else np.abs(mu_p).max())

print('== timing cfg2-like')
for N, D in ((2048, 16), (8192, 16)):
    X = rng.uniform(size=(N, D)); w = rng.normal(size=D); y = np.sin(2 * np.pi * X @ w)[:, None] + 0.1 * rng.normal(size=(N, 1))
    inv_sp = lambda v: np.log(np.expm1(v))
    model = {'lengthscale': inv_sp(np.full(D, np.sqrt(D) * 0.3)), 'signal_variance': inv_sp(np.array(1.0)), 'noise_variance': inv_sp(np.array(1e-2)), 'constant': np.array(0.0)}
    pn = defs.GPParams(model=model)
    dev = objectives.DeviceDataset({0: defs.SubDataset(X, y)})
    ctx = nat.default_context()
    for prof in (0, 2):
        ctx.profile_enable(prof)
        for it in range(3):
            t0 = time.time(); v, g = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC); t1 = time.time()
            t2 = time.time(); v2 = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC); t3 = time.time()
        print(N, 'nll+grad %.2f ms' % ((t1 - t0) * 1e3), 'nll only %.2f ms' % ((t3 - t2) * 1e3), v, v2)
        if prof:
            objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, pn, dev, utils.DEFAULT_WARP_FUNC)
            for k, (ms, cnt) in ctx.profile_get().items(): print('    %-16s %9.3f ms  %4d launches' % (k, ms, cnt))
    if N <= 2048:
        po = o.GPParams(model=model)
        vo, go = o.nll_value_and_grad(o.constant, o.squared_exponential, po, {0: o.SubDataset(X, y)}, o.DEFAULT_WARP_FUNC)
        print('   vs oracle val', abs(v - vo) / abs(vo), 'grad', {k: f'{rel(g[k], go[k]):.1e}' for k in g})
print('DONE')
