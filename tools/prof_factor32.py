"""fp32 factorisation of the cfg-3 shape (N = 16384 by default): stage and per-launch times of hbo_factor for a list of option
settings.  usage: prof_factor32.py [N] ["opt=v opt2=w" ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import gp, kernel, mean, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
settings = [a for a in sys.argv[1:] if '=' in a] or ['']
plain = 'plain' in sys.argv[1:]   # no stage events: three factorisations per setting, for a rocprofv3 kernel trace
rng = np.random.Generator(np.random.PCG64(3))
d, f = 32, 64
isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64)))
model = {'lengthscale': isp(np.ones(f)), 'signal_variance': isp(1.0), 'noise_variance': isp(1e-2),
         'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
         'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
to = lambda t: {k: to(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
x = rng.uniform(size=(n, d)).astype(np.float32); y = (np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))).astype(np.float32)
g = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp, defs.GPParams(model=to(model), config={'mlp_features': (f,)}), utils.DEFAULT_WARP_FUNC)
ctx = nat.default_context()
for st in settings:
    for kv in st.split():
        ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    if plain:
        for it in range(3):
            g.update_model_params(g.params.model); g.setup_predictor(0)
        continue
    out = {}
    for lvl in (1, 2):
        ctx.profile_enable(lvl)
        best = None
        for it in range(3):
            g.update_model_params(g.params.model)
            t0 = time.perf_counter(); g.setup_predictor(0); el = time.perf_counter() - t0
            pf = ctx.profile_get()
            if best is None or el < best[0]: best = (el, pf)
        out[lvl] = best
    print('== %s: wall %.1f ms' % (st or 'defaults', out[1][0] * 1e3), {k: round(v[0], 2) for k, v in out[1][1].items()})
    print('   per launch (level 2):', {k: '%.2f ms / %d = %.1f us' % (v[0], v[1], 1e3 * v[0] / max(v[1], 1)) for k, v in out[2][1].items()
                                      if k in ('potf2', 'trsm', 'split3', 'syrk_col', 'syrk_trailing', 'syrk_bulk', 'trtri_gemm', 'trtri_diag')})
