"""A/B of context options over the legs a schedule change can move, with a value check against the first setting:
  ab_suite.py [legs=nll2048,nll4096,nll8192,T64,shard8] opt=a,opt2=b  opt=c ...      (each further argument is one setting; '-' = defaults)
Per leg and setting: median ms per NLL+grad evaluation (3 rounds, interleaved), NLL difference and max gradient difference
(relative to max|g|) against the first setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

args = sys.argv[1:]
legs = 'nll2048,nll4096,nll8192,T64,shard8'
if args and args[0].startswith('legs='):
    legs = args.pop(0)[5:]
settings = args or ['-']
ctx = nat.default_context()
ctx.profile_enable(0)
touched = {}


def apply(setting):
    for k, v in touched.items():
        ctx.set_option(k, v)           # back to the recorded defaults
    if setting == '-':
        return
    for opt in setting.split(','):
        k, v = opt.split('=')
        touched.setdefault(k, DEFAULTS.get(k, 0))
        ctx.set_option(k, int(v))


DEFAULTS = {'lookahead': 1, 'overlap_trtri': 1, 'potrf_group': 0, 'group_inner': -1, 'persist_free': -1, 'small_nblk': -1, 'cu_yield': 2, 'trtri_free': 48,
            'trtri_at': 0, 'sweep': 1, 'sweep_qs': 0, 'batch_bg': -1, 'sweep_big': 4000, 'split_f1': 1, 'sweep_side': 1, 'lauum_persist': 1, 'sweep_free': 24}


def flat(g):
    out = []
    def rec(t):
        if isinstance(t, dict):
            for k in sorted(t): rec(t[k])
        else:
            out.append(np.ravel(np.asarray(t, dtype=np.float64)))
    rec(g)
    return np.concatenate(out)


def make(leg):
    if leg.endswith('v'):      # value only (the factorisation alone: Gram + potrf + reduce), e.g. shard8v, T64v, nll8192v
        g = make_vg(leg[:-1], True)
        return lambda: (g(), {'x': np.zeros(1)})
    return make_vg(leg, False)


def make_vg(leg, value_only):
    if leg.startswith('nll'):
        x, y, raw = bench.cfg2_inputs(n=int(leg[3:]))
        dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
        p = defs.GPParams(model=raw)
    else:
        data, raw = bench.cfg4_inputs()
        full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
        ds = full if leg == 'T64' else parallel.shard_dataset(full, 0, int(leg[5:]))
        dev = objectives.DeviceDataset(ds)
        p = defs.GPParams(model=raw)
    if value_only:
        return lambda: objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    return lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)


for leg in ([] if legs == 'none' else legs.split(',')):
    f = make(leg)
    res, vals = {}, {}
    reps = 5 if leg in ('T64', 'nll8192', 'T64v', 'nll8192v') else 10
    for rnd in range(3):
        for s in settings:
            apply(s)
            ctx.set_option('poison', 1)     # the value check sees skipped work as NaN, not as the previous evaluation's numbers
            v, g = f()
            ctx.set_option('poison', 0)
            f()
            vals[s] = (v, flat(g))
            t0 = time.perf_counter()
            for _ in range(reps): f()
            res.setdefault(s, []).append((time.perf_counter() - t0) / reps * 1e3)
    v0, g0 = vals[settings[0]]
    for s in settings:
        v, g = vals[s]
        print('%-8s %-44s %8.3f ms  (%s)  dNLL %.1e  dgrad %.1e' % (leg, s, np.median(res[s]), ' '.join('%.3f' % t for t in res[s]),
              abs(v - v0) / abs(v0), np.max(np.abs(g - g0)) / max(np.max(np.abs(g0)), 1e-300)), flush=True)
apply('-')
