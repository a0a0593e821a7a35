"""Grid scan of context options on the headline evaluation: scan_opt.py N name=a,b,c name2=x,y ...  (cartesian product;
3 rounds interleaved so that drift shows; prints the median per setting, best first)"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1])
axes = [(a.split('=')[0], [int(v) for v in a.split('=')[1].split(',')]) for a in sys.argv[2:]]
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
ctx.profile_enable(0)
res = {}
for rnd in range(3):
    for combo in itertools.product(*[v for _, v in axes]):
        for (name, _), v in zip(axes, combo):
            ctx.set_option(name, v)
        f(); f()
        t0 = time.perf_counter()
        for _ in range(8): f()
        res.setdefault(combo, []).append((time.perf_counter() - t0) / 8 * 1e3)
names = [a for a, _ in axes]
for combo, ts in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    print(' '.join(f'{k}={v}' for k, v in zip(names, combo)), ' median %.3f ms  (%s)' % (np.median(ts), ' '.join('%.2f' % t for t in ts)))
