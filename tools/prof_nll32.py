"""fp32 NLL+grad of the cfg-2 shape (the reference's default dtype): stage times.  prof_nll32.py [N] [opt=v ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
x, y, raw = bench.cfg2_inputs(n=n)
to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
ctx = nat.default_context()
for opt in sys.argv[1:]:
    if '=' in opt: ctx.set_option(opt.split('=')[0], int(opt.split('=')[1]))
for dt, xx, yy, rr in (('f32', x.astype(np.float32), y.astype(np.float32), to32(raw)), ('f64', x, y, raw)):
    dev = objectives.DeviceDataset({0: defs.SubDataset(xx, yy)})
    p = defs.GPParams(model=rr)
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    for lvl in (0, 1):
        ctx.profile_enable(lvl)
        v, g = f(); f()
        t0 = time.perf_counter()
        for _ in range(5): f()
        t1 = time.perf_counter()
        print(dt, 'N', n, f'level {lvl}: {2e2*(t1-t0):.2f} ms  nll {v:.6f}')
    print('  ', {k: round(ms, 2) for k, (ms, cnt) in ctx.profile_get().items()})
    dev.close()
