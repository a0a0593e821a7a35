"""fp32 legs (the reference's default dtype) for A/B of two builds: NLL + gradient of one matrix at N = 1024 / 2048 / 4096 / 8192 and of the
cfg-4 batch cast to fp32 -- ms per evaluation (median of 3 rounds) and the value, to compare builds to the bit.  HBO_LIB=<other build>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
ctx = nat.default_context(); ctx.profile_enable(0)
def leg(name, ds, raw, reps):
    dev = objectives.DeviceDataset(ds)
    p = defs.GPParams(model=to32(raw))
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    v, g = f(); f()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    gs = float(np.sum([np.sum(np.asarray(t, dtype=np.float64)) for t in g.values()])) if isinstance(g, dict) else 0.0
    print('%-10s %8.3f ms   nll %.9g  sum(grad) %.9g' % (name, np.median(ts), v, gs), flush=True)
    dev.close()
for n in (1024, 2048, 4096, 8192):
    x, y, raw = bench.cfg2_inputs(n=n)
    leg('nll%d' % n, {0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))}, raw, 10 if n < 8192 else 5)
data, raw = bench.cfg4_inputs()
leg('T64', {k: defs.SubDataset(x.astype(np.float32), y.astype(np.float32)) for k, (x, y) in data.items()}, raw, 5)
