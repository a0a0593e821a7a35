"""Per-kernel-class HIP-event breakdown of one cfg2 NLL+grad evaluation (profile level 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
for opt in sys.argv[2:]:
    k, v = opt.split('='); ctx.set_option(k, int(v))
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
LEVELS = (0, int(os.environ.get("HBO_PROF_LEVEL", "2")))
for lvl in LEVELS:
    ctx.profile_enable(lvl)
    f(); f()
    t0 = time.perf_counter(); f(); t1 = time.perf_counter()
    print(f'level {lvl}: {1e3*(t1-t0):.2f} ms')
for k, (ms, cnt) in ctx.profile_get().items():
    print(f'   {k:16s} {ms:8.3f} ms {cnt:5d} launches  {1e3*ms/cnt:8.1f} us avg')
