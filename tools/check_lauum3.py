"""fp32 NLL+grad with K^-1 = W^T W on the bf16 cores (lauum_bf16x3 = 1) against the fp32-MFMA form and fp64: error and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
from tests.helpers import flatten
ctx = nat.default_context()
to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
for n in [int(a) for a in sys.argv[1:]] or [2000, 4096, 8192, 16384]:
    x, y, raw = bench.cfg2_inputs(n=n)
    d64 = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    v64, g64 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=raw), d64, utils.DEFAULT_WARP_FUNC)
    g64 = flatten(g64); d64.close()
    d32 = objectives.DeviceDataset({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))})
    p32 = defs.GPParams(model=to32(raw))
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p32, d32, utils.DEFAULT_WARP_FUNC)
    for flag in (0, 1):
        ctx.set_option('lauum_bf16x3', flag)
        v, g = f(); f()
        t0 = time.perf_counter()
        for _ in range(5): f()
        ms = 2e2 * (time.perf_counter() - t0)
        g = flatten(g)
        print('N = %5d lauum_bf16x3 = %d: %.2f ms   nll rel err %.2e   grad err / max|g| %.2e' % (n, flag, ms, abs(v - v64) / abs(v64), np.abs(g - g64).max() / np.abs(g64).max()), flush=True)
    d32.close()
