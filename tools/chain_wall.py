"""Per-panel phase times inside the persistent chain kernel (library built with `make EXTRA=-DHBO_CHAIN_TIMING`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
x, y, raw = bench.cfg2_inputs(n=8192)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
ctx.set_option('chain', 1)
for opt in sys.argv[1:]:
    k, v = opt.split('='); ctx.set_option(k, int(v))
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f(); f()
buf = (C.c_ulonglong * (6 * 256))()
nat.lib().hbo_dbg_chain_stamps(buf)
a = np.array(buf[:6 * 64], dtype=np.uint64).reshape(64, 6).astype(np.float64) / 100.0
d = np.diff(a, axis=1)
print("panel: (unused) wait_col  potf2  trsm  barrier   | total")
for i in list(range(0, 64, 4)) + [1, 2, 3, 61, 62, 63]:
    print('%4d  %8.1f %10.1f %7.1f %6.1f %7.1f   | %7.1f' % (i, d[i, 0], d[i, 1], d[i, 2], d[i, 3], d[i, 4], a[i, 5] - a[i, 0]))
print('sums (ms): wait %.2f  update %.2f  potf2 %.2f  trsm %.2f  barrier %.2f  total %.2f' % tuple(list(d.sum(axis=0) / 1e3) + [(a[-1, 5] - a[0, 0]) / 1e3]))
