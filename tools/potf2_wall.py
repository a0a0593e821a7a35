"""In-kernel wall-clock duration of every potf2 of one cfg2 evaluation (library built with -DHBO_POTF2_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
args = [a for a in sys.argv[1:] if a != 'cfg4']
if 'cfg4' in sys.argv[1:]:
    data, raw = bench.cfg4_inputs(tasks=64)
    dev = objectives.DeviceDataset({k: defs.SubDataset(x, y) for k, (x, y) in data.items()})
    NP = 16
else:
    x, y, raw = bench.cfg2_inputs(n=8192)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
    NP = 64
ctx = nat.default_context()
for opt in args:
    k, v = opt.split('='); ctx.set_option(k, int(v))
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f(); f()
buf = (C.c_ulonglong * (3 * 256))()
nat.lib().hbo_dbg_potf2_wall(buf)
a = np.array(buf[:3 * NP], dtype=np.uint64).reshape(NP, 3)
dur = (a[:, 1] - a[:, 0]).astype(np.float64) / 100.0   # us
gap = (a[1:, 0] - a[:-1, 0]).astype(np.float64) / 100.0
hw = a[:, 2]
print('potf2 in-kernel us: mean %.1f  first-half %.1f  second-half %.1f' % (dur.mean(), dur[:NP // 2].mean(), dur[NP // 2:].mean()))
print('panel period us   : mean %.1f  first-half %.1f  second-half %.1f' % (gap.mean(), gap[:NP // 2].mean(), gap[NP // 2:].mean()))
for i in range(0, NP, 4):
    print(' panels %2d-%2d dur %s  period %s  xcc/cu %s' % (i, i + 3, np.round(dur[i:i + 4], 1), np.round(gap[i:i + 4] if i + 4 < NP else gap[i:], 1),
          [(int(h >> 32) & 0xf, (int(h) >> 8) & 0xf, (int(h) >> 13) & 0x7) for h in hw[i:i + 4]]))
