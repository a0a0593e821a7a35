"""Workgroup-slot occupancy over time of one GEMM launch (lauum = mode 3, trtri A/B = 1/2) from in-kernel wall-clock
stamps (library built with `make EXTRA=-DHBO_GEMM_TIMING`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
index = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # which launch of that mode within the evaluation (mode + 100: persistent ones)
if len(sys.argv) > 3 and sys.argv[3] == 'cfg4':
    data, raw = bench.cfg4_inputs(tasks=64)
    dev = objectives.DeviceDataset({k: defs.SubDataset(x, y) for k, (x, y) in data.items()})
else:
    x, y, raw = bench.cfg2_inputs(n=8192)
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
for opt in sys.argv[4:]:
    k, v = opt.split('='); nat.default_context().set_option(k, int(v))
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f()
lib = nat.lib()
lib.hbo_dbg_gemm_wall.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.hbo_dbg_gemm_wall(None, mode, index)
f()
buf = (C.c_ulonglong * (4 * 8192))()
lib.hbo_dbg_gemm_wall(buf, 0, 0)
a = np.array(buf[:], dtype=np.uint64).reshape(8192, 4)
a = a[a[:, 1] > 0]                       # workgroups that ran a tile
t0 = a[:, 0].min()
st = (a[:, 0] - t0).astype(np.float64) / 100; en = (a[:, 1] - t0).astype(np.float64) / 100
ks = a[:, 3].astype(np.float64)
span = en.max()
print('%d tiles, span %.1f us, sum tile time %.1f ms, mean slots busy %.1f' % (len(a), span, (en - st).sum() / 1e3, (en - st).sum() / span))
print('us per k-step: median %.3f  p10 %.3f  p90 %.3f' % tuple(np.percentile((en - st) / ks, [50, 10, 90])))
edges = np.linspace(0, span, 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    busy = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    print('  %7.0f-%7.0f us: %5.0f slots busy' % (lo, hi, busy))
# the 12 last-finishing tiles
order = np.argsort(-en)[:12]
print('last finishers: (start, end, ksteps) ', [(round(st[i]), round(en[i]), int(ks[i])) for i in order])
