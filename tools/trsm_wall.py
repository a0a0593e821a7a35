"""Per-workgroup in-kernel wall time of the trsm of the panels around the early-inverse kick-off (library built
with -DHBO_POTF2_TIMING): separates waiting for a workgroup slot from slow execution."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
x, y, raw = bench.cfg2_inputs(n=8192)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f()
lib = nat.lib()
lib.hbo_dbg_trsm_wall.argtypes = [C.c_void_p, C.c_int]
for panel in [int(a) for a in sys.argv[1:]] or [34, 35, 36, 37, 38, 50]:
    lib.hbo_dbg_trsm_wall(None, panel)
    f()
    buf = (C.c_ulonglong * (3 * 128))()
    lib.hbo_dbg_trsm_wall(buf, 0)
    nwg = (8192 + 128 - (panel + 1) * 128) // 64
    a = np.array(buf[:3 * nwg], dtype=np.uint64).reshape(nwg, 3)
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0).astype(np.float64) / 100; dur = (a[:, 1] - a[:, 0]).astype(np.float64) / 100
    cus = {(int(h >> 32) & 0xf, (int(h) >> 13) & 0x7, (int(h) >> 8) & 0xf) for h in a[:, 2]}
    print('panel %d: %d wgs on %d distinct (xcc,se,cu); start spread %.1f us (median %.1f); in-kernel us min %.1f median %.1f max %.1f; span %.1f us'
          % (panel, nwg, len(cus), start.max(), np.median(start), dur.min(), np.median(dur), dur.max(), (a[:, 1].max() - t0) / 100.0))
