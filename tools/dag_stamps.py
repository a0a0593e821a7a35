"""Per-panel wall-clock stamps of the resident tile-task schedule (debug build: make -C hyperbo_amd/csrc EXTRA=-DHBO_DAG_DEBUG).
usage: dag_stamps.py N [opt=v ...]   -- columns in microseconds relative to the end of the previous panel solve"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1])
ctx = nat.default_context()
ctx.set_option('dag', 1)
for a in sys.argv[2:]:
    ctx.set_option(a.split('=')[0], int(a.split('=')[1]))
x, y, raw = bench.cfg2_inputs(n=n)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f(); f()
st = np.zeros(128 * 16, dtype=np.uint64)
nat.lib().hbo_dbg_dag_stamps(st.ctypes.data_as(C.c_void_p))
raw_st = st.reshape(128, 16).copy()
st = st.reshape(128, 16).astype(np.float64) / 100.0   # us
w = raw_st[120]
if w[4]:
    print('workers %d  tasks %d  per worker: scheduler %.1f us, tiles %.1f us, publish %.1f us;  per task: sched %.1f tile %.1f publish %.1f us' % (
        w[4], w[3], w[0] / 100 / w[4], w[1] / 100 / w[4], w[2] / 100 / w[4], w[0] / 100 / max(w[3], 1), w[1] / 100 / max(w[3], 1), w[2] / 100 / max(w[3], 1)))
nb = n // 128
print('panel | potf2: enter pass end | solve: first-pass last-pass first-done last-done | col tasks of p+1: draw start last-done   (us after the previous solve ended)')
for q in range(nb):
    ref = st[q - 1, 1] if q else st[0, 5]
    r = lambda k: st[q, k] - ref
    print('%4d | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f   step %.1f' % (
        q, r(5), r(6), r(7), r(8), r(9), r(0), r(1), r(2), r(3), r(4), st[q, 1] - ref))
print('total chain us', st[nb - 1, 1] - st[0, 5])
