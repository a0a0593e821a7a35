"""In-kernel timeline of every tagged launch of ONE evaluation (library built with EXTRA=-DHBO_TIMELINE):
  timeline.py [nll8192 | shard8 | T64 | nll4096 ...] [value] [opt=v ...] [rows=lo:hi]
Per launch: first workgroup start / last workgroup end (100 MHz wall clock, us relative to the first launch).  Prints the chain's
launches panel by panel (start, duration, gap to the previous chain launch) and a per-kind summary by phase of the factorisation."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

args = sys.argv[1:]
leg = 'nll8192'
value_only = False
rows = None
opts = []
for a in args:
    if a == 'value': value_only = True
    elif a.startswith('rows='): rows = tuple(int(v) for v in a[5:].split(':'))
    elif '=' in a: opts.append(a)
    else: leg = a
if leg.startswith('nll'):
    x, y, raw = bench.cfg2_inputs(n=int(leg[3:]))
    dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
else:
    data, raw = bench.cfg4_inputs()
    full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
    dev = objectives.DeviceDataset(full if leg == 'T64' else parallel.shard_dataset(full, 0, int(leg[5:])))
ctx = nat.default_context()
for o in opts:
    k, v = o.split('='); ctx.set_option(k, int(v))
p = defs.GPParams(model=raw)
if value_only:
    f = lambda: objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
else:
    f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
f(); f(); f()
lib = nat.lib()
lib.hbo_dbg_timeline.restype = C.c_int
lib.hbo_dbg_timeline(1, None, None, None)
f()
MAXN = 16384
times = (C.c_ulonglong * (2 * MAXN))()
names = C.create_string_buffer(16 * MAXN)
ps = (C.c_int * MAXN)()
n = lib.hbo_dbg_timeline(0, times, names, ps)
t = np.array(times[:2 * n], dtype=np.uint64).reshape(n, 2)
nm = [names.raw[16 * i:16 * i + 16].split(b'\0')[0].decode() for i in range(n)]
pp = np.array(ps[:n])
ok = t[:, 1] > 0
t0 = t[ok, 0].min()
st = (t[:, 0].astype(np.int64) - np.int64(t0)) / 100.0
en = (t[:, 1].astype(np.int64) - np.int64(t0)) / 100.0
print('%d tagged launches, span %.1f us' % (n, en[ok].max()))
chain = ('syrk_col', 'syrk_inner', 'potf2', 'trsm', 'f1')
prev_end = None
print('--- chain launches (start, dur, gap after the previous chain launch) ---')
for i in range(n):
    if not ok[i] or nm[i] not in chain: continue
    gap = st[i] - prev_end if prev_end is not None else 0.0
    prev_end = en[i]
    if rows is None or rows[0] <= pp[i] < rows[1]:
        print('%-10s p=%3d  start %9.1f  dur %7.1f  gap %6.1f' % (nm[i], pp[i], st[i], en[i] - st[i], gap))
print('--- other launches ---')
for i in range(n):
    if not ok[i] or nm[i] in chain: continue
    if rows is None or nm[i] in ('f2', 'lauum'):
        print('%-10s p=%3d  start %9.1f  dur %7.1f  end %9.1f' % (nm[i], pp[i], st[i], en[i] - st[i], en[i]))
# per-kind summary by halves of the factorisation
pmax = max(pp[i] for i in range(n) if nm[i] == 'potf2') + 1
for lo, hi in ((0, pmax // 2), (pmax // 2, pmax)):
    line = 'panels %3d-%3d:' % (lo, hi - 1)
    for k in chain:
        d = [en[i] - st[i] for i in range(n) if ok[i] and nm[i] == k and lo <= pp[i] < hi]
        if d: line += '  %s n=%d mean %.1f' % (k, len(d), np.mean(d))
    pst = [st[i] for i in range(n) if ok[i] and nm[i] == 'potf2' and lo <= pp[i] < hi]
    if len(pst) > 1: line += '  | period %.1f' % ((pst[-1] - pst[0]) / (len(pst) - 1))
    print(line)
