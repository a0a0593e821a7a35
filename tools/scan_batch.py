import os, sys, time, itertools
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
T = int(sys.argv[1])
axes = [(a.split('=')[0], [int(v) for v in a.split('=')[1].split(',')]) for a in sys.argv[2:]]
data, raw = bench.cfg4_inputs()
# the heaviest LPT shard of eight, as bench.py's shard_of_8
keys = sorted(data, key=lambda k: -data[k][0].shape[0] ** 3)
if T == 8:
    loads = [0.0] * 8; shards = [[] for _ in range(8)]
    for k in keys:
        i = int(np.argmin(loads)); shards[i].append(k); loads[i] += float(data[k][0].shape[0]) ** 3
    sel = shards[int(np.argmax(loads))]
else:
    sel = list(data)[:T]
dev = objectives.DeviceDataset({k: defs.SubDataset(*data[k]) for k in sel})
ctx = nat.default_context()
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
res = {}
for rnd in range(3):
    for combo in itertools.product(*[v for _, v in axes]):
        for (name, _), v in zip(axes, combo): ctx.set_option(name, v)
        f(); f()
        t0 = time.perf_counter()
        for _ in range(10): f()
        res.setdefault(combo, []).append((time.perf_counter() - t0) / 10 * 1e3)
names = [a for a, _ in axes]
for combo, ts in sorted(res.items(), key=lambda kv: np.median(kv[1])):
    print(f'T={len(sel)}', ' '.join(f'{k}={v}' for k, v in zip(names, combo)), ' median %.3f ms  (%s)' % (np.median(ts), ' '.join('%.2f' % t for t in ts)))
