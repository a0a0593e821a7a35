"""HIP-event breakdown of one cfg4 (64 ragged tasks) NLL+grad evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
# T = number of tasks of cfg 4 (first T of the generator), or "shard8" = the heaviest LPT shard of eight of the 64 tasks
# (what rank 0 of an 8-GPU job holds: bench.py's multitask.shard_of_8)
if len(sys.argv) > 1 and sys.argv[1] == 'shard8':
    from hyperbo_amd import parallel
    data, raw = bench.cfg4_inputs()
    full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
    mine = parallel.shard_dataset(full, 0, 8)
    data = {k: (s.x, s.y) for k, s in mine.items()}
    T = len(data)
else:
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    data, raw = bench.cfg4_inputs(tasks=T)
dev = objectives.DeviceDataset({k: defs.SubDataset(x, y) for k, (x, y) in data.items()})
ctx = nat.default_context()
for opt in sys.argv[2:]:
    k, v = opt.split('='); ctx.set_option(k, int(v))
p = defs.GPParams(model=raw)
f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
for lvl in ((0,) if os.environ.get("HBO_PROF_LEVEL") == "0" else (0, 2)):
    ctx.profile_enable(lvl)
    f(); f()
    t0 = time.perf_counter(); f(); t1 = time.perf_counter()
    print(f'level {lvl}: {1e3*(t1-t0):.2f} ms  ({T} tasks, sum n^3 = {sum(float(x.shape[0])**3 for x, _ in data.values()):.3e})')
for k, (ms, cnt) in ctx.profile_get().items():
    print(f'   {k:16s} {ms:8.3f} ms {cnt:5d} launches  {1e3*ms/cnt:8.1f} us avg')
