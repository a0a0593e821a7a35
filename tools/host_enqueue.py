"""How much of an evaluation's wall time the host spends queueing it (stage "host_enqueue" of hbo_profile): if the host is not
well ahead of the device, launch latency is on the critical path and a captured graph would pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat, parallel
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
ctx = nat.default_context()
data, raw = bench.cfg4_inputs()
full = {k: defs.SubDataset(x, y) for k, (x, y) in data.items()}
legs = {'shard8': parallel.shard_dataset(full, 0, 8), 'T64': full}
x2, y2, raw2 = bench.cfg2_inputs()
for name, ds, rw in (('shard8', legs['shard8'], raw), ('T64', legs['T64'], raw), ('nll8192', {0: defs.SubDataset(x2, y2)}, raw2)):
  dev = objectives.DeviceDataset(ds)
  f = lambda i: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=bench.perturb(rw, i, 0)), dev, utils.DEFAULT_WARP_FUNC)
  f(-1); f(-2)
  t0 = time.perf_counter()
  for i in range(10): f(i)
  wall = (time.perf_counter() - t0) / 10 * 1e3
  ctx.profile_enable(1)
  hs = []
  for i in range(5):
    f(i); hs.append(ctx.profile_get()['host_enqueue'][0])
  ctx.profile_enable(0)
  print(f'{name}: wall {wall:.3f} ms per evaluation; host enqueue (profiling on) {np.median(hs):.3f} ms')
  dev.close()
