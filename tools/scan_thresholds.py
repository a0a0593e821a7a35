"""The scheduling thresholds of sched.hip, re-derived: for one fp64 matrix of N points (NLL + gradient, SE-ARD, D = 16) the time
under each alternative of the choices the schedule makes by block count -- one-sweep inverse on / off (`sweep`), the 64-tile limit
(`small_nblk`), panels per trailing update (`potrf_group`), look-ahead on / off -- median of 3 interleaved rounds, plus the value
and gradient difference against the default (schedules must agree to rounding).  Prints a markdown table
(profiles/r05_sched_thresholds.md is this output): the default must be the fastest column, or within the box-to-box noise of it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

ctx = nat.default_context()
DEFAULTS = {'sweep': 1, 'small_nblk': -1, 'potrf_group': 0, 'lookahead': 1}
SETTINGS = [('default', {}), ('sweep=0', {'sweep': 0}), ('sweep=2', {'sweep': 2}), ('small_nblk=24', {'small_nblk': 24}), ('small_nblk=32', {'small_nblk': 32}),
            ('small_nblk=48', {'small_nblk': 48}), ('small_nblk=64', {'small_nblk': 64}), ('group=2', {'potrf_group': 2}), ('group=3', {'potrf_group': 3}),
            ('group=4', {'potrf_group': 4}), ('group=6', {'potrf_group': 6}), ('lookahead=0', {'lookahead': 0})]
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 7168, 8192]


def flat(g):
  return np.concatenate([np.ravel(np.asarray(g[k], dtype=np.float64)) for k in sorted(g)])


print('| N (blocks) | ' + ' | '.join(n for n, _ in SETTINGS) + ' | max dNLL, dgrad |')
print('|---|' + '---|' * (len(SETTINGS) + 1))
for n in sizes:
  x, y, raw = bench.cfg2_inputs(n=n)
  dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
  p = defs.GPParams(model=raw)
  f = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
  reps = 12 if n <= 2048 else (6 if n <= 5120 else 4)
  times = {name: [] for name, _ in SETTINGS}
  ref = None; dv = dg = 0.0
  for rnd in range(3):
    for name, opts in SETTINGS:
      for k, v in DEFAULTS.items():
        ctx.set_option(k, opts.get(k, v))
      v0, g0 = f()
      t0 = time.perf_counter()
      for _ in range(reps):
        v0, g0 = f()
      times[name].append((time.perf_counter() - t0) / reps * 1e3)
      if ref is None:
        ref = (v0, flat(g0))
      else:
        dv = max(dv, abs(v0 - ref[0]) / abs(ref[0])); dg = max(dg, float(np.max(np.abs(flat(g0) - ref[1])) / np.max(np.abs(ref[1]))))
  for k, v in DEFAULTS.items():
    ctx.set_option(k, v)
  dev.close()
  med = {name: float(np.median(t)) for name, t in times.items()}
  best = min(med.values())
  cells = [('**%.3f**' if med[name] <= best * 1.005 else '%.3f') % med[name] for name, _ in SETTINGS]
  print(f'| {n} ({-(-n // 128)}) | ' + ' | '.join(cells) + f' | {dv:.1e}, {dg:.1e} |', flush=True)
