"""Experiment: what does staggering a batch buy?  cfg 4 tasks split over G contexts (own streams / workspaces), one host thread
each, all on the same GPU, against the whole batch in one context.  usage: batch_groups.py [tasks] [groups ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
data, raw = bench.cfg4_inputs()
keys = list(data)[:T]
p = defs.GPParams(model=raw)
order = sorted(keys, key=lambda k: -data[k][0].shape[0])
for G in groups:
  parts = [order[g::G] for g in range(G)]          # round-robin over the size-sorted tasks: equal work per group
  ctxs = [nat.default_context() if G == 1 else nat.Context(0) for _ in range(G)]
  devs = [objectives.DeviceDataset({k: defs.SubDataset(*data[k]) for k in part}, ctx=c) for part, c in zip(parts, ctxs)]
  def run(i, reps, out):
    tot = 0.0
    for _ in range(reps):
      v, _, g, _ = devs[i].evaluate(mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC, want_grad=True)
      tot += v
    out[i] = tot
  def timed(reps):
    out = [0.0] * G
    th = [threading.Thread(target=run, args=(i, reps, out)) for i in range(G)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / reps * 1e3, sum(out) / reps
  timed(2)
  ts = [timed(6)[0] for _ in range(3)]
  print(f'T={T} groups={G}: {np.median(ts):.2f} ms per evaluation of the whole batch  ({" ".join("%.2f" % t for t in ts)})  nll sum {timed(1)[1]:.6f}', flush=True)
  for d in devs: d.close()
