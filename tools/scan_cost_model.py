"""Cost model of one rank's task shard (bench.py / parallel.lpt_partition): ms per NLL + gradient of a batch of fp64 SE-ARD tasks
(D = 4) as a function of the batch -- T = c0 + a * max_k nblk_k + b * sum_k n_k^3 -- fitted by least squares on measured batches
of 2..16 tasks of 1024..2560 points.  Prints the table and the fit; the constants go to hyperbo_amd/parallel.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

rng = np.random.default_rng(0)
_, raw = bench.cfg4_inputs()
d = 4
def task(n):
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  return defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1)))
rows = []
cases = [(t, [n] * t) for t in (2, 4, 8, 12, 16) for n in (1024, 1536, 2048, 2432)]
cases += [(8, list(rng.integers(1600, 2401, size=8))) for _ in range(6)] + [(4, list(rng.integers(1600, 2401, size=4))) for _ in range(3)]
for t, sizes in cases:
  dev = objectives.DeviceDataset({i: task(int(n)) for i, n in enumerate(sizes)})
  f = lambda i: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=bench.perturb(raw, i, 0)), dev, utils.DEFAULT_WARP_FUNC)
  f(-1); f(-2)
  t0 = time.perf_counter()
  for i in range(8):
    f(i)
  ms = (time.perf_counter() - t0) / 8 * 1e3
  dev.close()
  nblk = [-(-int(n) // 128) for n in sizes]
  rows.append((t, max(nblk), sum(nblk), float(sum(float(n)**3 for n in sizes)), ms))
  print(f'tasks {t:3d}  max_nblk {max(nblk):3d}  sum_nblk {sum(nblk):4d}  sum_n3 {rows[-1][3]:.3e}  {ms:.3f} ms', flush=True)
A = np.array([[1.0, r[1], r[3]] for r in rows]); y = np.array([r[4] for r in rows])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print('fit  T[ms] = c0 + a * max_nblk + b * sum n^3:  c0 = %.4f  a = %.5f  b = %.4e' % tuple(coef))
print('max relative residual %.3f, rms %.3f' % (np.max(np.abs(pred - y) / y), np.sqrt(np.mean(((pred - y) / y)**2))))
