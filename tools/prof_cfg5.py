"""cfg5: SE-ARD, N=65536, D=16, fp64: Gram (32 GiB) + blocked Cholesky (NLL value only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x, y, raw = bench.cfg2_inputs(seed=5, n=n)
raw['noise_variance'] = bench.inv_softplus(1e-1)
dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
ctx = nat.default_context()
for opt in sys.argv[2:]:
    k, v = opt.split('='); ctx.set_option(k, int(v))
ctx.profile_enable(1)
p = defs.GPParams(model=raw)
for it in range(2):
    t0 = time.perf_counter(); v = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC); t1 = time.perf_counter()
    prof = ctx.profile_get()
    print(f'N={n}: NLL {v:.6f} in {1e3*(t1-t0):.1f} ms; potrf {prof["potrf"][0]:.1f} ms = {n**3/3/(prof["potrf"][0]*1e-3)/1e12:.1f} TFLOP/s; gram {prof["gram"][0]:.2f} ms = {8*n*(n+1)/2/(prof["gram"][0]*1e-3)/1e12:.2f} TB/s written; trailing {prof["syrk_trailing"][0]:.1f} ms')
