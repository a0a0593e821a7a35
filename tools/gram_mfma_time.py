"""The fp32 cross Gram of cfg 3's shape (16384 x 8192, 64 features) through hbo_gram, matrix-core form and direct form -- run under
`rocprofv3 --kernel-trace --stats` (kernel durations) or `--pmc ...` (tools/pmc_kernel.py); the host copy of the result is not the point.
  gram_mfma_time.py [n1 n2 d [gram_mfma]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import definitions as defs
from hyperbo_amd.gp_utils import kernel, utils
n1, n2, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 8192, 64)
ctx = nat.default_context()
if len(sys.argv) > 4:
    ctx.set_option('gram_mfma', int(sys.argv[4]))
rng = np.random.default_rng(0)
x1 = np.tanh(rng.normal(size=(n1, d))).astype(np.float32); x2 = np.tanh(rng.normal(size=(n2, d))).astype(np.float32)
isp = lambda v: np.log(np.expm1(np.asarray(v, dtype=np.float64))).astype(np.float32)
p = defs.GPParams(model={'lengthscale': isp(np.ones(d)), 'signal_variance': isp(1.0), 'noise_variance': isp(1e-2)})
for _ in range(4):
    t0 = time.perf_counter(); g = kernel.matern52(p, x1, x2, warp_func=utils.DEFAULT_WARP_FUNC); t1 = time.perf_counter()
print('hbo_gram %d x %d, %d features: %.1f ms incl. the copy to the host; checksum %.6f' % (n1, n2, d, (t1 - t0) * 1e3, float(g[::97, ::89].sum())))
