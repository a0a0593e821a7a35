"""cfg 3 (N = 16384 fp32, Matern-5/2 on 64 MLP features, EI over 65 536 candidates): stage times of the streamed posterior with the
producer side overlapped (default) and serialised (hbo_tune post_serial = 1: every stage alone on the machine)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hyperbo_amd import _native as nat
ctx = nat.default_context()
for serial in (0, 1, 0):
  ctx.set_option('post_serial', serial)
  r = bench.bench_cfg3(ctx, stages=True)
  print('post_serial', serial, {k: r[k] for k in ('factor_ms', 'ei_ms', 'post_gemm_ms', 'stages_ei')})
ctx.set_option('post_serial', 0)
