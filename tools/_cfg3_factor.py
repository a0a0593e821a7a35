import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
ctx = nat.default_context()
for k, v in [a.split('=') for a in sys.argv[1:]]:
  ctx.set_option(k, int(v))
r = bench.bench_cfg3(ctx)
print({k: r[k] for k in ('factor_ms', 'potrf_ms', 'trtri_ms', 'ei_ms')})
