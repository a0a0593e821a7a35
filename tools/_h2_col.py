import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
ctx = nat.default_context()
for col in (0, 1, 0, 1):
  ctx.set_option('syrk3_col', col)
  r = bench.bench_cfg3(ctx)
  f = bench.bench_fp32_objective(ctx)
  print('syrk3_col', col, {k: r[k] for k in ('factor_ms', 'potrf_ms', 'trtri_ms')},
        {k: f[k] for k in ('ms_per_eval', 'nll_rel_err_vs_fp64', 'grad_err_over_max_vs_fp64')}, flush=True)
ctx.set_option('syrk3_col', 0)
for grp in (4, 6, 8, 12, 16):
  ctx.set_option('potrf_group', grp)
  r = bench.bench_cfg3(ctx)
  print('group', grp, {k: r[k] for k in ('factor_ms', 'potrf_ms', 'trtri_ms')}, flush=True)
