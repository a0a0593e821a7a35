"""fp32 factorisations with the trailing updates / inverse products on two-way fp16 splits (hbo_tune chol_f16x2 = 1, default for the
stationary covariances) against the exact three-way bf16 splits (0): cfg 3's factor stages, the cfg-2 shape in fp32 (time and
error against fp64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hyperbo_amd import _native as nat
ctx = nat.default_context()
for v in (0, 1, 0, 1):
  ctx.set_option('chol_f16x2', v)
  r = bench.bench_cfg3(ctx)
  f = bench.bench_fp32_objective(ctx)
  print('chol_f16x2', v, {k: r[k] for k in ('factor_ms', 'potrf_ms', 'trtri_ms', 'ei_ms')},
        {k: f[k] for k in ('ms_per_eval', 'nll_rel_err_vs_fp64', 'grad_err_over_max_vs_fp64')}, flush=True)
ctx.set_option('chol_f16x2', 1)
