"""bench.py -- headline benchmark of the MI355X-native HyperBO GP hot path.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env) the
ranks find each other through hyperbo_amd.parallel.SocketGroup on a port derived from MASTER_PORT; launched plainly
(`python bench.py --gpus 8`, no WORLD_SIZE) this process SPAWNS the N ranks itself (HBO_DEVICE = rank) and relays rank
0's JSON line.  Either way the data path is torch-free: rendezvous / barrier / max-over-ranks over localhost sockets,
the [nll, count, grad] all-reduce through libhbo's own RCCL binding (hbo_comm_*, xGMI); torch is never imported.

metric (BASELINE.json): GP NLL+grad evaluations/sec at N=8192, D=16, fp64  (configs[1]).
A "step" is one NLL+gradient evaluation of a single-task SE-ARD GP (X, y resident in HBM; only the
hyper-parameters change between evaluations, as in the reference's training loop gp.py:132-144).
At N>1 GPUs a single factorisation does not shard ("replicas only"): every rank runs
its own evaluation stream (different theta -- line-search points / restarts) and `value` is the
aggregate.  The task-sharded multi-task objective (configs[3]: 64 PD1-shaped sub-datasets, one
RCCL all-reduce of [nll, grad] per evaluation) is timed in the same run and reported under
"multitask"; configs[2] (factor + EI over 65 536 candidates, fp32) and configs[4] (N=65 536 Gram + Cholesky) are
timed on rank 0 at N=1 under "cfg3" / "cfg5".

Extra objects on the JSON line:
  roofline       -- the Cholesky trailing-update kernel (fp64 MFMA syrk): algorithmic flops per launch
                    / HIP-event duration of those launches inside the timed region.
  roofline_small -- the 64x64-tile trailing updates of the panel stream (column updates inside a group, the next group's
                    block columns, small remainders): the kernel class with the most kernel time per evaluation
                    (per-launch events of a separate untimed pass).
  roofline_potrf -- the WHOLE factorisation: N^3/3 flops / its HIP-event time (with the overlapped inverse beside it).
  cpu_baseline   -- oracle/cpu_baseline.py (NumPy/SciPy-LAPACK port, kind "port") on the host cores, with its thread / BLAS
                    provenance; jax_baseline: a JAX expression of the same formulas, only if `import jax` works on the box.
  train, bo_step -- GP.train() ms per step (Adam with per-step sub-sampling / L-BFGS) and one BO step at N = 8100 (O(N^2)
                    cache append + posterior at 64 queries vs re-factorisation).
  device         -- hipGetDeviceProperties name / CUs / memory, and a short GPU leg AFTER the CPU baseline (a sampler that only
                    looks at the end of the run sees the GPU working).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); the image
# exports it already -- set here too so that self-spawned ranks and bare shells get it before the HIP runtime loads
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet (v_mfma_f64_16x16x4_f64: 64 cycles per SIMD); the rate sustained in THIS run is measured
                               # by hbo_mfma_peak_probe and reported as peak_ubench / frac_of_ubench beside it
HBM_PEAK_TBPS = 8.0            # /opt/skills/guides/MI355X_MICROARCH.md
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)


def inv_softplus(v):
  return np.log(np.expm1(np.asarray(v, dtype=np.float64)))


def cfg2_inputs(seed=2, n=8192, d=16):
  """SURVEY.md 8(d) cfg 2: X~U[0,1]^{N x D}, y = sin(2 pi X w) + 0.1 eps; SE-ARD ls=sqrt(D)*0.3."""
  rng = np.random.Generator(np.random.PCG64(seed))
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
  raw = {'lengthscale': inv_softplus(np.full(d, np.sqrt(d) * 0.3)), 'signal_variance': inv_softplus(1.0),
         'noise_variance': inv_softplus(1e-2), 'constant': np.array(0.0)}
  return x, y, raw


def cfg4_inputs(seed=4, tasks=64, d=4):
  """SURVEY.md 8(d) cfg 4: T=64 tasks, N_k ~ U{1600..2400}, D=4, shared theta."""
  rng = np.random.Generator(np.random.PCG64(seed))
  sizes = rng.integers(1600, 2401, size=tasks)
  data = {}
  for k, n in enumerate(sizes):
    x = rng.uniform(size=(int(n), d)); w = rng.normal(size=d)
    data[k] = (x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(int(n), 1)))
  raw = {'lengthscale': inv_softplus(np.full(d, 0.4)), 'signal_variance': inv_softplus(1.0),
         'noise_variance': inv_softplus(1e-2), 'constant': np.array(0.1)}
  return data, raw


def perturb(raw, step, rank):
  """theta changes every evaluation (as under an optimiser); deterministic tiny perturbation."""
  out = dict(raw)
  out['lengthscale'] = raw['lengthscale'] + 1e-3 * np.sin(0.37 * (step + 1) + rank + np.arange(raw['lengthscale'].size))
  return out


def bulk_update_flops(n, group):
  """Algorithmic flops of every BULK trailing-update launch (gemm_kernel<double,true,true,128>): with
  look-ahead the update of panel group g is split into F1 (next group's block columns, 64x64 tiles,
  latency-critical) and F2 (the remaining m = nblk - g2 tile columns); F2 runs on 128x128 tiles while
  m(m+1)/2 >= 600 (hyperbo_amd/csrc/sched.hip run_potrf).  Lower triangle incl. diagonal tiles, K = 128*group."""
  nblk = (n + 127) // 128
  out = []
  for g0 in range(0, nblk, group):
    g1 = min(g0 + group, nblk)
    g2 = min(g1 + group, nblk)
    m = nblk - g2
    if g1 < nblk and g2 < nblk and m * (m + 1) // 2 >= 600:
      out.append(m * (m + 1) / 2 * 128 * 128 * 2.0 * 128 * (g1 - g0))
  return out


def small_update_flops(n, group):
  """Tile flops of the 64x64-tile SYRK launches of the panel stream (hyperbo_amd/csrc/sched.hip:run_potrf): the
  left-looking column updates inside a group (K = 128 .. 128*(group-1)), F1 = the next group's block columns
  (K = 128*group) and the bulk updates that have become too small for 128-tiles.  Lower tiles incl. the diagonal ones."""
  nblk = (n + 127) // 128
  fl = 0.0
  tile = 128.0 * 128 * 2 * 128
  for g0 in range(0, nblk, group):
    g1 = min(g0 + group, nblk)
    g2 = min(g1 + group, nblk)
    for p in range(g0 + 1, g1):
      fl += (nblk - p) * tile * (p - g0)
    if g1 < nblk:
      for c in range(g1, g2):
        fl += (nblk - c) * tile * (g1 - g0)
      m = nblk - g2
      if g2 < nblk and m * (m + 1) // 2 < 600:
        fl += m * (m + 1) / 2 * tile * (g1 - g0)
  return fl


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU, HBO_DEVICE = rank), pass our own
  arguments through, relay rank 0's JSON line.  The ranks meet on a loopback port: rank 0 takes the first free one of a
  range chosen here (no bind-then-close race), the others probe the range.  A rank that dies takes the others with it:
  the line then carries the error instead of a half-formed group waiting for its time-out."""
  import random
  import subprocess
  import uuid
  port = 20000 + random.SystemRandom().randrange(20000)
  token = uuid.uuid4().hex
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), HBO_DEVICE=str(r),
               HBO_BENCH_PORT=str(port), HBO_BENCH_TOKEN=token)
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                  stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
  failed = None
  while failed is None and any(p.poll() is None for p in procs):
    for r, p in enumerate(procs):
      if p.poll() not in (None, 0):
        failed = (r, p.returncode)
    time.sleep(0.2)
  if failed is None:
    failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
  if failed is not None:
    for p in procs:
      if p.poll() is None:
        p.terminate()
  out = procs[0].stdout.read() if procs[0].stdout else ''
  for p in procs:
    try:
      p.wait(timeout=10)
    except subprocess.TimeoutExpired:
      p.kill()
  if failed is not None and not out.strip():
    out = json.dumps({'metric': 'GP NLL+grad evals/sec at N=8192 D=16 fp64', 'value': None, 'n_gpus': n,
                      'error': f'rank {failed[0]} exited with code {failed[1]}; the other ranks were stopped'}) + '\n'
  sys.stdout.write(out)
  sys.stdout.flush()
  return 0 if failed is None else max(1, abs(failed[1]))


def bench_cfg3(ctx, stages=False):
  """BASELINE.json configs[2]: Matern-5/2 on tanh-MLP(32->64) features + linear_mlp mean, N=16384, fp32: factor once,
  EI over 65 536 candidates; stage times from HIP events (hbo_profile), wall times around the two calls."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.bo_utils import acfun
  from hyperbo_amd.gp_utils import gp, kernel, mean, utils
  rng = np.random.Generator(np.random.PCG64(3))
  d, f, n, m = 32, 64, 16384, 65536
  model = {'lengthscale': inv_softplus(np.ones(f)), 'signal_variance': inv_softplus(1.0), 'noise_variance': inv_softplus(1e-2),
           'mlp_params': {'Dense_0': {'kernel': rng.normal(size=(d, f)) / np.sqrt(d), 'bias': np.zeros(f)}},
           'linear_mean': {'kernel': rng.normal(size=(f, 1)) / np.sqrt(f), 'bias': np.zeros(1)}}
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  x = rng.uniform(size=(n, d)).astype(np.float32)
  y = (np.sin(x[:, :4].sum(axis=1, keepdims=True) * 2.0) + 0.1 * rng.normal(size=(n, 1))).astype(np.float32)
  xq = rng.uniform(size=(m, d)).astype(np.float32)
  g = gp.GP({0: defs.SubDataset(x, y)}, mean.linear_mlp, kernel.matern52_mlp,
            defs.GPParams(model=to32(model), config={'mlp_features': (f,)}), utils.DEFAULT_WARP_FUNC)
  ctx.profile_enable(1)
  def timed(reps):
    best = None
    for _ in range(reps):
      g.update_model_params(g.params.model)        # drops the cache -> re-factorise
      t0 = time.perf_counter(); g.setup_predictor(0); t1 = time.perf_counter()
      pf = ctx.profile_get()
      ei = acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq); t2 = time.perf_counter()
      pe = ctx.profile_get()
      cur = (t1 - t0, t2 - t1, pf, pe)
      if best is None or cur[0] + cur[1] < best[0] + best[1]:
        best = cur
    mu, var = g.predict(xq, 0)
    return best, (np.asarray(mu, dtype=np.float64), np.asarray(var, dtype=np.float64), np.asarray(ei, dtype=np.float64))
  # BOTH fp32 forms of the matrix-core products (round-5 review): the default two-way fp16 split (3 MFMAs, 2^-22 per product) and
  # the exact three-way bf16 split (6 MFMAs, every fp32 bit of both operands), each timed and each compared with the fp64 run below
  paths = {}
  ctx.set_option('post_f16x2', 0); ctx.set_option('chol_f16x2', 0)
  try:
    paths['bf16x3'] = timed(2)
  finally:
    ctx.set_option('post_f16x2', 1); ctx.set_option('chol_f16x2', 1)
  paths['f16x2'] = timed(3)
  best, (_, _, ei) = paths['f16x2']
  # every stage of the streamed posterior ALONE on the machine (hbo_tune post_serial: the cross Gram of chunk i + 1 no longer runs
  # beside the product of chunk i): what the co-running cross-Gram kernel costs in isolation, and what the product loses beside it
  ctx.set_option('post_serial', 1)
  try:
    acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq)
    t0 = time.perf_counter(); acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq); te_serial = time.perf_counter() - t0
    ps_ = ctx.profile_get()
  finally:
    ctx.set_option('post_serial', 0)
  ctx.profile_enable(0)
  assert np.isfinite(ei).all() and (ei >= 0).all()
  tf, te, pf, pe = best
  post_ms = pe['post_gemm'][0]
  post_tf = float(n) * n * m / (post_ms * 1e-3) / 1e12
  # the fp32 product runs on the bf16 matrix cores: both operands split exactly into three bf16 pieces, six bf16 MFMAs per
  # fp32 product (csrc/post3.hip) -- executed bf16 flops = 6 x the algorithmic fp32 flops
  out = {'workload': 'cfg3: Matern-5/2 o tanh-MLP(32->64) + linear_mlp mean, N=16384, fp32, factor + EI over 65536 candidates',
          'factor_ms': round(tf * 1e3, 2), 'potrf_ms': round(pf['potrf'][0], 2), 'trtri_ms': round(pf['trtri'][0], 2),
          'ei_ms': round(te * 1e3, 2), 'post_gemm_ms': round(post_ms, 2), 'post_gemm_tflops': round(post_tf, 1),
          'post_gemm_path': 'f16x2 (two-way fp16 split of both fp32 operands scaled by powers of two, 3 fp16 MFMAs per product, fp32 '
                            'accumulate; 2^-22 per product, as close to fp64 as the fp32-MFMA product: csrc/post2h.hip; round 4: bf16x3, 6 MFMAs)',
          'frac_f16_executed': round(3.0 * post_tf / BF16_MFMA_PEAK_TFLOPS, 4), 'ei_flops': float(n) * n * m,
          'factor_frac_fp32': round(float(n)**3 / 3 / (pf['potrf'][0] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
          'note': 'wall times include the host<->device copies of x_query / EI; post_gemm = V = L^-1 Kxq (algorithmic fp32 flops N^2 M); '
                  'frac_f16_executed = 3 x algorithmic flops (the fp16 MFMAs the product executes) against the dense fp16 / bf16 MFMA peak; '
                  'factor_frac_fp32 = N^3/3 over the potrf stage time against the fp32 MFMA peak; trailing updates of the factorisation and products of the inverse run as f16x2 too (round 5: factor 21.5 -> 17.6 ms)'}
  chunks = ps_['cross_gram'][1]
  xg_ms = ps_['cross_gram'][0] / chunks
  ch = m // chunks
  xg_flops = float(n) * ch * (3 * f + 24)      # per pair: F x (subtract, multiply, add) + sqrt, exp, Matern polynomial
  out['roofline_xgram'] = {
      'kernel': 'gram_kernel<float,true,2>: cross Gram k(X, Xq) of one chunk of %d candidates x %d training points, %d features' % (ch, n, f),
      'isolated_ms_per_chunk': round(xg_ms, 3), 'beside_the_product_ms_per_chunk': round(pe['cross_gram'][0] / pe['cross_gram'][1], 3),
      'hbm_written_gb': round(4.0 * n * ch / 1e9, 3), 'hbm_tbps': round(4.0 * n * ch / (xg_ms * 1e-3) / 1e12, 3),
      'hbm_frac': round(4.0 * n * ch / (xg_ms * 1e-3) / 1e12 / HBM_PEAK_TBPS, 4),
      'valu_tflops': round(xg_flops / (xg_ms * 1e-3) / 1e12, 2), 'valu_frac_fp32': round(xg_flops / (xg_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
      'post_gemm_isolated_ms': round(ps_['post_gemm'][0], 2), 'post_gemm_beside_the_gram_ms': round(post_ms, 2),
      'ei_ms_serialised': round(te_serial * 1e3, 2),
      'note': 'isolated = hbo_tune post_serial 1 (producer and consumer of the streamed posterior on one stream); the default overlaps '
              'the cross Gram of chunk i + 1 with the product of chunk i, where it runs in the slots the resident product grid leaves'}
  # the same model and candidates in fp64 on the device: the yardstick both fp32 forms are held to (max abs differences over all 65 536
  # candidates; tests/test_gpu_parity.py::test_cfg3_full_size_against_host_lapack holds the fp64 run itself to host LAPACK)
  g64 = gp.GP({0: defs.SubDataset(x.astype(np.float64), y.astype(np.float64))}, mean.linear_mlp, kernel.matern52_mlp,
              defs.GPParams(model=model, config={'mlp_features': (f,)}), utils.DEFAULT_WARP_FUNC)
  xq64 = xq.astype(np.float64)
  mu64, var64 = g64.predict(xq64, 0)
  ei64 = np.asarray(acfun.expected_improvement(model=g64, sub_dataset_key=0, x_queries=xq64), dtype=np.float64)
  mu64, var64 = np.asarray(mu64, dtype=np.float64), np.asarray(var64, dtype=np.float64)
  del g64
  out['default_path'] = 'f16x2'
  out['paths'] = {}
  for name, ((tf_, te_, pf_, pe_), (mu_, var_, ei_)) in paths.items():
    out['paths'][name] = {
        'factor_ms': round(tf_ * 1e3, 2), 'potrf_ms': round(pf_['potrf'][0], 2), 'ei_ms': round(te_ * 1e3, 2), 'post_gemm_ms': round(pe_['post_gemm'][0], 2),
        'mfmas_per_product': 3 if name == 'f16x2' else 6, 'bits_per_product': 22 if name == 'f16x2' else 24,
        'max_abs_dmu_vs_fp64': float(np.max(np.abs(mu_ - mu64))), 'max_abs_dvar_vs_fp64': float(np.max(np.abs(var_ - var64))),
        'max_abs_dei_vs_fp64': float(np.max(np.abs(ei_ - ei64))),
        'scale_mu_var_ei': [float(np.max(np.abs(mu64))), float(np.max(np.abs(var64))), float(np.max(np.abs(ei64)))]}
  out['paths_note'] = ('f16x2 (default for the stationary covariances): two-way fp16 split of operands scaled by powers of two, 3 fp16 MFMAs per product; '
                       'bf16x3 (hbo_tune post_f16x2 = chol_f16x2 = 0; always used by the dot-product kernel and hbo_spd_*): exact three-way bf16 split, '
                       '6 MFMAs per product = strict fp32.  Top-level factor_ms / ei_ms of this object are the DEFAULT path.')
  if stages:
    out['stages_ei'] = {k: (round(v[0], 3), v[1]) for k, v in pe.items()}
    out['stages_factor'] = {k: (round(v[0], 3), v[1]) for k, v in pf.items()}
  return out


def bench_cfg5(ctx):
  """BASELINE.json configs[4]: SE-ARD, N=65536, D=16, fp64: Gram (32 GiB) + blocked Cholesky (NLL value only)."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
  n = 65536
  x, y, raw = cfg2_inputs(seed=5, n=n)
  raw['noise_variance'] = inv_softplus(1e-1)
  dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
  ctx.set_option('potrf_group', 0)
  ctx.profile_enable(1)
  p = defs.GPParams(model=raw)
  best = None
  for _ in range(2):
    t0 = time.perf_counter()
    v = objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, p, dev, utils.DEFAULT_WARP_FUNC)
    el = time.perf_counter() - t0
    prof = ctx.profile_get()
    if best is None or el < best[0]:
      best = (el, prof, v)
  ctx.profile_enable(0)
  dev.close()
  el, prof, v = best
  assert np.isfinite(v)
  potrf_s = prof['potrf'][0] * 1e-3
  gram_ms = prof['gram'][0]
  tf = float(n)**3 / 3 / potrf_s / 1e12
  return {'workload': 'cfg5: SE-ARD N=65536 D=16 fp64, Gram (lower tiles, 17 GB written) + blocked Cholesky, NLL value',
          'nll': float(v), 'total_s': round(el, 3), 'gram_ms': round(gram_ms, 2),
          'gram_tbps': round(8.0 * n * (n + 1) / 2 / (gram_ms * 1e-3) / 1e12, 3), 'potrf_s': round(potrf_s, 4),
          'potrf_tflops': round(tf, 2), 'frac': round(tf / FP64_MFMA_PEAK_TFLOPS, 4)}


def bench_fp32_objective(ctx):
  """The headline workload in the reference's DEFAULT dtype (float32 unless JAX_ENABLE_X64): cfg-2 shape, fp32
  NLL + gradient, theta changing every evaluation; the large products (trailing updates, inverse, K^-1 = W^T W) run on the fp16
  matrix cores from two-way splits scaled by powers of two (hbo_tune chol_f16x2; the dot-product kernel and hbo_spd_* keep the
  exact three-way bf16 splits).  Error against the fp64 evaluation of the same theta beside it."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import kernel, mean, objectives, utils
  x, y, raw = cfg2_inputs()
  to32 = lambda t: {k: to32(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, dtype=np.float32)
  d32 = objectives.DeviceDataset({0: defs.SubDataset(x.astype(np.float32), y.astype(np.float32))})
  f = lambda i: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=to32(perturb(raw, i, 0))), d32,
                                              utils.DEFAULT_WARP_FUNC)
  f(0); f(1)
  steps = 10
  t0 = time.perf_counter()
  for i in range(steps):
    v32, g32 = f(2 + i)
  el = time.perf_counter() - t0
  d32.close()
  d64 = objectives.DeviceDataset({0: defs.SubDataset(x, y)})
  v64, g64 = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, defs.GPParams(model=perturb(raw, 1 + steps, 0)), d64,
                                           utils.DEFAULT_WARP_FUNC)
  d64.close()
  flat = lambda t: np.concatenate([np.ravel(np.asarray(t[k], dtype=np.float64)) for k in sorted(t)])
  return {'workload': f'cfg-2 shape in fp32 (N={x.shape[0]}, D={x.shape[1]}): NLL+grad, products on the fp16 matrix cores (f16x2: 3 MFMAs per product; round 4: bf16x3, 6)',
          'ms_per_eval': round(el / steps * 1e3, 3), 'evals_per_s': round(steps / el, 2),
          'nll_rel_err_vs_fp64': float(abs(v32 - v64) / abs(v64)),
          'grad_err_over_max_vs_fp64': float(np.max(np.abs(flat(g32) - flat(g64))) / np.max(np.abs(flat(g64))))}


def bench_train():
  """GP.train() (gp.py:53-195): Adam with a fresh sub-sampled batch every step, and L-BFGS on one resident batch -- the
  callers that turn NLL+grad evaluations into pre-training time.  64 tasks x 2000 points, batch_size 500."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import gp, kernel, mean, objectives, utils
  rng = np.random.default_rng(0)
  tasks, n, d, bs = 64, 2000, 4, 500
  data = {}
  for k in range(tasks):
    x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
    data[k] = defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1)))
  model = lambda: {'lengthscale': np.zeros(d), 'signal_variance': np.array(0.0), 'noise_variance': np.array(-2.0), 'constant': np.array(0.0)}
  out = {'workload': f'GP.train(): {tasks} tasks x {n} points, D={d}, batch_size {bs}, fp64, SE-ARD + constant mean'}
  for method, st in (('adam', 40), ('lbfgs', 8)):
    p = defs.GPParams(model=model(), config={'method': method, 'batch_size': bs, 'max_training_step': st, 'learning_rate': 0.01,
                                             'objective': objectives.nll})
    g = gp.GP(data, mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC)
    g.train(key=1)
    g.params.model = model()
    t0 = time.perf_counter(); g.train(key=2); el = time.perf_counter() - t0
    out[f'{method}_ms_per_step'] = round(el / st * 1e3, 3)
  # the reference's own training regime (gp_test.py:58-148, data_utils.py:72-100): many small sub-datasets -- every task fits one
  # 128-block, so an evaluation is ONE launch (small.hip) + the batched MLP passes (mlp.hip); blocked = the same with
  # hbo_tune small_fused = 0
  from hyperbo_amd import _native as nat
  ctx = nat.default_context()
  small = {'workload': 'Adam steps on 24 sub-datasets x 100 points, D=4, fp64 (batch_size > n: resident)'}
  tasks, n = 24, 100
  data = {}
  for k in range(tasks):
    x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
    data[k] = defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1)))
  feats = (8, 8)
  def mlp_model():
    mm = model(); mm['lengthscale'] = np.zeros(feats[-1]); fin = d
    mm['mlp_params'] = {}
    for l, f in enumerate(feats):
      mm['mlp_params'][f'Dense_{l}'] = {'kernel': rng.normal(size=(fin, f)) / np.sqrt(fin), 'bias': np.zeros(f)}; fin = f
    mm['linear_mean'] = {'kernel': rng.normal(size=(feats[-1], 1)) / np.sqrt(feats[-1]), 'bias': np.zeros(1)}
    return mm
  st = 200
  for name, mk, cov, mu in (('se_constant', model, kernel.squared_exponential, mean.constant),
                            ('matern52_mlp_linear_mlp', mlp_model, kernel.matern52_mlp, mean.linear_mlp)):
    for fused in (1, 0):
      ctx.set_option('small_fused', fused)
      try:
        el = None
        for rep in range(2):
          p = defs.GPParams(model=mk(), config={'method': 'adam', 'batch_size': n + 1, 'max_training_step': st, 'learning_rate': 1e-3,
                                                'objective': objectives.nll, 'mlp_features': feats})
          g = gp.GP(data, mu, cov, p, utils.DEFAULT_WARP_FUNC)
          t0 = time.perf_counter(); g.train(key=rep); el = time.perf_counter() - t0
      finally:
        ctx.set_option('small_fused', 1)
      small[f'{name}_ms_per_step' + ('' if fused else '_blocked')] = round(el / st * 1e3, 4)
  # the same with a FRESH batch per step (sub-datasets of 400 points, batch_size 100: rows gathered on the device from the resident
  # dataset; the indices of a batch are one vectorised draw on a helper thread)
  data4 = {}
  for k in range(tasks):
    x = rng.uniform(size=(400, d)); w = rng.normal(size=d)
    data4[k] = defs.SubDataset(x, np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(400, 1)))
  for fused in (1, 0):
    ctx.set_option('small_fused', fused)
    try:
      el = None
      for rep in range(2):
        p = defs.GPParams(model=model(), config={'method': 'adam', 'batch_size': 100, 'max_training_step': st, 'learning_rate': 1e-3, 'objective': objectives.nll})
        g = gp.GP(data4, mean.constant, kernel.squared_exponential, p, utils.DEFAULT_WARP_FUNC)
        t0 = time.perf_counter(); g.train(key=rep); el = time.perf_counter() - t0
    finally:
      ctx.set_option('small_fused', 1)
    small['se_constant_resampled_400_to_100_ms_per_step' + ('' if fused else '_blocked')] = round(el / st * 1e3, 4)
  out['small_tasks'] = small
  return out


def bench_bo_step():
  """One BO iteration at N = 8100 (bayesopt.py:186-190): append one observation to the cached factorisation and predict
  at 64 queries -- the O(N^2) row append against the reference's re-factorisation."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import gp, kernel, mean, utils
  n = 8100
  x, y, raw = cfg2_inputs(n=n + 8)
  xq = x[:64]
  out = {'workload': f'BO step: append 1 observation to a cached N={n} factorisation + posterior at 64 queries, fp64'}
  for inc, name in ((True, 'append_ms'), (False, 'refactor_ms')):
    m = gp.GP({0: defs.SubDataset(x[:n], y[:n])}, mean.constant, kernel.squared_exponential,
              defs.GPParams(model=raw, config={'incremental_cache': inc}), utils.DEFAULT_WARP_FUNC)
    m.predict(xq, 0)
    ts = []
    for i in range(4):
      m.update_sub_dataset((x[n + i:n + i + 1], y[n + i:n + i + 1]), 0, is_append=True)
      t0 = time.perf_counter(); m.predict(xq, 0); ts.append(time.perf_counter() - t0)
    out[name] = round(1e3 * float(np.median(ts)), 3)
    if inc:
      # the acquisition's value and gradient at 16 restart points against the same cache (bayesopt.py:116-125: what a
      # gradient-based maximiser of the acquisition calls in its inner loop)
      from hyperbo_amd.bo_utils import acfun
      f = lambda: acfun.expected_improvement.value_and_grad(model=m, sub_dataset_key=0, x_queries=xq[:16])
      f(); f()
      t0 = time.perf_counter()
      for _ in range(10):
        f()
      out['ei_value_and_grad_16_ms'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
  return out


def bench_hgp():
  """Acquisition on an HGP (hyperbo/bo_utils/acfun.py:72-82): S = 32 model-parameter samples of an SE-ARD GP over N = 512
  observations, EI at 64 queries.  `batched` = hbo_acq_samples (the S factorisations as one batch, posteriors queued on the
  device), `loop` = what the reference's structure gives: one factorisation + posterior per sample from Python."""
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.bo_utils import acfun
  from hyperbo_amd.gp_utils import gp, kernel, mean, utils
  rng = np.random.default_rng(7)
  n, d, S, M = 512, 8, 32, 64
  x = rng.uniform(size=(n, d)); w = rng.normal(size=d)
  y = np.sin(2 * np.pi * x @ w)[:, None] + 0.1 * rng.normal(size=(n, 1))
  xq = rng.uniform(size=(M, d))
  samples = [{'lengthscale': inv_softplus(np.full(d, 0.5)) + 0.2 * rng.normal(size=d), 'signal_variance': inv_softplus(1.0) + 0.1 * rng.normal(),
              'noise_variance': inv_softplus(1e-2) + 0.1 * rng.normal(), 'constant': np.array(0.1 * rng.normal())} for _ in range(S)]
  data = {0: defs.SubDataset(x, y)}
  hgp = gp.HGP(data, mean.constant, kernel.squared_exponential, defs.GPParams(model=samples[0], samples=samples), utils.DEFAULT_WARP_FUNC)
  batched = lambda: acfun.expected_improvement(model=hgp, sub_dataset_key=0, x_queries=xq)
  def loop():
    vals = []
    for smp in samples:
      g = gp.GP(data, mean.constant, kernel.squared_exponential, defs.GPParams(model=smp), utils.DEFAULT_WARP_FUNC)
      vals.append(acfun.expected_improvement(model=g, sub_dataset_key=0, x_queries=xq))
    return np.mean(vals, axis=0)
  a, b = batched(), loop()
  out = {'workload': f'EI on an HGP: {S} parameter samples x N={n} observations, D={d}, {M} queries, fp64',
         'max_abs_diff_batched_vs_loop': float(np.max(np.abs(a - b)))}
  for name, f in (('batched_ms', batched), ('loop_ms', loop)):
    f()
    t0 = time.perf_counter()
    for _ in range(5):
      f()
    out[name] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
  out['speedup'] = round(out['loop_ms'] / out['batched_ms'], 2)
  return out


def cpu_provenance():
  """Where the CPU baseline's flops come from: thread settings and the BLAS / LAPACK the port calls."""
  info = {'omp_num_threads': os.environ.get('OMP_NUM_THREADS'), 'openblas_num_threads': os.environ.get('OPENBLAS_NUM_THREADS')}
  try:
    cfg = np.show_config(mode='dicts')
    dep = cfg.get('Build Dependencies', {})
    info['numpy_blas'] = '%s %s' % (dep.get('blas', {}).get('name'), dep.get('blas', {}).get('version'))
  except Exception:  # pylint: disable=broad-except
    pass
  try:
    import scipy
    scfg = scipy.show_config(mode='dicts')
    dep = scfg.get('Build Dependencies', {})
    info['scipy_lapack'] = '%s %s' % (dep.get('lapack', {}).get('name'), dep.get('lapack', {}).get('version'))
  except Exception:  # pylint: disable=broad-except
    pass
  try:
    from threadpoolctl import threadpool_info
    info['threadpools'] = [{k: t.get(k) for k in ('user_api', 'internal_api', 'version', 'num_threads', 'threading_layer')} for t in threadpool_info()]
  except Exception:  # pylint: disable=broad-except
    pass
  return info


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--n', type=int, default=8192)
  ap.add_argument('--d', type=int, default=16)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-multitask', action='store_true')
  ap.add_argument('--no-extra', action='store_true', help='skip the cfg3 / cfg5 legs')
  ap.add_argument('--cpu-evals', type=int, default=8)
  ap.add_argument('--jax', action='store_true', help='also time a JAX expression of the same NLL+grad on the host (only if jax imports)')
  ap.add_argument('--secondary-timeout', type=float, default=420.0, help='seconds for the multitask + CPU legs')
  args = ap.parse_args()

  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    return spawn_ranks(args.gpus)

  # stdout carries ONE JSON line: librccl prints its version banner to fd 1 when the communicator is built, so fd 1 is
  # pointed at stderr for the duration of the run and the line goes out through the saved descriptor
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)

  def emit(line):
    os.write(json_fd, (json.dumps(line) + '\n').encode())

  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
  os.environ.setdefault('HBO_DEVICE', str(local_rank))
  pgroup = None
  if world > 1:
    from hyperbo_amd import parallel as _par
    if 'HBO_BENCH_PORT' in os.environ:     # spawned by this script: the port is ours
      pgroup = _par.SocketGroup(rank, world, int(os.environ['HBO_BENCH_PORT']), scan=32, token=os.environ.get('HBO_BENCH_TOKEN', ''))
    else:                                  # torch.distributed.run: MASTER_PORT belongs to the launcher, take one next to it
      base = int(os.environ.get('MASTER_PORT', '29500')) + 1
      pgroup = _par.SocketGroup(rank, world, base, scan=32,
                               token=os.environ.get('TORCHELASTIC_RUN_ID', '') + ':' + os.environ.get('MASTER_PORT', ''))

  from hyperbo_amd import _native as nat
  from hyperbo_amd import parallel
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

  ctx = nat.default_context()
  wf = utils.DEFAULT_WARP_FUNC
  # panels per trailing update: libhbo's own choice (option 0: 3 up to 96 blocks, 4 above, 8 from 256 blocks on and for fp32
  # on the bf16 cores -- sched.hip:run_potrf) unless $HBO_BENCH_POTRF_GROUP pins it; the roofline's algorithmic flops per
  # launch below are computed from the value in effect for the headline's fp64 matrix.  (Until round 3 the bench pinned 3 for
  # the whole context, which also put the cfg 3 / fp32 legs on 3 instead of their 8: factor 24.6 instead of 22.1 ms.)
  potrf_group = int(os.environ.get('HBO_BENCH_POTRF_GROUP', '0'))
  ctx.set_option('potrf_group', potrf_group)
  nblk_headline = (args.n + 127) // 128
  group_in_effect = potrf_group if potrf_group > 0 else (3 if nblk_headline <= 96 else (8 if nblk_headline >= 256 else 4))

  def sync():
    # every libhbo entry point returns with its streams drained (hbo.h: calls are synchronous), so the device is
    # idle here; the barrier lines the ranks up
    if pgroup is not None:
      pgroup.barrier()

  def max_over_ranks(t):
    return t if pgroup is None else pgroup.allreduce_max(t)

  # ---------------- headline: cfg 2 NLL+grad ------------------------------------------------
  x, y, raw = cfg2_inputs(n=args.n, d=args.d)
  dev = objectives.DeviceDataset({0: defs.SubDataset(x, y)})

  def step_fn(i):
    p = defs.GPParams(model=perturb(raw, i, rank))
    return objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev, wf)

  # timed region: only the launches of the dominant kernel (bulk trailing update) are bracketed with HIP events
  # (profile level -1); the per-stage table comes from a separate, untimed pass below
  ctx.profile_enable(-1)
  for i in range(args.warmup):
    step_fn(-1 - i)
  sync()
  prof = {}
  t0 = time.perf_counter()
  last = None
  for i in range(args.steps):
    last = step_fn(i)
    for k, (ms, cnt) in ctx.profile_get().items():
      a = prof.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
  sync()
  elapsed = max_over_ranks(time.perf_counter() - t0)
  stage_prof, stage_evals = {}, 3
  ctx.profile_enable(1)
  for i in range(stage_evals):
    step_fn(10_000 + i)
    for k, (ms, cnt) in ctx.profile_get().items():
      a = stage_prof.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
  launch_prof = {}
  ctx.profile_enable(2)          # per-launch events of the panel stream's kernels: one more untimed evaluation
  step_fn(20_000)
  for k, (ms, cnt) in ctx.profile_get().items():
    launch_prof[k] = (ms, cnt)
  assert np.isfinite(last[0]), 'NLL is not finite'
  ms_per_step = elapsed / args.steps * 1e3
  value = world * args.steps / elapsed

  group = group_in_effect
  fl = bulk_update_flops(args.n, group)
  roofline = None
  if 'syrk_bulk' in prof and prof['syrk_bulk'][1] > 0 and fl:
    tot_ms, launches = prof['syrk_bulk']
    # (hbo_tune f2_split puts a bulk update's leading columns in a launch of their own: more launches, the same tiles and flops)
    assert launches >= len(fl) * args.steps and launches % args.steps == 0, (launches, len(fl), args.steps)
    flops_total = sum(fl) * args.steps
    achieved = flops_total / (tot_ms * 1e-3) / 1e12
    roofline = {'bound': 'mfma', 'kernel': f'gemm_kernel<double,true,true,128> (bulk Cholesky trailing update, syrk K={128 * group})',
                'achieved': round(achieved, 3), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                'launches': launches, 'avg_launch_ms': round(tot_ms / launches, 4),
                'algorithmic_gflop_per_launch': round(flops_total / launches / 1e9, 3), 'launches_per_eval': launches // args.steps,
                'note': 'HIP events around each launch on its own stream inside the timed region; the launches '
                        'overlap with the look-ahead panel chain and the early triangular inverse on other streams'}
  if roofline is not None:
    # HBM traffic per launch from committed rocprofv3 PMC passes of this same command (separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md: on gfx950 it
    # reports half the bytes of wide coalesced streaming reads).  bench.py cannot run rocprof itself.
    pmc_path = next((pth for pth in (os.path.join(ROOT, 'profiles', f) for f in ('r06_pmc_hbm.json', 'r05_pmc_hbm.json', 'r04_pmc_hbm.json', 'r03_pmc_hbm.json')) if os.path.exists(pth)), None)
    if pmc_path:
      pmc_all = json.load(open(pmc_path))
      pmc = pmc_all.get('gemm_kernel<double, true, true, 128>')
      if pmc:
        roofline['traffic'] = int((2 * pmc['FETCH_SIZE_KB'] + pmc['WRITE_SIZE_KB']) * 1024)
        roofline['traffic_note'] = ('bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of this '
                                    f'command, profiles/{os.path.basename(pmc_path)}')
  roofline_small = None
  if 'syrk_col' in launch_prof and 'syrk_trailing' in launch_prof:
    sm_ms = launch_prof['syrk_col'][0] + launch_prof['syrk_trailing'][0]
    sm_fl = small_update_flops(args.n, group)
    sm_tf = sm_fl / (sm_ms * 1e-3) / 1e12
    roofline_small = {'bound': 'mfma', 'kernel': 'gemm_kernel<double,true,true,64> (panel-stream trailing updates: column updates inside a group, '
                                                 'next group\'s block columns, small remainders)',
                      'flops': sm_fl, 'ms': round(sm_ms, 4), 'launches': launch_prof['syrk_col'][1] + launch_prof['syrk_trailing'][1],
                      'achieved': round(sm_tf, 3), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(sm_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                      'chain_kernels_ms': {k: round(launch_prof[k][0], 4) for k in ('potf2', 'trsm', 'syrk_col', 'syrk_trailing') if k in launch_prof},
                      'note': 'sum of the per-launch HIP-event durations of ONE untimed evaluation (profile level 2); the launches sit '
                              'on the panel stream, one after the other, beside the bulk update and the overlapped inverse'}
  stages = {k: round(v[0] / stage_evals, 4) for k, v in stage_prof.items()}   # separate pass with all stage events on
  ctx.profile_enable(0)
  roofline_potrf = None
  if 'potrf' in stages and stages['potrf'] > 0:
    fl_potrf = float(args.n)**3 / 3.0
    tf = fl_potrf / (stages['potrf'] * 1e-3) / 1e12
    roofline_potrf = {'bound': 'mfma', 'what': 'whole blocked Cholesky (panel chain + trailing updates), stage events, '
                                                'with the overlapped part of the inverse running beside it',
                      'flops': fl_potrf, 'ms': stages['potrf'], 'achieved': round(tf, 3), 'peak': FP64_MFMA_PEAK_TFLOPS,
                      'unit': 'TFLOP/s', 'frac': round(tf / FP64_MFMA_PEAK_TFLOPS, 4)}

  # the whole evaluation against the fp64 MFMA roofline: N^3 algorithmic flops (potrf + inverse + K^-1) / wall time per step
  eval_tf = float(args.n)**3 / (ms_per_step * 1e-3) / 1e12
  roofline_eval = {'bound': 'mfma', 'what': 'whole NLL+grad evaluation: N^3 algorithmic flops (potrf N^3/3 + inverse N^3/3 + K^-1 N^3/3) over the '
                                             'wall time per step of the timed region',
                   'flops': float(args.n)**3, 'ms': round(ms_per_step, 4), 'achieved': round(eval_tf, 3), 'peak': FP64_MFMA_PEAK_TFLOPS,
                   'unit': 'TFLOP/s', 'frac': round(eval_tf / FP64_MFMA_PEAK_TFLOPS, 4)}
  def gram_traffic():
    pth = next((q for q in (os.path.join(ROOT, 'profiles', f) for f in ('r06_pmc_hbm.json', 'r05_pmc_hbm.json', 'r04_pmc_hbm.json')) if os.path.exists(q)), None)
    if pth is None:
      return None
    g = json.load(open(pth)).get('gram_kernel<double, true, 0>')
    return int((2 * g['FETCH_SIZE_KB'] + g['WRITE_SIZE_KB']) * 1024) if g else None
  # the Gram build against the HBM roofline: algorithmic bytes = the lower tiles it writes (+ X once), HIP-event time of the stage
  roofline_gram = None
  if 'gram' in stages and stages['gram'] > 0:
    nblk_g = (args.n + 127) // 128
    gram_bytes = 8.0 * (nblk_g * (nblk_g + 1) / 2 * 128 * 128 + args.n * args.d)
    gtb = gram_bytes / (stages['gram'] * 1e-3) / 1e12
    roofline_gram = {'bound': 'hbm', 'kernel': 'gram_kernel<double,true,0> (SE-ARD Gram, lower 128-tiles + (noise + eps) I, identity padding)',
                     'achieved': round(gtb * 1e3, 1), 'peak': HBM_PEAK_TBPS * 1e3, 'unit': 'GB/s', 'frac': round(gtb / HBM_PEAK_TBPS, 4),
                     'algorithmic_bytes': gram_bytes, 'ms': stages['gram'], 'traffic': gram_traffic(),
                     'note': 'bound by VALU issue, not by HBM (profiles/r04_elementwise.md): the stores are 272 MB at this size'}
  if roofline_gram is not None and rank == 0 and world == 1 and not args.no_extra:
    # the same kernel at D = 4 and D = 64 (the Gram stage of a value-only evaluation, HIP events): the distance loop costs 2 VALU
    # instructions per pair and feature, the epilogue ~27 per pair (fp64 exp), so D moves the kernel between its two bounds
    # (profiles/r05_gram_isa.md)
    other = {}
    for dd in (4, 64):
      try:
        xg, yg, rawg = cfg2_inputs(seed=2, n=args.n, d=dd)
        devg = objectives.DeviceDataset({0: defs.SubDataset(xg, yg)})
        ctx.profile_enable(1)
        pg = defs.GPParams(model=rawg)
        best_g = None
        for _ in range(3):
          objectives.neg_log_marginal_likelihood(mean.constant, kernel.squared_exponential, pg, devg, wf)
          gms = ctx.profile_get()['gram'][0]
          best_g = gms if best_g is None else min(best_g, gms)
        ctx.profile_enable(0)
        devg.close()
        gb = 8.0 * (nblk_g * (nblk_g + 1) / 2 * 128 * 128 + args.n * dd)
        other[f'd{dd}'] = {'ms': round(best_g, 4), 'achieved': round(gb / (best_g * 1e-3) / 1e9, 1), 'frac': round(gb / (best_g * 1e-3) / 1e12 / HBM_PEAK_TBPS, 4)}
      except Exception as e:  # pylint: disable=broad-except
        other[f'd{dd}'] = {'error': str(e)[:100]}
    roofline_gram['other_feature_dims'] = other
  peak_ubench = None
  try:
    import ctypes as _C2
    tfp = _C2.c_double(0.0)
    if nat.lib().hbo_mfma_peak_probe(ctx.handle, 2.0, _C2.byref(tfp)) == 0 and tfp.value > 0:
      peak_ubench = {'tflops': round(tfp.value, 2), 'what': 'v_mfma_f64_16x16x4_f64 back to back on every SIMD for ~2 ms, measured in this run '
                                                            '(hbo_mfma_peak_probe)', 'datasheet': FP64_MFMA_PEAK_TFLOPS}
      for rf in (roofline, roofline_small, roofline_potrf, roofline_eval):
        if rf is not None:
          rf['frac_of_ubench'] = round(rf['achieved'] / tfp.value, 4)
  except Exception as e:  # pylint: disable=broad-except
    peak_ubench = {'error': str(e)[:100]}

  extra = {}
  device_info = {}
  try:
    import ctypes as _C
    nm = _C.create_string_buffer(128); cus = _C.c_int32(0); mem = _C.c_int64(0)
    ndev = max(nat.lib().hbo_device_count(), 1)
    if nat.lib().hbo_device_info(int(os.environ.get('HBO_DEVICE', '0')) % ndev, nm, 128, _C.byref(cus), _C.byref(mem)) == 0:
      device_info = {'name': nm.value.decode(), 'cus': cus.value, 'mem_gb': round(mem.value / 2**30, 1)}
  except Exception as e:  # pylint: disable=broad-except
    device_info = {'error': str(e)[:100]}

  def result_line(cpu, multitask):
    return {
        'metric': 'GP NLL+grad evals/sec at N=8192 D=16 fp64', 'value': round(value, 4), 'unit': 'evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'cfg2: single-task SE-ARD GP + constant mean, N={args.n}, D={args.d}, fp64, NLL+grad '
                               '(Gram -> blocked Cholesky -> trtri -> lauum -> gradient contraction)',
                   'parallelism': 'single GPU' if world == 1 else f'{world} independent replicas (one per GPU)',
                   'potrf_group': group},
        'algorithmic_tflops': round(float(args.n)**3 / (ms_per_step * 1e-3) / 1e12, 3),
        'stages_ms_per_step': stages,
        'stages_note': 'separate untimed pass of 3 evaluations with every stage bracketed by HIP events; the timed '
                       'region brackets only the roofline kernel (stage events cost 0.3 ms per evaluation).  Rows overlap: '
                       'syrk_bulk / syrk_trailing / trtri_early run INSIDE potrf (three streams); nll_reduce, wt_z run on the idle '
                       'panel stream BESIDE trtri / lauum (wt_z waits for the inverse, then shares the machine with lauum: its '
                       'bracket spans that) -- the sum of the rows is not the evaluation time',
        'torch_imported': 'torch' in sys.modules,
        'device': device_info,
        'roofline': roofline, 'roofline_small': roofline_small, 'roofline_potrf': roofline_potrf, 'roofline_eval': roofline_eval,
        'roofline_gram': roofline_gram, 'peak_ubench': peak_ubench, 'cpu_baseline': cpu, 'multitask': multitask, 'hgp': extra.get('hgp'),
        'jax_baseline': extra.get('jax'), 'train': extra.get('train'), 'bo_step': extra.get('bo_step'), 'fp32': extra.get('fp32'),
        'cfg3': extra.get('cfg3'), 'cfg5': extra.get('cfg5'),
    }

  # The headline is measured; the secondary legs below must not be able to lose it: if they do not finish in time
  # (a communicator that never forms, a rank that died) every rank's watchdog ends the process and rank 0 prints the
  # line without them.
  import threading
  def bail():
    if rank == 0:
      emit(result_line(None, {'error': f'secondary legs did not finish within {args.secondary_timeout} s'}))
    os._exit(0)
  watchdog = threading.Timer(args.secondary_timeout, bail)
  watchdog.daemon = True
  watchdog.start()

  # ---------------- secondary: cfg 4 multi-task objective, task-sharded ----------------------
  multitask = None
  comm = None
  if not args.no_multitask:
    data, raw4 = cfg4_inputs()
    full = {k: defs.SubDataset(xx, yy) for k, (xx, yy) in data.items()}
    mine = parallel.shard_dataset(full, rank, world)
    dev4 = objectives.DeviceDataset(mine)
    comm = None
    comm_kind = 'none'
    if world > 1:
      # the [nll, count, grad] all-reduce: libhbo's own RCCL binding (ncclAllReduce on the context's stream, xGMI),
      # unique id broadcast over the socket group
      try:
        comm = parallel.RcclComm(ctx, rank, world, pgroup.bcast_bytes)
        comm.allreduce_sum(np.zeros(4))          # first collective builds the rings now; raises if it cannot
        comm_kind = 'rccl (libhbo, xGMI)'
      except Exception as e:  # pylint: disable=broad-except
        comm = None
        comm_kind = f'host sockets (RCCL communicator unavailable: {str(e)[:120]})'
      # every rank must use the same transport: agree on whether all of them built the RCCL one
      if not all(pgroup.allgather(comm is not None)):
        if comm is not None:
          comm_kind = 'host sockets (RCCL communicator unavailable on another rank)'
        comm = parallel.SocketComm(pgroup)
    def step4(i):
      p = defs.GPParams(model=perturb(raw4, i, 0))
      return objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p, dev4, wf, comm=comm)
    for i in range(2):
      step4(-1 - i)
    sync()
    k4 = max(3, args.steps // 4)
    t0 = time.perf_counter()
    for i in range(k4):
      v4 = step4(i)
    sync()
    el4 = max_over_ranks(time.perf_counter() - t0)
    # parity inside the bench: the mean NLL of ALL 64 tasks at the unperturbed theta against the oracle's committed
    # value (tests/golden/cfg4_t64_oracle.npz) -- with task shards this goes through the all-reduce
    p0 = defs.GPParams(model=raw4)
    v_ref = objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p0, dev4, wf, comm=comm)[0]
    fx = os.path.join(ROOT, 'tests', 'golden', 'cfg4_t64_oracle.npz')
    expected = float(np.load(fx)['nll_mean']) if os.path.exists(fx) else None
    if expected is not None:
      assert abs(v_ref - expected) <= 1e-9 * abs(expected), ('cfg4 mean NLL differs from the oracle fixture', v_ref, expected)
    # every rank's own shard time (device time before the collective) and the collective itself, from the events of the
    # device-resident route (hbo_objective_sharded); host sockets: wall time of the whole call
    rank_ms = imbalance = coll_us = None
    rank_ms_source = 'device events of hbo_objective_sharded'
    if pgroup is not None:
      t_local = getattr(comm, 'last_timing', None)
      if t_local:
        all_ms = pgroup.allgather(float(t_local[0]))
        rank_ms = [round(v, 3) for v in all_ms]
        imbalance = round(max(all_ms) / (sum(all_ms) / len(all_ms)), 3)
        coll_us = round(max(pgroup.allgather(float(t_local[1]))), 1)
      else:
        # transports without the device-resident route (host sockets): each rank times its own shard without the collective
        p1 = defs.GPParams(model=perturb(raw4, 1, 0))
        local = lambda: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential, p1, dev4, wf)
        local(); sync()
        t0 = time.perf_counter()
        for _ in range(3):
          local()
        sync()
        all_ms = pgroup.allgather((time.perf_counter() - t0) / 3 * 1e3)
        rank_ms = [round(v, 3) for v in all_ms]
        imbalance = round(max(all_ms) / (sum(all_ms) / len(all_ms)), 3)
        rank_ms_source = 'host wall clock of the local shard, no collective'
    comm_us = None
    if comm is not None:
      buf = np.zeros(2 + 7)
      for _ in range(5):
        comm.allreduce_sum(buf)
      t0 = time.perf_counter()
      for _ in range(50):
        comm.allreduce_sum(buf)
      comm_us = round(max_over_ranks(time.perf_counter() - t0) / 50 * 1e6, 1)
    multitask = {'workload': 'cfg4: 64 PD1-shaped sub-datasets, N_k in [1600,2400], D=4, fp64, mean-NLL+grad, '
                             'LPT task shards + one all-reduce of [nll,count,grad]',
                 'evals_per_s': round(k4 / el4, 3), 'ms_per_eval': round(el4 / k4 * 1e3, 3), 'steps': k4,
                 'scaling': 'strong', 'comm': comm_kind, 'comm_us': comm_us, 'nll': float(v4[0]),
                 'nll_unperturbed': float(v_ref), 'nll_oracle_fixture': expected, 'local_tasks': len(mine),
                 'rank_ms': rank_ms, 'rank_ms_source': rank_ms_source if rank_ms else None, 'imbalance': imbalance,
                 'allreduce_in_eval_us': coll_us}
    if world == 1:
      # what one rank of an 8-GPU job would hold: the heaviest LPT shard of 8 (eight tasks), timed alone
      shard8 = parallel.shard_dataset(full, 0, 8)
      dev8 = objectives.DeviceDataset(shard8)
      f8 = lambda i: objectives.nll_value_and_grad(mean.constant, kernel.squared_exponential,
                                                   defs.GPParams(model=perturb(raw4, i, 0)), dev8, wf)
      f8(-1); f8(-2)
      t0 = time.perf_counter()
      for i in range(10):
        f8(i)
      multitask['shard_of_8'] = {'tasks': len(shard8), 'ms_per_eval': round((time.perf_counter() - t0) / 10 * 1e3, 3)}
      # the partition's cost model (parallel.SHARD_COST_MODEL: latency of the longest chain + throughput over n^3, measured by
      # tools/scan_cost_model.py) next to the clock: the timed shard, all eight shards as modelled, their imbalance
      sizes = {k: v.x.shape[0] for k, v in full.items()}
      parts = parallel.lpt_partition(sizes, 8)
      model_ms = [parallel.shard_cost_ms([sizes[k] for k in part]) for part in parts]
      multitask['shard_of_8'].update({'model_ms': round(parallel.shard_cost_ms([v.x.shape[0] for v in shard8.values()]), 3),
                                      'model_ms_all_shards': [round(v, 3) for v in model_ms],
                                      'model_imbalance': round(max(model_ms) / (sum(model_ms) / len(model_ms)), 4),
                                      'model': dict(parallel.SHARD_COST_MODEL, form='c0 + a * max nblk + b * sum n^3 [ms]')})
      dev8.close()
      # PROJECTION, NOT MEASURED (no multi-GPU node has run this): T = 64 on one GPU over (heaviest shard of 8 alone + the all-reduce);
      # comm_us_assumed = 30 us for one ~200-byte ncclAllReduce over xGMI (an assumption: the xGMI transport has never been exercised)
      comm_assumed_us = 30.0
      multitask['projected_8gpu'] = {'label': 'projection, not measured', 'speedup': round(multitask['ms_per_eval'] / (multitask['shard_of_8']['ms_per_eval'] + comm_assumed_us * 1e-3), 3),
                                     'formula': 'T64_ms / (shard_ms + comm_us / 1000)', 'T64_ms': multitask['ms_per_eval'],
                                     'shard_ms': multitask['shard_of_8']['ms_per_eval'], 'comm_us_assumed': comm_assumed_us}
    dev4.close()

  # ---------------- cfg 3 and cfg 5 (rank 0, N=1 only; driver-timed instead of builder-tool numbers) ----------
  if rank == 0 and world == 1 and not args.no_extra:
    try:
      extra['cfg3'] = bench_cfg3(ctx)
      extra['cfg5'] = bench_cfg5(ctx)
      extra['train'] = bench_train()
      extra['bo_step'] = bench_bo_step()
      extra['fp32'] = bench_fp32_objective(ctx)
      extra['hgp'] = bench_hgp()
    except Exception as e:  # pylint: disable=broad-except
      extra['error'] = str(e)[:200]

  # ---------------- CPU baseline (rank 0, N=1 only) -------------------------------------------
  cpu = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import cpu_baseline
    eff_cpus, cpu_quota = cpu_baseline.effective_cpus()
    cpu_baseline.set_threads(eff_cpus)   # (256 spinning OpenMP threads under a cgroup quota of a few CPUs run at the speed of one)
    # two forms of the same port: (a) potrf / potri as ONE LAPACK call each on SciPy's OpenBLAS pool (its build stops at 64 threads),
    # (b) tile algorithms over OpenMP on every core, one single-threaded BLAS call per tile (oracle/cpu_port.c).  The faster one is the
    # baseline; both are reported.
    def time_form(fn, evals, **kw):
      t0 = time.perf_counter()
      vals = [fn(x, y, perturb(raw, i, 0), **kw)[0] for i in range(evals)]
      return evals / (time.perf_counter() - t0), vals
    forms = {}
    tile_timings = []
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=max(1, min(64, eff_cpus)), user_api='blas'):
      rate_l, vals_l = time_form(cpu_baseline.nll_and_grad_se_ard_constant_omp, max(2, args.cpu_evals // 4))
    forms['lapack_pool'] = round(rate_l, 4)
    nb_best, rate_best = 128, 0.0
    for nb in (128, 192, 256):   # one evaluation each, then the rest on the best tile size -- if that form is the faster one at all
      r1, _ = time_form(cpu_baseline.nll_and_grad_se_ard_constant_tiled, 1, nb=nb)
      forms[f'tiled_nb{nb}_1eval'] = round(r1, 4)
      if r1 > rate_best:
        nb_best, rate_best = nb, r1
    rate_t, vals_t = rate_best, None
    if rate_best > 0.8 * rate_l:
      rate_t, vals_t = time_form(cpu_baseline.nll_and_grad_se_ard_constant_tiled, args.cpu_evals, nb=nb_best, timings=tile_timings)
      forms['tiled'] = round(rate_t, 4)
    tiled_wins = vals_t is not None and rate_t >= rate_l
    rate, vals = (rate_t, vals_t) if tiled_wins else (rate_l, vals_l)
    omp_thr = cpu_baseline.omp_threads()
    cpu = {'value': round(rate, 4), 'unit': 'evals/s', 'cores': eff_cpus, 'cores_visible': os.cpu_count(), 'cgroup_cpu_quota': cpu_quota,
           'threads_used': omp_thr if tiled_wins else min(eff_cpus, 64), 'kind': 'port',
           'form': (f'tiled potrf / trtri / lauum, nb = {nb_best}, {omp_thr} OpenMP threads x single-threaded OpenBLAS tile calls' if tiled_wins
                    else 'LAPACK potrf / potri on the OpenBLAS thread pool'),
           'forms_evals_per_s': forms,
           'sample': f'{len(vals)} NLL+grad evaluations of the same N={args.n}, D={args.d} fp64 workload '
                     f'(oracle/cpu_baseline.py + oracle/cpu_port.c: Gram build, gradient contraction and -- in the tiled form -- potrf / '
                     f'trtri / lauum in C/OpenMP on all cores; the other form calls LAPACK potrf / potrs / potri via SciPy / OpenBLAS)',
           'seconds': round(len(vals) / rate, 2), 'nll_matches_gpu': bool(abs(vals[-1] - float(step_fn(len(vals) - 1)[0])) <= 1e-8 * abs(vals[-1]))}
    if tile_timings:
      cpu['tiled_stage_seconds'] = {k_: round(float(np.mean([t_[k_] for t_ in tile_timings])), 4) for k_ in tile_timings[0]}
    cpu['tflops'] = round(cpu['value'] * float(args.n)**3 / 1e12, 4)
    cpu.update(cpu_provenance())
    caps = [t.get('num_threads') for t in cpu.get('threadpools', []) if t.get('user_api') == 'blas' and t.get('num_threads')]
    cpu['blas_threads_cap'] = min(caps) if caps else None
    cpu['note'] = ('a stated baseline, not a target: the faster of the two forms above; blas_threads_cap is the limit of the OpenBLAS build that the '
                   'LAPACK-pool form runs into, threads_used what the tiled form ran on; the GPU / CPU ratio says nothing about kernel quality')
    if args.jax:
      try:
        import jax  # noqa: F401  pylint: disable=unused-import
        from oracle import jax_baseline
        extra['jax'] = jax_baseline.time_nll_and_grad(x, y, perturb(raw, 0, 0), budget_s=20.0)
      except ImportError:
        extra['jax'] = {'error': 'jax is not importable on this box'}
    # the GPU again, after the CPU leg (a utilisation sampler that only sees the end of the run finds it busy here)
    t0 = time.perf_counter()
    for i in range(10):
      step_fn(30_000 + i)
    device_info['ms_per_step_after_cpu_leg'] = round((time.perf_counter() - t0) / 10 * 1e3, 4)

  watchdog.cancel()
  if rank == 0:
    emit(result_line(cpu, multitask))
  dev.close()
  if pgroup is not None:
    pgroup.barrier()
    if comm is not None and hasattr(comm, 'close'):
      comm.close()
    pgroup.close()


if __name__ == '__main__':
  sys.exit(main() or 0)
