"""bench.py -- headline benchmark of the MI355X-native HyperBO GP hot path.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

metric (BASELINE.json): GP NLL+grad evaluations/sec at N=8192, D=16, fp64  (configs[1]).
A "step" is one NLL+gradient evaluation of a single-task SE-ARD GP (X, y resident in HBM; only the
hyper-parameters change between evaluations, as in the reference's training loop gp.py:132-144).
At N>1 GPUs a single factorisation does not shard ("replicas only"): every rank runs
its own evaluation stream (different theta -- line-search points / restarts) and `value` is the
aggregate.  The task-sharded multi-task objective (configs[3]: 64 PD1-shaped sub-datasets, one
RCCL all-reduce of [nll, grad] per evaluation) is timed in the same run and reported under
"multitask".

Extra objects on the JSON line:
  roofline     -- the Cholesky trailing-update kernel (fp64 MFMA syrk): algorithmic flops per launch
                  / HIP-event duration of those launches inside the timed region.
  cpu_baseline -- oracle/cpu_baseline.py (NumPy/SciPy-LAPACK port, kind "port") on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet; confirmed 77.1 by tools/mfma_probe.hip (64 cyc/instr)


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
  m(m+1)/2 >= 600 (hyperbo_amd/csrc/api.hip run_potrf).  Lower triangle incl. diagonal tiles, K = 128*group."""
  nblk = (n + 127) // 128
  out = []
  for g0 in range(0, nblk, group):
    g1 = min(g0 + group, nblk)
    g2 = min(g1 + group, nblk)
    m = nblk - g2
    if g1 < nblk and g2 < nblk and m * (m + 1) // 2 >= 600:
      out.append(m * (m + 1) / 2 * 128 * 128 * 2.0 * 128 * (g1 - g0))
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--n', type=int, default=8192)
  ap.add_argument('--d', type=int, default=16)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-multitask', action='store_true')
  ap.add_argument('--cpu-evals', type=int, default=8)
  ap.add_argument('--secondary-timeout', type=float, default=420.0, help='seconds for the multitask + CPU legs')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  os.environ.setdefault('HBO_DEVICE', str(local_rank))
  dist = None
  torch = None
  if world > 1:
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)   # rendezvous/barriers only
    if torch.cuda.is_available():
      torch.cuda.set_device(local_rank % torch.cuda.device_count())

  from hyperbo_amd import _native as nat
  from hyperbo_amd import parallel
  from hyperbo_amd.basics import definitions as defs
  from hyperbo_amd.gp_utils import kernel, mean, objectives, utils

  ctx = nat.default_context()
  wf = utils.DEFAULT_WARP_FUNC
  # panels per trailing update: set explicitly (libhbo's own choice for <= 96 blocks is also 3, K = 384) because the
  # roofline's algorithmic flops per launch below are computed from it
  potrf_group = int(os.environ.get('HBO_BENCH_POTRF_GROUP', '3'))
  ctx.set_option('potrf_group', potrf_group)

  def sync():
    if torch is not None and torch.cuda.is_available():
      torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()

  def max_over_ranks(t):
    if dist is None:
      return t
    tt = torch.tensor([t], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())

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
  assert np.isfinite(last[0]), 'NLL is not finite'
  ms_per_step = elapsed / args.steps * 1e3
  value = world * args.steps / elapsed

  group = potrf_group
  fl = bulk_update_flops(args.n, group)
  roofline = None
  if 'syrk_bulk' in prof and prof['syrk_bulk'][1] > 0 and fl:
    tot_ms, launches = prof['syrk_bulk']
    assert launches == len(fl) * args.steps, (launches, len(fl), args.steps)
    flops_total = sum(fl) * args.steps
    achieved = flops_total / (tot_ms * 1e-3) / 1e12
    roofline = {'bound': 'mfma', 'kernel': f'gemm_kernel<double,true,true,128> (bulk Cholesky trailing update, syrk K={128 * group})',
                'achieved': round(achieved, 3), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                'launches': launches, 'avg_launch_ms': round(tot_ms / launches, 4),
                'algorithmic_gflop_per_launch': round(sum(fl) / len(fl) / 1e9, 3),
                'note': 'HIP events around each launch on its own stream inside the timed region; the launches '
                        'overlap with the look-ahead panel chain and the early triangular inverse on other streams'}
  if roofline is not None:
    # HBM traffic per launch from committed rocprofv3 PMC passes of this same command (separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md: on gfx950 it
    # reports half the bytes of wide coalesced streaming reads).  bench.py cannot run rocprof itself.
    pmc_path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm.json')
    if os.path.exists(pmc_path):
      pmc = json.load(open(pmc_path)).get('gemm_kernel<double, true, true, 128>')
      if pmc:
        roofline['traffic'] = int((2 * pmc['FETCH_SIZE_KB'] + pmc['WRITE_SIZE_KB']) * 1024)
        roofline['traffic_note'] = 'bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, profiles/r01_pmc_hbm.json'
  stages = {k: round(v[0] / stage_evals, 4) for k, v in stage_prof.items()}   # separate pass with all stage events on
  ctx.profile_enable(0)

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
                       'region brackets only the roofline kernel (stage events cost 0.3 ms per evaluation)',
        'roofline': roofline, 'cpu_baseline': cpu, 'multitask': multitask,
    }

  # The headline is measured; the secondary legs below must not be able to lose it: if they do not finish in time
  # (a communicator that never forms, a rank that died) every rank's watchdog ends the process and rank 0 prints the
  # line without them.
  import threading
  def bail():
    if rank == 0:
      print(json.dumps(result_line(None, {'error': f'secondary legs did not finish within {args.secondary_timeout} s'})), flush=True)
    os._exit(0)
  watchdog = threading.Timer(args.secondary_timeout, bail)
  watchdog.daemon = True
  watchdog.start()

  # ---------------- secondary: cfg 4 multi-task objective, task-sharded ----------------------
  multitask = None
  if not args.no_multitask:
    data, raw4 = cfg4_inputs()
    full = {k: defs.SubDataset(xx, yy) for k, (xx, yy) in data.items()}
    mine = parallel.shard_dataset(full, rank, world)
    dev4 = objectives.DeviceDataset(mine)
    comm = None
    comm_kind = 'none'
    if world > 1:
      def bcast(b):
        obj = [b]
        dist.broadcast_object_list(obj, src=0)
        return obj[0]
      # the [nll, count, grad] all-reduce: RCCL through torch.distributed's own 'nccl' backend (the well-trodden path
      # on ROCm; HBO_BENCH_COMM=libhbo selects libhbo's direct RCCL binding), gloo if no communicator can be built
      pref = os.environ.get('HBO_BENCH_COMM', 'torch-nccl')
      try:
        if pref == 'libhbo':
          comm = parallel.RcclComm(ctx, rank, world, bcast)
          comm_kind = 'rccl (libhbo, xGMI)'
        else:
          import datetime
          grp = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=120))
          comm = parallel.TorchDistComm(device=f'cuda:{local_rank % torch.cuda.device_count()}', group=grp)
          comm.allreduce_sum(np.zeros(4))          # builds the communicator now; raises if it cannot
          comm_kind = 'torch.distributed nccl (RCCL over xGMI)'
      except Exception as e:  # pylint: disable=broad-except
        comm = None
        comm_kind = f'torch.distributed gloo (RCCL communicator unavailable: {str(e)[:120]})'
      # every rank must use the same transport: agree (over gloo) on whether all of them built the RCCL one
      ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64)
      dist.all_reduce(ok, op=dist.ReduceOp.MIN)
      if float(ok.item()) < 0.5:
        if comm is not None:
          comm_kind = 'torch.distributed gloo (RCCL communicator unavailable on another rank)'
        comm = parallel.TorchDistComm()
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
    multitask = {'workload': 'cfg4: 64 PD1-shaped sub-datasets, N_k in [1600,2400], D=4, fp64, mean-NLL+grad, '
                             'LPT task shards + one all-reduce of [nll,count,grad]',
                 'evals_per_s': round(k4 / el4, 3), 'ms_per_eval': round(el4 / k4 * 1e3, 3), 'steps': k4,
                 'scaling': 'strong', 'comm': comm_kind, 'nll': float(v4[0]), 'local_tasks': len(mine)}
    dev4.close()

  # ---------------- CPU baseline (rank 0, N=1 only) -------------------------------------------
  cpu = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import cpu_baseline
    t0 = time.perf_counter()
    vals = []
    for i in range(args.cpu_evals):
      v, _ = cpu_baseline.nll_and_grad_se_ard_constant_omp(x, y, perturb(raw, i, 0))
      vals.append(v)
    el = time.perf_counter() - t0
    cpu = {'value': round(args.cpu_evals / el, 4), 'unit': 'evals/s', 'cores': os.cpu_count(), 'kind': 'port',
           'sample': f'{args.cpu_evals} NLL+grad evaluations of the same N={args.n}, D={args.d} fp64 workload '
                     f'(oracle/cpu_baseline.py + oracle/cpu_port.c: Gram build and gradient contraction in C/OpenMP on all '
                     f'cores, LAPACK potrf/potrs/potri via SciPy/OpenBLAS)',
           'seconds': round(el, 2), 'nll_matches_gpu': bool(abs(vals[-1] - float(step_fn(args.cpu_evals - 1)[0])) <= 1e-8 * abs(vals[-1]))}

  watchdog.cancel()
  if rank == 0:
    print(json.dumps(result_line(cpu, multitask)), flush=True)
  dev.close()
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
