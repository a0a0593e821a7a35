"""world_size-2 gloo test of the task-sharded objective's reduction path (CPU, no GPU):
each rank evaluates ITS shard with the oracle standing in for the device call, the
[nll_sum, count, grad_sum] buffer goes through helpers.TorchDistComm (gloo; test-only), and the result must
equal the single-process mean over all tasks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch.distributed as dist
  import helpers
  from hyperbo_amd import parallel
  from hyperbo_amd.basics import definitions as defs
  from oracle import hyperbo_oracle as o
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  rng = np.random.default_rng(0)
  model = helpers.make_model(rng, 'constant', False, 2)
  sizes = [9, 14, 6, 11, 8]
  full = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, 2)) for i, n in enumerate(sizes)}
  full[99] = defs.SubDataset(*helpers.synthetic_task(rng, 4, 2, m=2), aligned=1)   # skipped
  mine = parallel.shard_dataset(full, rank, world)
  params = o.GPParams(model=model)
  nll_sum, grad_sum = 0.0, np.zeros(helpers.flatten(model).size)
  for k, s in mine.items():
    v, g = o.nll_sub_dataset_value_and_grad(o.constant, o.squared_exponential, params, s.x, s.y, o.DEFAULT_WARP_FUNC)
    nll_sum += v; grad_sum += helpers.flatten(g)
  comm = helpers.TorchDistComm()
  value, grad, count = parallel.sharded_mean_nll(nll_sum, len(mine), grad_sum, comm)
  np.savez(os.path.join(out_dir, f'r{rank}.npz'), value=value, grad=grad, count=count, keys=np.array(sorted(mine)))
  dist.barrier()
  dist.destroy_process_group()


def test_sharded_objective_matches_single_process(tmp_path):
  torch = pytest.importorskip('torch')
  import torch.multiprocessing as mp
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers
  from oracle import hyperbo_oracle as o
  port = _free_port()
  mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
  r0, r1 = np.load(tmp_path / 'r0.npz'), np.load(tmp_path / 'r1.npz')
  assert int(r0['count']) == int(r1['count']) == 5
  assert not set(r0['keys']) & set(r1['keys']) and len(r0['keys']) + len(r1['keys']) == 5
  rng = np.random.default_rng(0)
  model = helpers.make_model(rng, 'constant', False, 2)
  full = {i: o.SubDataset(*helpers.synthetic_task(rng, n, 2)) for i, n in enumerate([9, 14, 6, 11, 8])}
  full[99] = o.SubDataset(*helpers.synthetic_task(rng, 4, 2, m=2), aligned=1)
  val, g = o.nll_value_and_grad(o.constant, o.squared_exponential, o.GPParams(model=model), full, o.DEFAULT_WARP_FUNC)
  for r in (r0, r1):
    assert abs(float(r['value']) - val) <= 1e-12 * abs(val)
    np.testing.assert_allclose(r['grad'], helpers.flatten(g), rtol=1e-11, atol=1e-12)


# ---- the torch-free process group (hyperbo_amd.parallel.SocketGroup / SocketComm) --------------------------------
def _socket_worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers
  from hyperbo_amd import parallel
  from hyperbo_amd.basics import definitions as defs
  from oracle import hyperbo_oracle as o
  assert 'torch' not in sys.modules
  group = parallel.SocketGroup(rank, world, port, scan=8, token='unit-test')
  uid = group.bcast_bytes(bytes(range(128)) if rank == 0 else b'')
  group.barrier()
  mx = group.allreduce_max(10.0 - rank)
  rng = np.random.default_rng(0)
  model = helpers.make_model(rng, 'constant', False, 2)
  sizes = [9, 14]                      # two tasks, three ranks: the last rank's shard is empty
  full = {i: defs.SubDataset(*helpers.synthetic_task(rng, n, 2)) for i, n in enumerate(sizes)}
  mine = parallel.shard_dataset(full, rank, world)
  params = o.GPParams(model=model)
  nll_sum, grad_sum = 0.0, np.zeros(helpers.flatten(model).size)
  for k, s in mine.items():
    v, g = o.nll_sub_dataset_value_and_grad(o.constant, o.squared_exponential, params, s.x, s.y, o.DEFAULT_WARP_FUNC)
    nll_sum += v; grad_sum += helpers.flatten(g)
  value, grad, count = parallel.sharded_mean_nll(nll_sum, len(mine), grad_sum, parallel.SocketComm(group))
  np.savez(os.path.join(out_dir, f's{rank}.npz'), value=value, grad=grad, count=count, ntasks=len(mine), mx=mx,
           uid_ok=(uid == bytes(range(128))), torch_loaded=('torch' in sys.modules))
  group.barrier()
  group.close()


def test_socket_group_sharded_objective_three_ranks_two_tasks(tmp_path):
  """world_size 3 over localhost sockets, no torch: rendezvous past an occupied port, byte broadcast, max, and the
  [nll, count, grad] sum with one EMPTY shard must equal the single-process mean over the two tasks."""
  import multiprocessing as mp
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import helpers
  from oracle import hyperbo_oracle as o
  blocker = socket.socket(); blocker.bind(('127.0.0.1', 0)); port = blocker.getsockname()[1]   # hub must skip this one
  ctx = mp.get_context('spawn')
  procs = [ctx.Process(target=_socket_worker, args=(r, 3, port, str(tmp_path))) for r in range(3)]
  [p.start() for p in procs]
  [p.join(120) for p in procs]
  blocker.close()
  assert all(p.exitcode == 0 for p in procs)
  res = [np.load(tmp_path / f's{r}.npz') for r in range(3)]
  assert sorted(int(r['ntasks']) for r in res) == [0, 1, 1]
  rng = np.random.default_rng(0)
  model = helpers.make_model(rng, 'constant', False, 2)
  full = {i: o.SubDataset(*helpers.synthetic_task(rng, n, 2)) for i, n in enumerate([9, 14])}
  val, g = o.nll_value_and_grad(o.constant, o.squared_exponential, o.GPParams(model=model), full, o.DEFAULT_WARP_FUNC)
  for r in res:
    assert int(r['count']) == 2 and float(r['mx']) == 10.0 and bool(r['uid_ok']) and not bool(r['torch_loaded'])
    assert abs(float(r['value']) - val) <= 1e-12 * abs(val)
    np.testing.assert_allclose(r['grad'], helpers.flatten(g), rtol=1e-11, atol=1e-12)
  assert np.array_equal(res[0]['grad'], res[1]['grad']) and np.array_equal(res[0]['grad'], res[2]['grad'])   # rank-order sum: bitwise equal


# ---- SocketGroup: typed frames, mutual token proof, no pickle ------------------------------------------------------
def test_wire_format_round_trip_and_rejects_garbage():
  sys.path.insert(0, ROOT)
  from hyperbo_amd import parallel
  import inspect
  assert 'import pickle' not in inspect.getsource(parallel) and 'pickle.loads' not in inspect.getsource(parallel)
  msg = [None, True, False, 3, -7, 2.5, 'id', bytes(range(40)), np.arange(5, dtype=np.float64), [1, [2.0, b'x']]]
  out, end = parallel._decode(parallel._encode(msg))
  assert end == len(parallel._encode(msg))
  assert out[:8] == msg[:8] and np.array_equal(out[8], msg[8]) and out[9] == [1, [2.0, b'x']]
  for bad in (b'', b'XX' + b'\0' * 16, b'HBZ\x00\x00\x00\x00', b'HBB\xff\xff\xff\x7f', b'HBD\x03\x00\x00\x00abc'):
    with pytest.raises(Exception):
      parallel._decode(bad)
  with pytest.raises(TypeError):
    parallel._encode(object())


def _hub_proc(port, token, out_path):
  sys.path.insert(0, ROOT)
  from hyperbo_amd import parallel
  g = parallel.SocketGroup(0, 2, port, token=token, timeout=30)
  res = g.allgather('hub')
  g.close()
  open(out_path, 'w').write(repr(res))


def test_socket_group_ignores_strangers_and_wrong_tokens(tmp_path):
  """A connection that sends garbage, one that holds the wrong token and one that claims rank 0 are dropped; the
  legitimate rank still joins, and a second claim of its rank does not replace it."""
  import multiprocessing as mp
  import struct
  sys.path.insert(0, ROOT)
  from hyperbo_amd import parallel
  port = _free_port()
  out = tmp_path / 'hub.txt'
  ctx = mp.get_context('spawn')
  hub = ctx.Process(target=_hub_proc, args=(port, 'secret', str(out)))
  hub.start()
  import time
  def connect():
    for _ in range(100):
      try:
        return socket.create_connection(('127.0.0.1', port), timeout=2.0)
      except OSError:
        time.sleep(0.05)
    raise RuntimeError('hub did not come up')
  # 1. garbage (what a pickle-based peer would have sent: 8-byte length + payload)
  s = connect(); s.recv(4096); s.sendall(struct.pack('<Q', 1 << 40) + b'\x80\x04junk'); s.close()
  # 2. well-formed greeting with the wrong token
  s = connect(); hello = parallel._recv_msg(s)
  parallel._send_msg(s, [parallel._MAGIC, 1, b'n' * 16, parallel._mac('wrong', b'client', hello[2], b'1')])
  s.settimeout(2.0)
  try:
    assert s.recv(64) == b''       # the hub closed the connection without answering
  except (ConnectionError, socket.timeout):
    pass
  s.close()
  # 3. right token, rank out of range
  s = connect(); hello = parallel._recv_msg(s)
  parallel._send_msg(s, [parallel._MAGIC, 0, b'n' * 16, parallel._mac('secret', b'client', hello[2], b'0')]); s.close()
  # 4. a client with the wrong token never accepts the hub either
  with pytest.raises(TimeoutError):
    parallel.SocketGroup(1, 2, port, token='wrong', timeout=1.5)
  # 5. the legitimate rank 1
  g = parallel.SocketGroup(1, 2, port, token='secret', timeout=30)
  assert g.allgather('one') == ['hub', 'one']
  g.close()
  hub.join(30)
  assert hub.exitcode == 0 and out.read_text() == "['hub', 'one']"
