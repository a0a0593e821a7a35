"""Task sharding of the multi-task objective over the GPUs of one node (one process per GPU).

The multi-task NLL (hyperbo/gp_utils/objectives.py:181-195) is a mean over independent
sub-datasets, so tasks are partitioned statically (longest-processing-time first on n^3) and the
only exchange is ONE sum-all-reduce of [nll_sum, n_tasks, grad_sum] (<= ~10 KB) per evaluation.
Communicators:
  RcclComm      -- libhbo's RCCL binding (hbo_comm_*), device buffers over xGMI; the unique id is
                   distributed by any bootstrap callable (SocketGroup.bcast_bytes below -- no torch).
  SocketComm    -- host-side sum through the SocketGroup hub (tests, and the fallback when no RCCL
                   communicator can be built, e.g. two ranks sharing one GPU).
  LocalComm     -- single process.
SocketGroup is the torch-free process group of one node (rendezvous, barrier, max, byte broadcast over
127.0.0.1): the product path needs neither torch nor an MPI launcher.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import hmac
import os
import socket
import struct
import time
from typing import Callable, Dict, Hashable, List, Sequence

import numpy as np


# Measured cost of one rank's shard on an MI355X (tools/scan_cost_model.py, fp64 SE-ARD D = 4, batches of 2-16 tasks of 1024-2560
# points, profiles/r05_shard_cost_model.txt):  T [ms] = c0 + a * max_k nblk_k + b * sum_k n_k^3  (nblk = ceil(n / 128)).
# The chain of dependent panel steps is as long as the LARGEST task of the batch (the batch moves in lockstep: a), everything
# else is throughput (b = 1 / 50 TFLOP/s over n^3).  rms residual 5 %, worst 14 %.
SHARD_COST_MODEL = {'c0': 0.363, 'a': 0.0362, 'b': 2.016e-11}


def shard_cost_ms(sizes: Sequence[int], model: Dict[str, float] = None) -> float:
  """Modelled milliseconds of one NLL + gradient evaluation of a shard holding tasks of these sizes (0 for an empty shard)."""
  m = model or SHARD_COST_MODEL
  sizes = [int(n) for n in sizes if int(n) > 0]
  if not sizes:
    return 0.0
  return m['c0'] + m['a'] * max(-(-n // 128) for n in sizes) + m['b'] * sum(float(n)**3 for n in sizes)


def lpt_partition(sizes: Dict[Hashable, int], num_shards: int, power: float = 3.0, model: Dict[str, float] = None) -> List[List[Hashable]]:
  """Greedy longest-processing-time assignment of tasks to shards; deterministic.  Tasks in descending n^power; each goes to the
  shard whose MODELLED time after taking it (shard_cost_ms: latency of the longest chain + throughput over n^3) is the smallest --
  with tasks of similar block counts this is the classic LPT on n^3, and a shard that already holds a long chain is the cheaper
  home for the next long task."""
  if num_shards <= 0:
    raise ValueError('num_shards must be positive')
  order = sorted(sizes.items(), key=lambda kv: (-float(kv[1])**power, str(kv[0])))
  held: List[List[int]] = [[] for _ in range(num_shards)]
  shards: List[List[Hashable]] = [[] for _ in range(num_shards)]
  for key, n in order:
    s = min(range(num_shards), key=lambda i: (shard_cost_ms(held[i] + [n], model), i))
    shards[s].append(key)
    held[s].append(int(n))
  return shards


def shard_dataset(dataset, rank: int, world_size: int, exclude_aligned: bool = True):
  """Returns this rank's sub-dict of `dataset` (same selection rule as objectives.py:181-185)."""
  from hyperbo_amd.gp_utils import objectives
  items = objectives.included_sub_datasets(dataset, exclude_aligned)
  sizes = {k: s.x.shape[0] for k, s in items}
  mine = set(lpt_partition(sizes, world_size)[rank])
  return {k: s for k, s in items if k in mine}


_MAGIC = b'HBOGRP2\n'
_MAX_FRAME = 16 << 20   # bytes; the largest legitimate message is a gradient vector or a 128-byte RCCL id


# Wire format: typed, length-capped frames -- nothing that arrives on the socket is ever unpickled or evaluated.
#   frame = b'HB' + type (1 byte) + payload length (u32, little endian) + payload
#   N none | T true | F false | D float64 | I int64 | S utf-8 string | B bytes | A float64 vector | L list of frames
def _encode(obj) -> bytes:
  if obj is None:
    t, payload = b'N', b''
  elif obj is True:
    t, payload = b'T', b''
  elif obj is False:
    t, payload = b'F', b''
  elif isinstance(obj, (bytes, bytearray)):
    t, payload = b'B', bytes(obj)
  elif isinstance(obj, str):
    t, payload = b'S', obj.encode('utf-8')
  elif isinstance(obj, (int, np.integer)):
    t, payload = b'I', struct.pack('<q', int(obj))
  elif isinstance(obj, (float, np.floating)):
    t, payload = b'D', struct.pack('<d', float(obj))
  elif isinstance(obj, np.ndarray):
    t, payload = b'A', np.ascontiguousarray(obj, dtype=np.float64).ravel().tobytes()
  elif isinstance(obj, (list, tuple)):
    t, payload = b'L', b''.join(_encode(o) for o in obj)
  else:
    raise TypeError(f'SocketGroup cannot send a {type(obj).__name__}')
  if len(payload) > _MAX_FRAME:
    raise ValueError('SocketGroup: message too large')
  return b'HB' + t + struct.pack('<I', len(payload)) + payload


def _decode(buf: bytes, pos: int = 0):
  if buf[pos:pos + 2] != b'HB' or len(buf) < pos + 7:
    raise ValueError('SocketGroup: malformed frame')
  t = buf[pos + 2:pos + 3]
  n, = struct.unpack_from('<I', buf, pos + 3)
  body = buf[pos + 7:pos + 7 + n]
  if n > _MAX_FRAME or len(body) != n:
    raise ValueError('SocketGroup: malformed frame')
  end = pos + 7 + n
  if t == b'N':
    return None, end
  if t == b'T':
    return True, end
  if t == b'F':
    return False, end
  if t == b'B':
    return bytes(body), end
  if t == b'S':
    return body.decode('utf-8'), end
  if t == b'I' and n == 8:
    return struct.unpack('<q', body)[0], end
  if t == b'D' and n == 8:
    return struct.unpack('<d', body)[0], end
  if t == b'A' and n % 8 == 0:
    return np.frombuffer(body, dtype=np.float64).copy(), end
  if t == b'L':
    out, p = [], 0
    while p < n:
      o, p = _decode(body, p)
      out.append(o)
    return out, end
  raise ValueError('SocketGroup: unknown frame type')


def _send_msg(sock, obj):
  sock.sendall(_encode(obj))


def _recv_exact(sock, n):
  buf = bytearray()
  while len(buf) < n:
    chunk = sock.recv(n - len(buf))
    if not chunk:
      raise ConnectionError('SocketGroup: peer closed the connection')
    buf += chunk
  return bytes(buf)


def _recv_msg(sock):
  head = _recv_exact(sock, 7)
  if head[:2] != b'HB':
    raise ValueError('SocketGroup: malformed frame')
  n, = struct.unpack('<I', head[3:7])
  if n > _MAX_FRAME:
    raise ValueError('SocketGroup: frame exceeds the size cap')
  return _decode(head + _recv_exact(sock, n))[0]


def _mac(token: str, *parts: bytes) -> bytes:
  return hmac.new(token.encode('utf-8'), b'|'.join(parts), hashlib.sha256).digest()


class SocketGroup:
  """Process group of the ranks of ONE node over loopback TCP (always 127.0.0.1): rank 0 is the hub.

  `port` is where rank 0 listens; with `scan` > 0 rank 0 takes the first free port in [port, port + scan) and the
  other ranks probe that range for the hub -- so the group can be derived from a launcher's MASTER_PORT (taken by the
  launcher itself) without a second agreed port.  Both sides prove that they hold `token` before anything else is
  exchanged (HMAC-SHA256 over fresh nonces, in both directions); a rank that is already registered is not replaced.
  Messages are typed, length-capped frames (None / bool / int / float / str / bytes / float64 vectors / lists of those):
  nothing received is unpickled.  Collectives are hub-and-spoke exchanges (latency ~0.1 ms): rendezvous, barrier,
  max / gather of scalars, broadcast of the 128-byte RCCL id.  The [nll, count, grad] all-reduce itself goes over RCCL."""

  def __init__(self, rank: int, world_size: int, port: int, addr: str = '127.0.0.1', scan: int = 0, token: str = '',
               timeout: float = 120.0):
    del addr   # single-node by definition: never listen on, or connect to, anything but loopback
    addr = '127.0.0.1'
    self.rank, self.world_size = int(rank), int(world_size)
    self._peers: List[socket.socket] = []
    self._hub = None
    deadline = time.time() + timeout
    ports = list(range(port, port + max(scan, 1)))
    if self.rank == 0:
      srv = None
      for p in ports:
        try:
          srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
          srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
          srv.bind((addr, p))
          break
        except OSError:
          srv.close(); srv = None
      if srv is None:
        raise RuntimeError(f'SocketGroup: no free port in {ports[0]}..{ports[-1]}')
      srv.listen(self.world_size + 8)
      srv.settimeout(1.0)
      by_rank = {}
      while len(by_rank) < self.world_size - 1:
        if time.time() > deadline:
          raise TimeoutError(f'SocketGroup: {len(by_rank) + 1} of {self.world_size} ranks arrived')
        try:
          conn, _ = srv.accept()
        except socket.timeout:
          continue
        try:
          conn.settimeout(10.0)
          nonce_h = os.urandom(16)
          _send_msg(conn, [_MAGIC, self.world_size, nonce_h])
          r = _recv_msg(conn)
          if not (isinstance(r, list) and len(r) == 4 and r[0] == _MAGIC and isinstance(r[1], int) and isinstance(r[2], bytes)
                  and isinstance(r[3], bytes) and 0 < r[1] < self.world_size):
            raise ValueError('bad greeting')
          if not hmac.compare_digest(r[3], _mac(token, b'client', nonce_h, str(r[1]).encode())):
            raise ValueError('bad token')
          if r[1] in by_rank:
            raise ValueError('rank already registered')   # keep the first one
          _send_msg(conn, _mac(token, b'hub', r[2]))
          conn.settimeout(timeout)
          conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
          by_rank[r[1]] = conn
        except Exception:  # pylint: disable=broad-except   (a stray connection: not one of ours)
          conn.close()
      srv.close()
      self._peers = [by_rank[r] for r in range(1, self.world_size)]
    else:
      while self._hub is None:
        if time.time() > deadline:
          raise TimeoutError('SocketGroup: hub (rank 0) not found')
        for p in ports:
          try:
            c = socket.create_connection((addr, p), timeout=2.0)
            c.settimeout(5.0)
            h = _recv_msg(c)
            if not (isinstance(h, list) and len(h) == 3 and h[0] == _MAGIC and h[1] == self.world_size and isinstance(h[2], bytes)):
              c.close(); continue
            nonce_c = os.urandom(16)
            _send_msg(c, [_MAGIC, self.rank, nonce_c, _mac(token, b'client', h[2], str(self.rank).encode())])
            proof = _recv_msg(c)
            if not (isinstance(proof, bytes) and hmac.compare_digest(proof, _mac(token, b'hub', nonce_c))):
              c.close(); continue   # somebody else's hub (or ours refused us)
            c.settimeout(timeout)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._hub = c
            break
          except Exception:  # pylint: disable=broad-except   (nobody / somebody else on that port)
            continue
        if self._hub is None:
          time.sleep(0.05)

  def allgather(self, obj):
    """Every rank's `obj`, in rank order, on every rank."""
    if self.world_size == 1:
      return [obj]
    if self.rank == 0:
      out = [obj] + [_recv_msg(c) for c in self._peers]
      for c in self._peers:
        _send_msg(c, out)
      return out
    _send_msg(self._hub, obj)
    return _recv_msg(self._hub)

  def barrier(self):
    self.allgather(None)

  def allreduce_max(self, value: float) -> float:
    return float(max(self.allgather(float(value))))

  def bcast_bytes(self, data: bytes) -> bytes:
    """Rank 0's bytes on every rank (the RCCL unique id)."""
    return self.allgather(data if self.rank == 0 else None)[0]

  def close(self):
    for c in self._peers + ([self._hub] if self._hub is not None else []):
      try:
        c.close()
      except OSError:
        pass
    self._peers, self._hub = [], None


class SocketComm:
  """Sum-all-reduce of a small float64 buffer through the SocketGroup hub (host memory, no GPU involved)."""

  def __init__(self, group: SocketGroup):
    self.group = group
    self.rank, self.world_size = group.rank, group.world_size

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    parts = self.group.allgather(np.ascontiguousarray(buf, dtype=np.float64))
    out = np.zeros_like(parts[0])
    for p in parts:          # rank order on every rank: bitwise identical results everywhere
      out = out + p
    return out


class LocalComm:
  rank, world_size = 0, 1

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    return np.asarray(buf, dtype=np.float64)


class RcclComm:
  """RCCL all-reduce through libhbo (ncclAllReduce on the context's stream, xGMI)."""

  def __init__(self, ctx, rank: int, world_size: int, bcast_bytes: Callable[[bytes], bytes]):
    from hyperbo_amd import _native as nat
    self._nat, self.ctx = nat, ctx
    self.rank, self.world_size = rank, world_size
    uid = C.create_string_buffer(nat.UNIQUE_ID_BYTES)
    if rank == 0:
      rc = nat.lib().hbo_comm_unique_id(uid)
      if rc != nat.HBO_OK:
        raise nat.HboError(rc, (nat.lib().hbo_last_error(None) or b'').decode())
    uid_bytes = bcast_bytes(bytes(uid.raw))
    buf = C.create_string_buffer(uid_bytes, nat.UNIQUE_ID_BYTES)
    ctx.check(nat.lib().hbo_comm_init(ctx.handle, rank, world_size, buf), allow_not_pd=False)

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(buf, dtype=np.float64).copy()
    self.ctx.check(self._nat.lib().hbo_comm_allreduce_sum(
        self.ctx.handle, a.ctypes.data_as(C.POINTER(C.c_double)), a.size), allow_not_pd=False)
    return a

  # objectives.nll_value_and_grad(..., comm=this) takes the device-resident route: hbo_objective_sharded reduces the shard on
  # the device, all-reduces in place on the context's stream and copies [nll, count, grad] back once
  native_sharded = True
  last_timing = None   # (ms of device time on this rank's shard, us of the all-reduce) of the last sharded evaluation

  def close(self):
    self._nat.lib().hbo_comm_destroy(self.ctx.handle)


def sharded_mean_nll(local_nll_sum: float, local_count: int, local_grad_sum: np.ndarray, comm):
  """All-reduce [nll_sum, count, grad_sum] and return the mean over ALL tasks (objectives.py:192-195)."""
  buf = np.concatenate([[float(local_nll_sum), float(local_count)], np.asarray(local_grad_sum, dtype=np.float64)])
  buf = comm.allreduce_sum(buf)
  total, count, grad = buf[0], buf[1], buf[2:]
  if count <= 0:
    return 0.0, grad * 0.0, 0
  if np.isnan(count):   # a peer contributed NaN (local failure): the objective is NaN, not zero
    return float('nan'), grad * np.nan, 0
  return total / count, grad / count, int(round(count))
