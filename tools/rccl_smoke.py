"""Single-rank RCCL smoke test of libhbo's hbo_comm_* binding (nranks=1 on one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat, parallel
ctx = nat.default_context()
comm = parallel.RcclComm(ctx, 0, 1, lambda b: b)
x = np.arange(37, dtype=np.float64) * 0.5
y = comm.allreduce_sum(x)
assert np.array_equal(x, y), (x, y)
for _ in range(3):
    y = comm.allreduce_sum(y)
comm.close()
print('rccl single-rank allreduce ok')
