import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import cpu_baseline
x, y, raw = bench.cfg2_inputs()
import torch
print('torch threads default', torch.get_num_threads(), torch.__config__.parallel_info().split('\n')[0:3])
for nt in (32, 64, 128, 256):
    t0 = time.perf_counter(); v, g = cpu_baseline.nll_and_grad_se_ard_constant_torch(x, y, raw, threads=nt); t1 = time.perf_counter()
    t2 = time.perf_counter(); v, g = cpu_baseline.nll_and_grad_se_ard_constant_torch(x, y, raw, threads=nt); t3 = time.perf_counter()
    print(nt, 'threads: %.2f s, %.2f s' % (t1 - t0, t3 - t2), v)
