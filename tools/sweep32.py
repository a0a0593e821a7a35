"""fp32 factor / solve / inverse on the bf16 cores against fp64 LAPACK over sizes that exercise odd block counts and partial groups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as spla
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import linalg
ctx = nat.default_context()
sizes = [int(a) for a in sys.argv[1:]] or [4097, 4225, 4353, 4736, 5250, 5633, 6017, 7000, 8321]
for n in sizes:
    rng = np.random.default_rng(n)
    g = rng.normal(size=(n, 64)).astype(np.float32)
    a = (g @ g.T / 64 + np.eye(n, dtype=np.float32) * 0.5).astype(np.float32)
    a = 0.5 * (a + a.T)
    b = rng.normal(size=(n, 1)).astype(np.float32)
    a64 = a.astype(np.float64)
    cref = spla.cholesky(a64, lower=True); iref = spla.cho_solve((cref, True), np.eye(n)); xref = iref @ b.astype(np.float64)
    out = []
    for on in (0, 1):
        ctx.set_option('bf16x3', on)
        chol, x = linalg.solve_linear_system(a, b)
        inv, _ = linalg.spd_inverse(a)
        rel = lambda u, v: np.abs(u - v).max() / np.abs(v).max()
        out.append((rel(chol, cref), rel(x, xref), rel(inv, iref), rel(np.tril(inv), np.tril(inv.T))))
    print('n = %5d (%2d blocks)  mfma: chol %.1e x %.1e inv %.1e | bf16x3: chol %.1e x %.1e inv %.1e  sym %.1e' % (
        n, (n + 127) // 128, *out[0][:3], *out[1]), flush=True)
ctx.set_option('bf16x3', 1)
