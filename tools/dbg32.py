"""fp32 factorisation / inverse on the bf16 matrix cores against fp64 NumPy: error of chol and of the inverse per option."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hyperbo_amd import _native as nat
from hyperbo_amd.basics import linalg
ctx = nat.default_context()
sizes = [int(a) for a in sys.argv[1:]] or [300, 4224, 5000]
for n in sizes:
    rng = np.random.default_rng(n)
    m_ = rng.normal(size=(n, n))
    a = (m_ @ m_.T / n + np.eye(n)).astype(np.float32)
    b = rng.normal(size=(n, 3)).astype(np.float32)
    a64 = a.astype(np.float64)
    ref = np.linalg.cholesky(a64); inv_ref = np.linalg.inv(a64)
    for name, opts in (('fp32 MFMA', dict(syrk_bf16x3=0, trtri_bf16x3=0)), ('bf16x3 updates', dict(syrk_bf16x3=1, trtri_bf16x3=0)),
                       ('bf16x3 updates + inverse', dict(syrk_bf16x3=1, trtri_bf16x3=1))):
        for k, v in opts.items(): ctx.set_option(k, v)
        chol, x = linalg.solve_linear_system(a, b)
        inv, _ = linalg.spd_inverse(a)
        print('n=%d %-26s |chol - ref| %.3e   |inv - ref| / |ref| %.3e   |x - ref| %.3e' % (
            n, name, np.abs(chol - ref).max(), np.abs(inv - inv_ref).max() / np.abs(inv_ref).max(),
            np.abs(x - np.linalg.solve(a64, b.astype(np.float64))).max()), flush=True)
