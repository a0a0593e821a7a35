"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/hyperbo_oracle.py) -- never imported by the product path.

SURVEY.md 8(d): "a harness hook --jax should time a builder-written JAX expression of the same formulas if and only if
`import jax` succeeds on the box".  This is that expression: the single-task SE-ARD + constant-mean NLL of
hyperbo/gp_utils/objectives.py:144-156 (K + (noise + 1e-6) I, Cholesky, cho_solve) and jax.value_and_grad of it
(what hyperbo/gp_utils/gp.py:134 calls), under x64, jitted, on the host CPU.  It is NOT the reference (the reference's own
Python cannot travel to the GPU box); it only answers "what does XLA-CPU make of the same mathematics here".  The image this
round was built in has no jax, so the module has never run: bench.py reports an error string instead of numbers if anything
in here fails.
"""
import time

import numpy as np


def time_nll_and_grad(x, y, raw, budget_s=20.0):
  import jax
  jax.config.update('jax_enable_x64', True)
  import jax.numpy as jnp
  import jax.scipy.linalg as jsl

  softplus = lambda v: jnp.logaddexp(v, 0.0) + 1e-10     # utils.DEFAULT_WARP_FUNC (utils.py:73-81)
  xj, yj = jnp.asarray(x, dtype=jnp.float64), jnp.asarray(y, dtype=jnp.float64)

  def nll(theta):
    ls, sv, noise = softplus(theta['lengthscale']), softplus(theta['signal_variance']), softplus(theta['noise_variance'])
    z = xj / ls
    sq = jnp.sum(z * z, axis=1)
    d2 = jnp.maximum(sq[:, None] + sq[None, :] - 2.0 * z @ z.T, 0.0)
    k = sv * jnp.exp(-0.5 * d2) + (noise + 1e-6) * jnp.eye(xj.shape[0])
    r = yj - theta['constant']
    chol = jsl.cholesky(k, lower=True)
    alpha = jsl.cho_solve((chol, True), r)
    return jnp.sum(0.5 * r.T @ alpha + jnp.sum(jnp.log(jnp.diag(chol))) + 0.5 * xj.shape[0] * jnp.log(2 * jnp.pi))

  theta = {k: jnp.asarray(v, dtype=jnp.float64) for k, v in raw.items()}
  f = jax.jit(jax.value_and_grad(nll))
  try:
    t0 = time.perf_counter(); v, g = f(theta); jax.block_until_ready(g); compile_s = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while n < 8 and time.perf_counter() - t0 < budget_s:
      v, g = f(theta); jax.block_until_ready(g); n += 1
    el = time.perf_counter() - t0
    return {'value': round(n / el, 4), 'unit': 'evals/s', 'evals': n, 'seconds': round(el, 2), 'compile_s': round(compile_s, 2),
            'nll': float(v), 'jax': jax.__version__, 'backend': jax.default_backend(),
            'kind': 'builder-written JAX expression of the same formulas (not the reference), distances by the norm trick'}
  except Exception as e:  # pylint: disable=broad-except
    return {'error': str(e)[:200]}
