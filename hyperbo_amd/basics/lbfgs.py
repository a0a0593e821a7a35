"""L-BFGS with a two-directional backtracking (Armijo + curvature) line search over parameter
pytrees -- host-side restatement of hyperbo/basics/lbfgs.py:51-349 for the native
`value_and_grad` (there is no autodiff: the caller passes val_and_grad_fn).

Behaviour kept from the reference: convergence on g.g <= tol (`tol` compares the SQUARED gradient
norm, lbfgs.py:241,293), first step = steepest descent with step 1/sqrt(g.g) (lbfgs.py:256),
Nocedal two-loop direction with gamma = s.y / y.y (lbfgs.py:143-183), line search growth 2.1 /
shrink tau, NaN/inf -> (cur_val, 0.) (lbfgs.py:136-139), stop when the curvature y.s is not finite
or < tol (lbfgs.py:322,341-344), resumable `state=(s_k, y_k, old_grads, old_params)`.
Deviation: a non-descent direction returns (cur_val, 0.) -- the reference returns `params` in place
of the value there (lbfgs.py:103-106), which its caller then compares with a float.
"""
import logging

import numpy as np


def tree_flatten(tree):
  """Deterministic (sorted-key) flattening of nested dicts of arrays -> (vector, unflatten)."""
  leaves = []

  def rec(t):
    if isinstance(t, dict):
      return {k: rec(t[k]) for k in sorted(t)}
    a = np.asarray(t, dtype=np.float64)
    leaves.append(a)
    return (len(leaves) - 1, a.shape, np.asarray(t).dtype)

  spec = rec(tree)
  vec = np.concatenate([l.ravel() for l in leaves]) if leaves else np.zeros(0)
  offsets = np.cumsum([0] + [l.size for l in leaves])

  def unflatten(v):
    def rec2(s):
      if isinstance(s, dict):
        return {k: rec2(x) for k, x in s.items()}
      idx, shape, dtype = s
      out = np.asarray(v[offsets[idx]:offsets[idx + 1]], dtype=np.float64).reshape(shape)
      return out.astype(dtype) if np.issubdtype(dtype, np.floating) else out
    return rec2(spec)

  return vec, unflatten


def backtracking_linesearch(val_and_grad_vec, cur_val, x, g, direction, alpha=1., c1=1e-4, c2=0.9, tau=0.5,
                            max_steps=50):
  """Returns (new_val, step).  `val_and_grad_vec(x) -> (val, grad_vector)`."""
  g_dot_d = float(np.vdot(g, direction))
  if g_dot_d > 0.:
    logging.info('Incorrect descent direction %f.  Exiting linesearch', g_dot_d)
    return cur_val, 0.
  t = c1 * g_dot_d
  new_val = cur_val
  for i in range(max_steps):
    new_val, new_g = val_and_grad_vec(x + alpha * direction)
    armijo = np.isfinite(new_val) and (cur_val + alpha * t >= new_val)
    logging.info('Linesearch: step %i orig: %f new: %f step size: %f Armijo cond %d', i, cur_val, new_val,
                 alpha, armijo)
    if armijo:
      if float(np.vdot(new_g, direction)) >= c2 * g_dot_d:
        return new_val, alpha
      alpha *= 2.1
    else:
      alpha *= tau
  if np.isfinite(new_val):
    return new_val, alpha
  return cur_val, 0.   # NaN / inf: stay where we started


def lbfgs_descent_dir_nocedal(g, s, y):
  """Two-loop recursion (Nocedal '80, p. 779); s/y: lists of parameter / gradient differences."""
  q = -g
  inv_p = [1. / float(np.vdot(y[i], s[i])) for i in range(len(s))]
  alphas = {}
  for i in range(len(s) - 1, -1, -1):
    a = inv_p[i] * float(np.vdot(s[i], q))
    alphas[i] = a
    q = q - a * y[i]
  gamma = float(np.vdot(s[-1], y[-1])) / float(np.vdot(y[-1], y[-1]))
  d = gamma * q
  for i in range(len(s)):
    beta = inv_p[i] * float(np.vdot(y[i], d))
    d = d + s[i] * (alphas[i] - beta)
  return d


def lbfgs(fn, params, memory=10, ls_steps=50, steps=100, alpha=1., tol=1e-6, ls_tau=0.5, val_and_grad_fn=None,
          state=None, callback=None):
  """Minimise fn(params).  Returns (value, params, state) like hyperbo/basics/lbfgs.py:186-349.

  val_and_grad_fn(params) -> (value, grads pytree) is REQUIRED (no autodiff); `fn` is kept in the
  signature for call-site compatibility and only used if it carries a `.value_and_grad` attribute.
  """
  if val_and_grad_fn is None:
    val_and_grad_fn = getattr(fn, 'value_and_grad', None)
    if val_and_grad_fn is None:
      raise TypeError('lbfgs needs val_and_grad_fn (there is no autodiff in hyperbo_amd)')
  x, unflatten = tree_flatten(params)

  def vg(v):
    val, grads = val_and_grad_fn(unflatten(v))
    gv, _ = tree_flatten(grads)
    return float(val), gv

  if state is None:
    s_k, y_k = [], []
    val, g = vg(x)
    if callback is not None:
      callback(step=0, model_params=unflatten(x), loss=val)
    gg = float(np.vdot(g, g))
    if gg <= tol:
      logging.info('LBFGS converged at start.')
      return val, unflatten(x), None
    old_x, old_g = x.copy(), g.copy()
    new_val, step = backtracking_linesearch(vg, val, x, g, -g, 1. / np.sqrt(gg), tau=ls_tau, max_steps=ls_steps)
    if new_val < val:
      x = x - step * g
    else:
      logging.info('Linesearch did not make progress.')
      return new_val, unflatten(x), (s_k, y_k, old_g, old_x)
  else:
    s_k, y_k, old_g, old_x = state
    s_k, y_k = list(s_k), list(y_k)
    new_val = None
  for i in range(1, steps + 1):
    val, g = vg(x)
    if float(np.vdot(g, g)) <= tol:
      logging.info('LBFGS converged in %d steps', i)
      new_val = val
      break
    if old_g is not None:
      y_k.append(g - old_g)
      s_k.append(x - old_x)
    s_k, y_k = s_k[-memory:], y_k[-memory:]
    old_x, old_g = x.copy(), g.copy()
    magnitude = float(np.vdot(y_k[-1], s_k[-1]))
    logging.info('LBFGS step %d val: %f', i, val)
    if callback is not None:
      callback(step=i, model_params=unflatten(x), loss=val)
    if np.isfinite(magnitude) and magnitude >= tol:
      d = lbfgs_descent_dir_nocedal(g, s_k, y_k)
      new_val, step = backtracking_linesearch(vg, val, x, g, d, alpha, tau=ls_tau, max_steps=ls_steps)
      if new_val >= val:
        logging.info('Linesearch did not make progress.')
        break
      x = x + step * d
    else:
      new_val = val
      logging.info('LBFGS terminating due to instability.')
      break
  return new_val, unflatten(x), (s_k, y_k, old_g, old_x)
