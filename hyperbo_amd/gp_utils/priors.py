"""Log-priors on hyper-parameters -- hyperbo/gp_utils/priors.py:37-45 restated on the host
(three scalar log-pdfs; tensorflow_probability is not a dependency here)."""
import numpy as np


def _normal_log_prob(x, loc, scale):
  x = np.asarray(x, dtype=np.float64)
  return -0.5 * ((x - loc) / scale)**2 - np.log(scale) - 0.5 * np.log(2 * np.pi)


def noise_prior(x):      # tfd.Normal(0., 0.1)
  return float(np.sum(_normal_log_prob(x, 0., 0.1)))


def lognormal_prior(x):  # tfd.LogNormal(0., 1.)
  x = np.asarray(x, dtype=np.float64)
  return float(np.sum(_normal_log_prob(np.log(x), 0., 1.) - np.log(x)))


def constant_prior(x):   # tfd.Normal(0., 1.)
  return float(np.sum(_normal_log_prob(x, 0., 1.)))


noise_prior.derivative = lambda x: -np.asarray(x, dtype=np.float64) / 0.01
lognormal_prior.derivative = lambda x: -(np.log(np.asarray(x, dtype=np.float64)) + 1.0) / np.asarray(x, dtype=np.float64)
constant_prior.derivative = lambda x: -np.asarray(x, dtype=np.float64)

DEFAULT_PRIORS = {
    'noise_variance': noise_prior,
    'signal_variance': lognormal_prior,
    'constant': constant_prior,
}


def gradient_of(log_prior_fn):
  d = getattr(log_prior_fn, 'derivative', None)
  if d is None:
    raise NotImplementedError(f'prior {log_prior_fn!r} has no `.derivative`; gradients need it (no autodiff)')
  return d
