"""The two-columns-per-step elimination of the fp64 leaf Cholesky (hyperbo_amd/csrc/chol.hip, leaf_cholesky) written
out in NumPy: both pivots and the second update vector are computed from the un-updated tile, the two rank-1 updates are
applied as one rank-2 update, the inverse takes the same eliminations.  Must reproduce L and L^-1 of LAPACK."""
import numpy as np
import pytest


def two_column_leaf(s0):
  n = s0.shape[0]
  s = s0.copy(); v = np.eye(n); inv = np.zeros(n)
  c = np.arange(n)
  for j in range(0, n, 2):
    d00, d10, d11 = s[j, j], s[j + 1, j], s[j + 1, j + 1]
    i0 = 1.0 / np.sqrt(d00); l10 = d10 * i0
    i1 = 1.0 / np.sqrt(d11 - l10 * l10)
    inv[j], inv[j + 1] = i0, i1
    f0 = np.where(c > j, s[j, :] * i0, 0.0)
    f1 = np.where(c > j + 1, (s[j + 1, :] - f0 * l10) * i1, 0.0)
    h0 = v[j, :] * i0
    h1 = (v[j + 1, :] - l10 * h0) * i1
    s = s - np.outer(f0, f0) - np.outer(f1, f1)      # one MFMA, two K slices
    v = v - np.outer(f0, h0) - np.outer(f1, h1)
  return np.tril(s * inv[None, :]), inv[:, None] * v


@pytest.mark.parametrize('seed', range(5))
def test_two_column_elimination_matches_lapack(seed):
  rng = np.random.default_rng(seed)
  a = rng.normal(size=(16, 16))
  s0 = a @ a.T + 16 * np.eye(16)
  l, m = two_column_leaf(s0)
  ref = np.linalg.cholesky(s0)
  assert np.max(np.abs(l - ref)) <= 1e-13 * np.max(np.abs(ref))
  assert np.max(np.abs(m @ ref - np.eye(16))) <= 1e-13


def test_identity_padding_is_a_fixed_point():
  s0 = np.eye(16); s0[:5, :5] = np.array([[4.0 if i == j else 0.1 for j in range(5)] for i in range(5)])
  l, m = two_column_leaf(s0)
  assert np.array_equal(l[5:, 5:], np.eye(11)) and np.array_equal(m[5:, 5:], np.eye(11))
  assert np.max(np.abs(l @ l.T - s0)) <= 1e-14
