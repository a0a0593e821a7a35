/* CPU port (OpenMP) of the O(N^2 D) pieces of the SE-ARD NLL+gradient, used ONLY by bench.py's
 * cpu_baseline leg and tests (test infrastructure, never linked into libhbo).  LAPACK potrf/potri
 * are called from Python (SciPy/OpenBLAS).  Restates hyperbo/gp_utils/kernel.py:63-81 (Gram) and
 * the contraction sum_ij G_ij dK_ij/dtheta that jax.grad of objectives.py:144-156 produces.
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cpu_port.c -o oracle/libcpu_port.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* K[i][j] = sv * exp(-0.5 * |xs_i - xs_j|^2) (+ diag_add on the diagonal); xs is n x d, pre-scaled. */
void hbo_cpu_gram_se(const double* xs, int64_t n, int64_t d, double sv, double diag_add, double* K) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < n; ++i) {
    const double* xi = xs + i * d;
    for (int64_t j = 0; j <= i; ++j) {
      const double* xj = xs + j * d;
      double u = 0;
      for (int64_t k = 0; k < d; ++k) { const double t = xi[k] - xj[k]; u += t * t; }
      double v = sv * exp(-0.5 * u);
      if (i == j) v += diag_add;
      K[i * n + j] = v;
      K[j * n + i] = v;
    }
  }
}

/* Kinv: n x n with the triangle {row >= col in C order} valid; alpha: n.  Outputs:
 * out[0] = sum_ij G_ij k_ij, out[1] = tr G, out[2+k] = sum_ij G_ij k_ij (xs_ik - xs_jk)^2,
 * with G = 0.5 (Kinv - alpha alpha^T), k_ij = sv exp(-u/2) recomputed. */
void hbo_cpu_contract_se(const double* xs, int64_t n, int64_t d, double sv, const double* Kinv,
                         const double* alpha, double* out) {
  const int64_t nacc = 2 + d;
  for (int64_t q = 0; q < nacc; ++q) out[q] = 0;
#pragma omp parallel
  {
    double acc[2 + 64];
    for (int64_t q = 0; q < nacc; ++q) acc[q] = 0;
#pragma omp for schedule(dynamic, 16) nowait
    for (int64_t i = 0; i < n; ++i) {
      const double* xi = xs + i * d;
      for (int64_t j = 0; j <= i; ++j) {
        const double* xj = xs + j * d;
        double u = 0, df[64];
        for (int64_t k = 0; k < d; ++k) { df[k] = xi[k] - xj[k]; u += df[k] * df[k]; }
        const double kij = sv * exp(-0.5 * u);
        const double w = (i == j) ? 1.0 : 2.0;
        const double G = 0.5 * (Kinv[i * n + j] - alpha[i] * alpha[j]);
        const double gk = w * G * kij;
        acc[0] += gk;
        if (i == j) acc[1] += G;
        for (int64_t k = 0; k < d; ++k) acc[2 + k] += gk * df[k] * df[k];
      }
    }
#pragma omp critical
    for (int64_t q = 0; q < nacc; ++q) out[q] += acc[q];
  }
}
