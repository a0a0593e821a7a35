/* CPU port (OpenMP) of the O(N^2 D) pieces of the SE-ARD NLL+gradient, used ONLY by bench.py's
 * cpu_baseline leg and tests (test infrastructure, never linked into libhbo).  LAPACK potrf/potri
 * are called from Python (SciPy/OpenBLAS).  Restates hyperbo/gp_utils/kernel.py:63-81 (Gram) and
 * the contraction sum_ij G_ij dK_ij/dtheta that jax.grad of objectives.py:144-156 produces.
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cpu_port.c -o oracle/libcpu_port.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <omp.h>

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

/* ---- all-core potrf / trtri / lauum over tiles (round 6) ---------------------------------------------------------------------
 * The OpenBLAS inside SciPy is built for at most 64 threads; the GPU box has 256 cores.  These three routines run the blocked
 * algorithms over nb x nb tiles with OpenMP across ALL cores, every tile operation a SINGLE-threaded BLAS / LAPACK call through the
 * function pointers the Python side takes from scipy.linalg.cython_blas / cython_lapack (the caller limits the BLAS pool to one thread
 * for the duration).  Column-major, UPPER: A = U^T U; on a C-ordered symmetric buffer that is the lower factor.  Test / bench
 * infrastructure only (oracle/cpu_baseline.py: nll_and_grad_se_ard_constant_tiled). */
typedef void (*dgemm_t)(char*, char*, int*, int*, int*, double*, double*, int*, double*, int*, double*, double*, int*);
typedef void (*dtrsm_t)(char*, char*, char*, char*, int*, int*, double*, double*, int*, double*, int*);
typedef void (*dsyrk_t)(char*, char*, int*, int*, double*, double*, int*, double*, double*, int*);
typedef void (*dpotrf_t)(char*, int*, double*, int*, int*);
typedef void (*dtrtri_t)(char*, char*, int*, double*, int*, int*);
typedef struct { dgemm_t gemm; dtrsm_t trsm; dsyrk_t syrk; dpotrf_t potrf; dtrtri_t trtri; } hbo_blas_fns;

#define BLK(A, i, j) ((A) + (int64_t)(j) * nb * n + (int64_t)(i) * nb)
static int bs_of(int64_t n, int nb, int i) { const int64_t r = n - (int64_t)i * nb; return (int)(r < nb ? r : nb); }

/* A (n x n, column-major, leading dimension n, upper triangle read) -> U in the upper triangle.  Returns LAPACK's info of the
 * first failing diagonal tile (offset by its position), 0 on success. */
int hbo_cpu_potrf_tiled(double* A, int64_t n64, int nb, const hbo_blas_fns* f) {
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  double one = 1.0, m1 = -1.0;
  for (int k = 0; k < T; ++k) {
    int bk = bs_of(n, nb, k), info = 0;
    f->potrf("U", &bk, BLK(A, k, k), &n, &info);
    if (info) return k * nb + info;
    const int ncol = n - (k + 1) * nb;   /* columns right of the diagonal tile, in strips of 64 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int c0 = 0; c0 < ncol; c0 += 64) {
      int w = ncol - c0 < 64 ? ncol - c0 : 64;
      f->trsm("L", "U", "T", "N", &bk, &w, &one, BLK(A, k, k), &n, A + ((int64_t)(k + 1) * nb + c0) * n + (int64_t)k * nb, &n);
    }
    const int m = T - 1 - k;              /* trailing tiles (i <= j), columns first: the next panel's column is finished early */
    const int npair = m * (m + 1) / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < npair; ++t) {
      int j = 0, r = t;
      while (r > j) { r -= j + 1; ++j; }
      const int ti = k + 1 + r, tj = k + 1 + j;
      int bi = bs_of(n, nb, ti), bj = bs_of(n, nb, tj);
      if (ti == tj) f->syrk("U", "T", &bi, &bk, &m1, BLK(A, k, ti), &n, &one, BLK(A, ti, ti), &n);
      else f->gemm("T", "N", &bi, &bj, &bk, &m1, BLK(A, k, ti), &n, BLK(A, k, tj), &n, &one, BLK(A, ti, tj), &n);
    }
  }
  return 0;
}

/* U (upper, from the routine above) -> V = U^-1 in place (tile algorithm; the strictly lower parts of the diagonal tiles are zeroed
 * so that hbo_cpu_lauum_tiled can read whole tiles). */
int hbo_cpu_trtri_tiled(double* A, int64_t n64, int nb, const hbo_blas_fns* f) {
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  double one = 1.0, m1 = -1.0;
  for (int k = 0; k < T; ++k) {
    int bk = bs_of(n, nb, k);
#pragma omp parallel for schedule(dynamic, 1)
    for (int mi = 0; mi < k; ++mi) {       /* A(m,k) = -A(m,k) U_kk^-1 */
      int bm = nb;
      f->trsm("R", "U", "N", "N", &bm, &bk, &m1, BLK(A, k, k), &n, BLK(A, mi, k), &n);
    }
    const int right = T - 1 - k;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int nj = 0; nj < right; ++nj)
      for (int mi = 0; mi < k; ++mi) {     /* A(m,n) += A(m,k) A(k,n) */
        int bm = nb, bn = bs_of(n, nb, k + 1 + nj);
        f->gemm("N", "N", &bm, &bn, &bk, &one, BLK(A, mi, k), &n, BLK(A, k, k + 1 + nj), &n, &one, BLK(A, mi, k + 1 + nj), &n);
      }
#pragma omp parallel for schedule(dynamic, 1)
    for (int nj = 0; nj < right; ++nj) {   /* A(k,n) = U_kk^-1 A(k,n) */
      int bn = bs_of(n, nb, k + 1 + nj);
      f->trsm("L", "U", "N", "N", &bk, &bn, &one, BLK(A, k, k), &n, BLK(A, k, k + 1 + nj), &n);
    }
    int info = 0;
    f->trtri("U", "N", &bk, BLK(A, k, k), &n, &info);
    if (info) return k * nb + info;
    double* d = BLK(A, k, k);
    for (int c = 0; c < bk; ++c) for (int r = c + 1; r < bk; ++r) d[(int64_t)c * n + r] = 0.0;
  }
  return 0;
}

/* out (upper tiles, i <= j) = V V^T for the upper-triangular V of the routine above: tile (i, j) = sum_{k >= j} V(i,k) V(j,k)^T,
 * one GEMM per tile over the contiguous row panels, longest K first. */
void hbo_cpu_lauum_tiled(const double* V, int64_t n64, int nb, const hbo_blas_fns* f, double* out) {
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  const int npair = T * (T + 1) / 2;
  double one = 1.0, zero = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < npair; ++t) {
    int j = 0, r = t;
    while (r > j) { r -= j + 1; ++j; }     /* column tile j ascending = K descending */
    const int i = r;
    int bi = bs_of(n, nb, i), bj = bs_of(n, nb, j), K = n - j * nb;
    f->gemm("N", "T", &bi, &bj, &K, &one, (double*)BLK(V, i, j), &n, (double*)BLK(V, j, j), &n, &zero, BLK(out, i, j), &n);
  }
}
int hbo_cpu_omp_threads(void) {
  int nthr = 1;
#pragma omp parallel
  {
#pragma omp master
    nthr = omp_get_num_threads();
  }
  return nthr;
}
