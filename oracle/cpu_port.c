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
 * The OpenBLAS inside SciPy is built for at most 64 threads (more concurrent callers than that run into its buffer table: a first
 * version of this file called single-threaded dgemm from 256 OpenMP threads and crashed inside it on the 256-core GPU box).  So the
 * tile products below use their OWN GEMM micro-kernel (AVX2 + FMA, 8 x 6 register tile, packed panels) and OpenMP across ALL cores;
 * only the nb x nb diagonal tiles go through LAPACK (dpotrf / dtrtri from scipy.linalg.cython_lapack), one call at a time from the
 * master thread.  Triangular solves against a diagonal tile are products with its explicit inverse (error ~ cond(tile) eps).
 * Column-major, UPPER: A = U^T U; on a C-ordered symmetric buffer that is the lower factor.  Test / bench infrastructure only
 * (oracle/cpu_baseline.py: nll_and_grad_se_ard_constant_tiled). */
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
typedef void (*dpotrf_t)(char*, int*, double*, int*, int*);
typedef void (*dtrtri_t)(char*, char*, int*, double*, int*, int*);
typedef struct { void* gemm; void* trsm; void* syrk; dpotrf_t potrf; dtrtri_t trtri; } hbo_blas_fns;   /* only potrf / trtri are used */

#define MR 8
#define NR 6
/* C (m x n, column-major, ldc) = beta C + alpha op(A) op(B), op given by element strides: A(i,k) = A[i*ars + k*acs], B(k,j) = B[k*brs + j*bcs].
 * One call = one tile product on ONE thread (m, n <= a few hundred, any k); pa / pb: caller's scratch of >= (m+MR)*k and (n+NR)*k doubles. */
static void tile_gemm(int m, int n, int k, double alpha, const double* A, int64_t ars, int64_t acs, const double* B, int64_t brs, int64_t bcs,
                      double beta, double* C, int64_t ldc, double* pa, double* pb) {
  const int mp = (m + MR - 1) / MR, np = (n + NR - 1) / NR;
  for (int ip = 0; ip < mp; ++ip) {          /* A panels: [k][MR], zero padded */
    double* d = pa + (int64_t)ip * MR * k;
    const int rows = m - ip * MR < MR ? m - ip * MR : MR;
    for (int kk = 0; kk < k; ++kk) {
      const double* a = A + (int64_t)(ip * MR) * ars + (int64_t)kk * acs;
      for (int r = 0; r < rows; ++r) d[kk * MR + r] = a[(int64_t)r * ars];
      for (int r = rows; r < MR; ++r) d[kk * MR + r] = 0.0;
    }
  }
  for (int jp = 0; jp < np; ++jp) {          /* B panels: [k][NR], scaled by alpha */
    double* d = pb + (int64_t)jp * NR * k;
    const int cols = n - jp * NR < NR ? n - jp * NR : NR;
    for (int kk = 0; kk < k; ++kk) {
      const double* b = B + (int64_t)kk * brs + (int64_t)(jp * NR) * bcs;
      for (int c = 0; c < cols; ++c) d[kk * NR + c] = alpha * b[(int64_t)c * bcs];
      for (int c = cols; c < NR; ++c) d[kk * NR + c] = 0.0;
    }
  }
  for (int jp = 0; jp < np; ++jp) {
    const double* bp = pb + (int64_t)jp * NR * k;
    const int cols = n - jp * NR < NR ? n - jp * NR : NR;
    for (int ip = 0; ip < mp; ++ip) {
      const double* ap = pa + (int64_t)ip * MR * k;
      const int rows = m - ip * MR < MR ? m - ip * MR : MR;
      __m256d c0[NR], c1[NR];
      for (int c = 0; c < NR; ++c) { c0[c] = _mm256_setzero_pd(); c1[c] = _mm256_setzero_pd(); }
      for (int kk = 0; kk < k; ++kk) {
        const __m256d a0 = _mm256_loadu_pd(ap + kk * MR), a1 = _mm256_loadu_pd(ap + kk * MR + 4);
#pragma GCC unroll 6
        for (int c = 0; c < NR; ++c) {
          const __m256d b = _mm256_broadcast_sd(bp + kk * NR + c);
          c0[c] = _mm256_fmadd_pd(a0, b, c0[c]);
          c1[c] = _mm256_fmadd_pd(a1, b, c1[c]);
        }
      }
      double* cc = C + (int64_t)(jp * NR) * ldc + ip * MR;
      if (rows == MR) {
        const __m256d vb = _mm256_set1_pd(beta);
        for (int c = 0; c < cols; ++c) {
          double* col = cc + (int64_t)c * ldc;
          if (beta == 0.0) { _mm256_storeu_pd(col, c0[c]); _mm256_storeu_pd(col + 4, c1[c]); }
          else { _mm256_storeu_pd(col, _mm256_fmadd_pd(vb, _mm256_loadu_pd(col), c0[c])); _mm256_storeu_pd(col + 4, _mm256_fmadd_pd(vb, _mm256_loadu_pd(col + 4), c1[c])); }
        }
      } else {
        double tmp[MR];
        for (int c = 0; c < cols; ++c) {
          _mm256_storeu_pd(tmp, c0[c]); _mm256_storeu_pd(tmp + 4, c1[c]);
          double* col = cc + (int64_t)c * ldc;
          for (int r = 0; r < rows; ++r) col[r] = (beta == 0.0 ? 0.0 : beta * col[r]) + tmp[r];
        }
      }
    }
  }
}
/* per-thread packing scratch for tiles of up to nb x nb: ONE block for all threads per call (allocating inside every parallel region --
 * 256 threads x 64 steps of 270 KB mmap / munmap pairs, serialised on the address-space lock -- took 12.5 s of a potrf on the 256-core box) */
static size_t scratch_doubles(int nb) { return (size_t)((2 * (int64_t)nb + MR + NR) * nb + 64); }
static double* scratch_all(int nb) { return (double*)aligned_alloc(64, sizeof(double) * scratch_doubles(nb) * (size_t)omp_get_max_threads()); }
#define SCRATCH(all, nb) double* pa = (all) + scratch_doubles(nb) * (size_t)omp_get_thread_num(); double* pb = pa + (int64_t)((nb) + MR) * (nb)

#define BLK(A, i, j) ((A) + (int64_t)(j) * nb * n + (int64_t)(i) * nb)
static int bs_of(int64_t n, int nb, int i) { const int64_t r = n - (int64_t)i * nb; return (int)(r < nb ? r : nb); }

/* inverse of the upper-triangular bk x bk tile T (ld n) into inv (ld nb, zeros below the diagonal) */
static int tile_inverse(const double* T, int n, int bk, int nb, double* inv, dtrtri_t trtri) {
  for (int c = 0; c < bk; ++c) for (int r = 0; r < bk; ++r) inv[(int64_t)c * nb + r] = r <= c ? T[(int64_t)c * n + r] : 0.0;
  int info = 0, ld = nb;
  trtri("U", "N", &bk, inv, &ld, &info);
  return info;
}

/* A (n x n, column-major, leading dimension n, upper triangle read) -> U in the upper triangle.  Returns LAPACK's info of the
 * first failing diagonal tile (offset by its position), 0 on success. */
int hbo_cpu_potrf_tiled(double* A, int64_t n64, int nb, const hbo_blas_fns* f) {
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  double* inv = (double*)aligned_alloc(64, sizeof(double) * (size_t)nb * nb);
  double* row = (double*)aligned_alloc(64, sizeof(double) * (size_t)nb * n);   /* copy of block row k (the solve reads it while writing A) */
  double* scr = scratch_all(nb);
  int bad = 0;
  for (int k = 0; k < T && !bad; ++k) {
    int bk = bs_of(n, nb, k), info = 0;
    f->potrf("U", &bk, BLK(A, k, k), &n, &info);
    if (info) { bad = k * nb + info; break; }
    if (tile_inverse(BLK(A, k, k), n, bk, nb, inv, f->trtri)) { bad = k * nb + 1; break; }
    const int ncol = n - (k + 1) * nb;   /* A(k, j) = U_kk^-T A(k, j), in strips of 48 columns */
#pragma omp parallel
    {
      SCRATCH(scr, nb);
#pragma omp for schedule(dynamic, 1)
      for (int c0 = 0; c0 < ncol; c0 += 48) {
        const int w = ncol - c0 < 48 ? ncol - c0 : 48;
        double* dst = A + ((int64_t)(k + 1) * nb + c0) * n + (int64_t)k * nb;
        double* src = row + (int64_t)c0 * nb;
        for (int c = 0; c < w; ++c) memcpy(src + (int64_t)c * nb, dst + (int64_t)c * n, sizeof(double) * bk);
        tile_gemm(bk, w, bk, 1.0, inv, nb, 1, src, 1, nb, 0.0, dst, n, pa, pb);   /* inv^T: A(i,k) = inv[k*nb + i] -> ars = nb, acs = 1 */
      }
    }
    const int m = T - 1 - k;
    const int npair = m * (m + 1) / 2;
#pragma omp parallel
    {
      SCRATCH(scr, nb);
#pragma omp for schedule(dynamic, 1)
      for (int t = 0; t < npair; ++t) {
        int j = 0, r = t;
        while (r > j) { r -= j + 1; ++j; }
        const int ti = k + 1 + r, tj = k + 1 + j;
        const int bi = bs_of(n, nb, ti), bj = bs_of(n, nb, tj);
        /* A(ti,tj) -= A(k,ti)^T A(k,tj) */
        tile_gemm(bi, bj, bk, -1.0, BLK(A, k, ti), n, 1, BLK(A, k, tj), 1, n, 1.0, BLK(A, ti, tj), n, pa, pb);
      }
    }
  }
  free(inv); free(row); free(scr);
  return bad;
}

/* U (upper, from the routine above) -> V = U^-1 in place (tile algorithm; the strictly lower parts of the diagonal tiles are zeroed
 * so that hbo_cpu_lauum_tiled can read whole tiles). */
int hbo_cpu_trtri_tiled(double* A, int64_t n64, int nb, const hbo_blas_fns* f) {
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  double* inv = (double*)aligned_alloc(64, sizeof(double) * (size_t)nb * nb);
  double* tmp = (double*)aligned_alloc(64, sizeof(double) * (size_t)nb * n);
  double* scr = scratch_all(nb);
  int bad = 0;
  for (int k = 0; k < T && !bad; ++k) {
    const int bk = bs_of(n, nb, k);
    if (tile_inverse(BLK(A, k, k), n, bk, nb, inv, f->trtri)) { bad = k * nb + 1; break; }
    const int right = T - 1 - k;
#pragma omp parallel
    {
      SCRATCH(scr, nb);
#pragma omp for schedule(dynamic, 1)
      for (int mi = 0; mi < k; ++mi) {       /* A(m,k) = -A(m,k) U_kk^-1 */
        double* src = tmp + (int64_t)mi * nb * nb;
        double* dst = BLK(A, mi, k);
        for (int c = 0; c < bk; ++c) memcpy(src + (int64_t)c * nb, dst + (int64_t)c * n, sizeof(double) * nb);
        tile_gemm(nb, bk, bk, -1.0, src, 1, nb, inv, 1, nb, 0.0, dst, n, pa, pb);
      }
#pragma omp for schedule(dynamic, 1) collapse(2)
      for (int nj = 0; nj < right; ++nj)
        for (int mi = 0; mi < k; ++mi) {     /* A(m,n) += A(m,k) A(k,n) */
          const int bn = bs_of(n, nb, k + 1 + nj);
          tile_gemm(nb, bn, bk, 1.0, BLK(A, mi, k), 1, n, BLK(A, k, k + 1 + nj), 1, n, 1.0, BLK(A, mi, k + 1 + nj), n, pa, pb);
        }
#pragma omp for schedule(dynamic, 1)
      for (int nj = 0; nj < right; ++nj) {   /* A(k,n) = U_kk^-1 A(k,n) */
        const int bn = bs_of(n, nb, k + 1 + nj);
        double* src = tmp + (int64_t)nj * nb * nb;
        double* dst = BLK(A, k, k + 1 + nj);
        for (int c = 0; c < bn; ++c) memcpy(src + (int64_t)c * nb, dst + (int64_t)c * n, sizeof(double) * bk);
        tile_gemm(bk, bn, bk, 1.0, inv, 1, nb, src, 1, nb, 0.0, dst, n, pa, pb);
      }
    }
    double* d = BLK(A, k, k);
    for (int c = 0; c < bk; ++c) for (int r = 0; r < bk; ++r) d[(int64_t)c * n + r] = inv[(int64_t)c * nb + r];
  }
  free(inv); free(tmp); free(scr);
  return bad;
}

/* out (upper tiles, i <= j) = V V^T for the upper-triangular V of the routine above: tile (i, j) = sum_{k >= j} V(i,k) V(j,k)^T,
 * accumulated tile by tile over k, longest sums first. */
void hbo_cpu_lauum_tiled(const double* V, int64_t n64, int nb, const hbo_blas_fns* f, double* out) {
  (void)f;
  int n = (int)n64;
  const int T = (n + nb - 1) / nb;
  const int npair = T * (T + 1) / 2;
  double* scr = scratch_all(nb);
#pragma omp parallel
  {
    SCRATCH(scr, nb);
#pragma omp for schedule(dynamic, 1)
    for (int t = 0; t < npair; ++t) {
      int j = 0, r = t;
      while (r > j) { r -= j + 1; ++j; }     /* column tile j ascending = K descending */
      const int i = r;
      const int bi = bs_of(n, nb, i), bj = bs_of(n, nb, j);
      for (int kt = j; kt < T; ++kt) {
        const int bk = bs_of(n, nb, kt);
        tile_gemm(bi, bj, bk, 1.0, BLK(V, i, kt), 1, n, BLK(V, j, kt), n, 1, kt == j ? 0.0 : 1.0, BLK(out, i, j), n, pa, pb);
      }
    }
  }
  free(scr);
}
void hbo_cpu_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int hbo_cpu_omp_threads(void) {
  int nthr = 1;
#pragma omp parallel
  {
#pragma omp master
    nthr = omp_get_num_threads();
  }
  return nthr;
}
