// Micro-benchmark of the diagonal-panel kernels (links chol.o directly).
#include "../hyperbo_amd/csrc/hbo_internal.h"
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void fill_spd(double* p, int64_t n, int64_t ld) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  int64_t r = i / n, c = i % n;
  unsigned h = (unsigned)(i * 2654435761u); h ^= h >> 13; h *= 0x5bd1e995; h ^= h >> 15;
  p[r * ld + c] = (r == c) ? 4.0 + (h & 0xff) / 256.0 : ((h & 0xffff) / 65536.0 - 0.5) * 0.01;
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 8192;
  int nblk = n / 128; int64_t ld = n + 16;
  double *A, *W; size_t na = (size_t)(n + 128) * ld;
  CK(hipMalloc(&A, na * 8)); CK(hipMalloc(&W, na * 8));
  CK(hipMemset(W, 0, na * 8)); CK(hipMemset(A, 0, na * 8));
  fill_spd<<<((size_t)n * n + 255) / 256, 256>>>(A, n, ld);
  TaskDesc h = {}; h.A = A; h.W = W; h.S = nullptr; h.n = n; h.npad = n; h.nblk = nblk; h.m = 1; h.ld = ld;
  TaskDesc* d; CK(hipMalloc(&d, sizeof h)); CK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
  int* info; CK(hipMalloc(&info, 4)); int inf = INT_MAX; CK(hipMemcpy(info, &inf, 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {  // correctness of one potf2 against a host Cholesky of the same diagonal block (and of the leaf inverses)
    const int pb = 3;
    std::vector<double> blk(128 * 128), Lg(128 * 128), Wg(128 * 128);
    CK(hipMemcpy2D(blk.data(), 128 * 8, A + (size_t)pb * 128 * ld + pb * 128, ld * 8, 128 * 8, 128, hipMemcpyDeviceToHost));
    launch_potf2(HBO_F64, d, 1, pb, info, 0, nullptr); CK(hipDeviceSynchronize());
    CK(hipMemcpy2D(Lg.data(), 128 * 8, A + (size_t)pb * 128 * ld + pb * 128, ld * 8, 128 * 8, 128, hipMemcpyDeviceToHost));
    CK(hipMemcpy2D(Wg.data(), 128 * 8, W + (size_t)pb * 128 * ld + pb * 128, ld * 8, 128 * 8, 128, hipMemcpyDeviceToHost));
    std::vector<double> Lh(blk);
    for (int j = 0; j < 128; ++j) {
      double dd = Lh[j * 128 + j];
      for (int k = 0; k < j; ++k) dd -= Lh[j * 128 + k] * Lh[j * 128 + k];
      dd = sqrt(dd); Lh[j * 128 + j] = dd;
      for (int i = j + 1; i < 128; ++i) { double v = Lh[i * 128 + j]; for (int k = 0; k < j; ++k) v -= Lh[i * 128 + k] * Lh[j * 128 + k]; Lh[i * 128 + j] = v / dd; }
    }
    double eL = 0, eW = 0, eU = 0;
    for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) { if (j <= i) eL = fmax(eL, fabs(Lg[i * 128 + j] - Lh[i * 128 + j])); else if (j / 16 == i / 16) eU = fmax(eU, fabs(Lg[i * 128 + j])); }
    for (int b = 0; b < 8; ++b)   // leaf inverse times leaf = identity
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double sum = 0;
        for (int k = 0; k < 16; ++k) sum += Wg[(b * 16 + i) * 128 + b * 16 + k] * (k >= j ? Lh[(b * 16 + k) * 128 + b * 16 + j] : 0.0);
        eW = fmax(eW, fabs(sum - (i == j ? 1.0 : 0.0)));
      }
    printf("potf2 check: max |L - L_host| = %.3e, max |upper of diagonal tiles| = %.3e, max |M L_leaf - I| = %.3e\n", eL, eU, eW);
  }
  const int reps = 20;
  // potf2 on (fresh copies of) the same diagonal block: p cycles over blocks so data stays SPD-ish
  launch_potf2(HBO_F64, d, 1, 0, info, 0, nullptr); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_potf2(HBO_F64, d, 1, 1 + i, info, 0, nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("potf2            %8.2f us per launch\n", ms / reps * 1e3);
#ifdef HBO_POTF2_TIMING
  {
    extern void dbg_read_stamps(unsigned long long*);
    unsigned long long st[64]; dbg_read_stamps(st);
    const double tot = (double)(st[32] - st[0]);
    printf("  stamps (cycles, %% of kernel body):\n");
    printf("   load+sync      %8llu %5.1f%%\n", st[1] - st[0], 100.0 * (st[1] - st[0]) / tot);
    printf("   first leaf     %8llu %5.1f%%\n", st[2] - st[1], 100.0 * (st[2] - st[1]) / tot);
    unsigned long long prev = st[2];
    for (int jb = 0; jb < 8; ++jb) {
      const unsigned long long b = st[3 + 3 * jb];
      if (jb < 7) printf("   jb=%d  (B) %6llu   (C) wave0 %6llu  all %6llu\n", jb, b - prev, st[4 + 3 * jb] - b, st[5 + 3 * jb] - b);
      else printf("   jb=%d  (B) %6llu\n", jb, b - prev);
      prev = jb < 7 ? st[5 + 3 * jb] : b;
    }
    printf("   store L        %8llu %5.1f%%\n", st[31] - st[30], 100.0 * (st[31] - st[30]) / tot);
    printf("   leaf inverses  %8llu %5.1f%%\n", st[32] - st[31], 100.0 * (st[32] - st[31]) / tot);
    printf("   total          %8.0f cycles\n", tot);
  }
#endif
  launch_trsm(HBO_F64, d, 1, 0, nblk, 0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_trsm(HBO_F64, d, 1, 0, nblk, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("trsm (%d rows)  %8.2f us per launch\n", n + 128 - 128, ms / reps * 1e3);
  launch_trtri_diag(HBO_F64, d, 1, 0, nblk, 0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_trtri_diag(HBO_F64, d, 1, 0, nblk, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("trtri_diag       %8.2f us per launch\n", ms / reps * 1e3);
  CK(hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost)); printf("info %d\n", inf);
  return 0;
}
