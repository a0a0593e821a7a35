// Micro-benchmark of the tile-GEMM modes (links the library objects directly).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip hyperbo_amd/csrc/gemm.o -o tools/gemm_bench
#include "../hyperbo_amd/csrc/hbo_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill(double* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 13; h *= 0x5bd1e995; h ^= h >> 15; p[i] = ((h & 0xffff) / 65536.0 - 0.5) * 0.01; }
}

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 8192;
  int nblk = n / 128; int64_t ld = n + 16;
  // HBO_BENCH_LD=<elements>: override the leading dimension (e.g. 16: every row aliases the same 128 bytes -> all operand
  // traffic hits in L1/L2: the compute-bound speed of a kernel, numerically meaningless)
  const char* ldenv = getenv("HBO_BENCH_LD");
  double *A, *W, *S;
  size_t na = (size_t)(n + 128) * ld;
  CK(hipMalloc(&A, na * 8)); CK(hipMalloc(&W, na * 8)); CK(hipMalloc(&S, na * 8));
  fill<<<(na + 255) / 256, 256>>>(A, na, 1); fill<<<(na + 255) / 256, 256>>>(W, na, 2); fill<<<(na + 255) / 256, 256>>>(S, na, 3);
  TaskDesc h = {}; h.A = A; h.W = W; h.S = S; h.n = n; h.npad = n; h.nblk = nblk; h.m = 1; h.ld = ldenv ? atoll(ldenv) : ld;
  TaskDesc* d; CK(hipMalloc(&d, sizeof h)); CK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, GemmArgs a, dim3 grid, double flops) {
    launch_gemm(HBO_F64, a, grid, 0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch_gemm(HBO_F64, a, grid, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-36s %8.3f ms  %6.1f TFLOP/s (algorithmic)\n", name, ms, flops / ms / 1e9);
  };
  if (argc > 2) {  // single-kernel mode for PMC runs: syrk K=1024 without C traffic
    GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = 0; a.kt = 8; a.c_lo = 8; a.c_hi = nblk; a.aug = 1 | 4;
    double m = nblk - 8;
    timeit("syrk K=1024 dbg=4", a, dim3(nblk + 1 - 8, nblk - 8, 1), m * (m + 1) / 2 * 128.0 * 128 * 2 * 128 * 8);
    return 0;
  }
  if (argc > 1 && getenv("HBO_BENCH_CHAIN")) {   // the panel chain's own GEMM launches, alone on the machine (latency, not throughput)
    {
    for (int p : {6, 30, 50})
      for (int kt : {1, 2}) {
        GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = p - kt; a.kt = kt; a.c_lo = p; a.c_hi = p + 1; a.aug = 1; a.small_tiles = 1;
        char nm[64]; snprintf(nm, 64, "column update p=%d K=%d", p, kt * 128);
        timeit(nm, a, dim3(nblk + 1 - p, 1, 1), (double)(nblk + 1 - p) * 128.0 * 128 * 2 * 128 * kt);
      }
    for (int g1 : {6, 30, 51}) {
      GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = g1 - 3; a.kt = 3; a.c_lo = g1; a.c_hi = g1 + 3; a.aug = 1;
      a.small_tiles = (int64_t)(nblk + 1 - a.c_lo) * 3 < 600;
      char nm[64]; snprintf(nm, 64, "F1 g1=%d K=384", g1);
      timeit(nm, a, dim3(nblk + 1 - g1, 3, 1), (double)(3 * (nblk + 1 - g1) - 3) * 128.0 * 128 * 2 * 384);
    }
    }
    return 0;
  }
  { // the tile core alone: a full square of 128-tiles with one K for all (C -= V^T V, the form of rocBLAS dgemm NT n^3): no
    // triangle, no tail beyond tiles mod slots
    GemmArgs a = {}; a.tasks = d; a.mode = GEMM_VTV; a.B = W; a.ldb = h.ld; a.V = S;
    timeit("square n^3 (VTV)", a, dim3(nblk, nblk, 1), 2.0 * n * n * n); }
  for (int kt : {2, 8}) {   // the 64-tile core: the same update on 64 x 64 tiles, no C traffic
    GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = 0; a.kt = kt; a.c_lo = kt; a.c_hi = nblk; a.aug = 1 | 4; a.small_tiles = 1;
    double m = nblk - kt;
    char nm[64]; snprintf(nm, 64, "syrk 64-tiles K=%d dbg=4", kt * 128);
    timeit(nm, a, dim3(nblk + 1 - kt, nblk - kt, 1), m * (m + 1) / 2 * 128.0 * 128 * 2 * 128 * kt);
  }
  if (getenv("HBO_BENCH_SQUARE_ONLY")) return 0;
  for (int dbg : {0, 2, 4})
  for (int kt : {1, 2, 4, 8}) {
    GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = 0; a.kt = kt; a.c_lo = kt; a.c_hi = nblk; a.aug = 1 | dbg;
    double m = nblk - kt;
    char nm[64]; snprintf(nm, 64, "syrk K=%d dbg=%d", kt * 128, dbg);
    timeit(nm, a, dim3(nblk + 1 - kt, nblk - kt, 1), m * (m + 1) / 2 * 128.0 * 128 * 2 * 128 * kt);
  }
  { // the shipped bulk trailing update: persistent, tiles from a counter, K = 384 / 512 over the trailing matrix after 6 / 24 panels
    int* ctr; CK(hipMalloc(&ctr, 64 * sizeof(int)));
    for (int pf : {32, 0})
    for (int kt : {3, 4})
    for (int c0 : {6, 24}) {
      GemmArgs a = {}; a.tasks = d; a.mode = GEMM_SYRK; a.p0 = c0 - kt; a.kt = kt; a.c_lo = c0; a.c_hi = nblk; a.aug = 1;
      a.persistent = pf ? 2 * (256 - pf) : 0; a.work_counter = pf ? ctr : nullptr;
      double m = nblk - c0;
      char nm[64]; snprintf(nm, 64, "bulk K=%d m=%d %s", kt * 128, (int)m, pf ? "persistent" : "plain");
      float best = 1e9;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 64 * sizeof(int), 0));
        CK(hipEventRecord(e0)); launch_gemm(HBO_F64, a, dim3(nblk + 1 - c0, nblk - c0, 1), 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      printf("%-36s %8.3f ms  %6.1f TFLOP/s\n", nm, best, (m * (m + 1) / 2 + m) * 128.0 * 128 * 2 * 128 * kt / best / 1e9);
    }
  }
  { GemmArgs a = {}; a.tasks = d; a.mode = GEMM_LAUUM;
    timeit("lauum", a, dim3(nblk, nblk, 1), (double)n * n * n / 3);
    // the same tiles drawn from a counter by a resident grid
    int* ctr; CK(hipMalloc(&ctr, sizeof(int)));
    for (int slots : {512, 480, 448}) {
      a.persistent = slots; a.work_counter = ctr;
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(ctr, 0, sizeof(int), 0));
        CK(hipEventRecord(e0)); launch_gemm(HBO_F64, a, dim3(nblk, nblk, 1), 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      printf("lauum, %d resident workgroups          %8.3f ms  %6.1f TFLOP/s\n", slots, best, (double)n * n * n / 3 / best / 1e9);
    }
  }
  { double tot = 0; float msum = 0;
    for (int s = 1; s < nblk; s *= 2) {
      int ng = (nblk + 2 * s - 1) / (2 * s);
      GemmArgs a = {}; a.tasks = d; a.p0 = s; a.c_hi = ng - 1; a.c_lo = s;   // (the last group and its row count: full)
      for (int mode : {GEMM_TRTRI_A, GEMM_TRTRI_B}) {
        a.mode = mode;
        launch_gemm(HBO_F64, a, dim3(ng * s, s, 1), 0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch_gemm(HBO_F64, a, dim3(ng * s, s, 1), 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); msum += ms;
        printf("  trtri s=%2d mode %d: %.3f ms\n", s, mode, ms);
      }
    }
    printf("%-36s %8.3f ms  %6.1f TFLOP/s (algorithmic, N^3/3 minus diag blocks)\n", "trtri levels", msum, (double)n * n * n / 3 / msum / 1e9); }
  return 0;
}
