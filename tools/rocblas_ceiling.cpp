// What a vendor-tuned fp64 GEMM reaches on this device at the shapes of the hot path -- a measurement of the ceiling for
// hyperbo_amd/csrc/gemm.hip, NOT part of the product (libhbo links no BLAS).
//   hipcc --offload-arch=gfx950 -O2 tools/rocblas_ceiling.cpp -lrocblas -o tools/rocblas_ceiling
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
  rocblas_handle h; rocblas_create_handle(&h);
  const int N = 8192;
  double *A, *B, *C; size_t bytes = (size_t)N * N * 8;
  CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&C, bytes));
  CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(C, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double alpha = -1.0, beta = 1.0;
  auto run = [&](const char* name, int m, int n, int k, rocblas_operation ta, rocblas_operation tb, int reps) {
    const int lda = N, ldb = N, ldc = N;
    rocblas_dgemm(h, ta, tb, m, n, k, &alpha, A, lda, B, ldb, &beta, C, ldc); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) rocblas_dgemm(h, ta, tb, m, n, k, &alpha, A, lda, B, ldb, &beta, C, ldc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-44s m=%5d n=%5d k=%5d  %8.3f ms  %6.1f TFLOP/s\n", name, m, n, k, ms, 2.0 * m * n * k / ms / 1e9);
  };
  // column-major views: our row-major C -= P P^T with k-contiguous rows of P is  C^T -= (P^T)^T (P^T)  = op(A)=T, op(B)=N
  run("rank-384 update, both operands k-contiguous (TN)", 7424, 7424, 384, rocblas_operation_transpose, rocblas_operation_none, 10);
  run("rank-512 update (TN)", 7424, 7424, 512, rocblas_operation_transpose, rocblas_operation_none, 10);
  run("rank-384 update (NT)", 7424, 7424, 384, rocblas_operation_none, rocblas_operation_transpose, 10);
  run("square 8192 (TN)", N, N, N, rocblas_operation_transpose, rocblas_operation_none, 3);
  run("square 8192 (NN)", N, N, N, rocblas_operation_none, rocblas_operation_none, 3);
  run("square 8192 (NT)", N, N, N, rocblas_operation_none, rocblas_operation_transpose, 3);
  run("square 4096 (NT)", 4096, 4096, 4096, rocblas_operation_none, rocblas_operation_transpose, 5);
  {
    rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_transpose, 7424, 384, &alpha, A, N, &beta, C, N); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_transpose, 7424, 384, &alpha, A, N, &beta, C, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s n=%5d k=%5d          %8.3f ms  %6.1f TFLOP/s (n^2 k flops)\n", "dsyrk lower, rank 384", 7424, 384, ms, 1.0 * 7424 * 7424 * 384 / ms / 1e9);
  }
  rocblas_destroy_handle(h);
  return 0;
}
