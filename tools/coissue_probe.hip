// Probe: do fp64 MFMA and fp64 VALU FMA execute concurrently on gfx950?
// (both run at 32 flop/clk/SIMD; if they are separate pipes a hybrid GEMM could exceed the MFMA peak)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// mode bit 0: MFMA waves active, bit 1: VALU waves active.  Waves [0, nm) do MFMA, [nm, nm+nv) do VALU.
template <int NV>
__global__ void __launch_bounds__(1024) split_waves(double* out, int iters, int nm, int mode) {
  int w = threadIdx.x >> 6;
  double s = 0;
  if (w < nm) {
    if (mode & 1) {
      d4 acc[4];
      double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode & 2) {
      double x[NV];
      double a = 1.0000001, b = 1e-9 * threadIdx.x;
#pragma unroll
      for (int i = 0; i < NV; ++i) x[i] = i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = __builtin_fma(x[i], a, b);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) s += x[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave interleaves: per iteration 4 MFMAs (4 accumulators) and 4*R VALU fmas (independent chains)
template <int R>
__global__ void __launch_bounds__(256) inline_mix(double* out, int iters) {
  d4 acc[4];
  double x[4 * R > 0 ? 4 * R : 1];
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double fa = 1.0000001, fb = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) x[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) x[i * R + r] = __builtin_fma(x[i * R + r], fa, fb);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, (size_t)cus * 4 * 1024 * 8));
  int iters = 4000;
  printf("split waves: nm MFMA waves + nv VALU waves per CU (one workgroup per CU)\n");
  for (int nm : {4, 8}) for (int nv : {4, 8}) {
    int threads = 64 * (nm + nv);
    double fm = (double)cus * nm * iters * 4 * 2048.0, fv = (double)cus * nv * 64 * iters * 16 * 2.0;
    double t1 = time_ms([&] { split_waves<16><<<cus, threads>>>(out, iters, nm, 1); }, 5);
    double t2 = time_ms([&] { split_waves<16><<<cus, threads>>>(out, iters, nm, 2); }, 5);
    double t3 = time_ms([&] { split_waves<16><<<cus, threads>>>(out, iters, nm, 3); }, 5);
    printf("nm=%d nv=%d: mfma only %.3f ms (%.1f TF)  valu only %.3f ms (%.1f TF)  both %.3f ms (%.1f TF total; sum of times %.3f)\n",
           nm, nv, t1, fm / t1 / 1e9, t2, fv / t2 / 1e9, t3, (fm + fv) / t3 / 1e9, t1 + t2);
  }
  printf("inline mix: one wave per SIMD / two waves per SIMD, 4 MFMA + 4R VALU fma per iteration\n");
#define RUN(R) \
  for (int wg : {1, 2}) { \
    double t = time_ms([&] { inline_mix<R><<<cus * wg, 256>>>(out, iters); }, 5); \
    double fm = (double)cus * wg * 4 * iters * 4 * 2048.0, fv = (double)cus * wg * 256 * iters * 4.0 * R * 2.0; \
    printf("R=%d wg/CU=%d: %.3f ms, mfma %.1f TF + valu %.1f TF = %.1f TF, %.1f cyc per MFMA slot\n", R, wg, t, fm / t / 1e9, fv / t / 1e9, (fm + fv) / t / 1e9, \
           t * 1e-3 * 2.4e9 / (iters * 4.0 * wg)); \
  }
  RUN(0) RUN(2) RUN(4) RUN(8) RUN(12) RUN(16)
  return 0;
}
