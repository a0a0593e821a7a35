// Probe: fp64/fp32 MFMA layout self-test + throughput microbenchmark on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// C(16x16) = A(16x4) * B(4x16), one wave.
__global__ void layout_f64(const double* A, const double* B, double* C) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[i=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][j=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
__global__ void layout_f32(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void bench_f64(double* out, int iters) {
  d4 acc[NACC];
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void bench_f32(float* out, int iters) {
  f4 acc[NACC];
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// fp64 VALU FMA throughput
__global__ void bench_fma64(double* out, int iters) {
  double x[16];
  double a = 1.0000001, b = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
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
  printf("device %s arch %s CUs %d clock %d kHz mem %.1f GB\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9);
  // layout test
  {
    std::vector<double> A(64), B(64), C(256), R(256, 0);
    for (int i = 0; i < 64; ++i) { A[i] = 0.5 + i * 0.37 + (i % 5); B[i] = 1.25 - i * 0.11 + (i % 7) * (i % 3); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_f64<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]));
    printf("layout_f64 max err %.3e %s\n", err, err < 1e-9 ? "OK" : "FAIL");
    std::vector<float> Af(64), Bf(64), Cf(256);
    for (int i = 0; i < 64; ++i) { Af[i] = (float)A[i]; Bf[i] = (float)B[i]; }
    float *fA = (float*)dA, *fB = (float*)dB, *fC = (float*)dC;
    CK(hipMemcpy(fA, Af.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(fB, Bf.data(), 256, hipMemcpyHostToDevice));
    layout_f32<<<1, 64>>>(fA, fB, fC); CK(hipMemcpy(Cf.data(), fC, 1024, hipMemcpyDeviceToHost));
    err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(Cf[i] - R[i]) / fabs(R[i]));
    printf("layout_f32 max rel err %.3e %s\n", err, err < 1e-5 ? "OK" : "FAIL");
  }
  double* out; CK(hipMalloc(&out, 256 * 8 * 1024 * 8 * 2));
  int iters = 2000;
  int cus = p.multiProcessorCount;
  for (int wpc : {4, 8, 16}) {
    int blocks = cus, threads = 64 * wpc;
    if (threads > 1024) { blocks = cus * (threads / 1024); threads = 1024; }
    {
      double ms = time_ms([&] { bench_f64<1><<<blocks, threads>>>(out, iters); }, 5);
      double fl = (double)blocks * (threads / 64) * iters * 1 * 2048.0;
      printf("f64 mfma 16x16x4 NACC=1 waves/CU=%d: %.3f ms, %.1f TFLOP/s, %.1f cyc/mfma/SIMD@2.4GHz\n", wpc, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 1.0 * wpc / 4));
    }
    {
      double ms = time_ms([&] { bench_f64<4><<<blocks, threads>>>(out, iters); }, 5);
      double fl = (double)blocks * (threads / 64) * iters * 4 * 2048.0;
      printf("f64 mfma 16x16x4 NACC=4 waves/CU=%d: %.3f ms, %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { bench_f64<16><<<blocks, threads>>>(out, iters); }, 5);
      double fl = (double)blocks * (threads / 64) * iters * 16 * 2048.0;
      printf("f64 mfma 16x16x4 NACC=16 waves/CU=%d: %.3f ms, %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { bench_f32<4><<<blocks, threads>>>((float*)out, iters); }, 5);
      double fl = (double)blocks * (threads / 64) * iters * 4 * 2048.0;
      printf("f32 mfma 16x16x4 NACC=4 waves/CU=%d: %.3f ms, %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { bench_fma64<<<blocks, threads>>>(out, iters); }, 5);
      double fl = (double)blocks * threads * iters * 16 * 2.0;
      printf("f64 valu fma waves/CU=%d: %.3f ms, %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    }
  }
  // sustained fp64 MFMA throughput (DVFS): back-to-back launches for ~0.5 s, report per-window TF
  {
    int blocks = cus * 2, threads = 256;  // 2 waves per SIMD
    int it2 = 20000;                      // ~ 20000*4*64 cycles = 2.1 ms per launch at 2.4 GHz
    hipEvent_t ev[64]; for (int i = 0; i < 64; ++i) CK(hipEventCreate(&ev[i]));
    CK(hipEventRecord(ev[0]));
    for (int w = 1; w < 64; ++w) {
      for (int r = 0; r < 4; ++r) bench_f64<4><<<blocks, threads>>>(out, it2);
      CK(hipEventRecord(ev[w]));
    }
    CK(hipDeviceSynchronize());
    printf("sustained f64 mfma (TF per ~10ms window):");
    for (int w = 1; w < 64; ++w) {
      float ms; CK(hipEventElapsedTime(&ms, ev[w - 1], ev[w]));
      double fl = 4.0 * blocks * (threads / 64) * (double)it2 * 4 * 2048.0;
      if (w % 4 == 1) printf(" %.1f", fl / ms / 1e9);
    }
    printf("\n");
  }
  // simple HBM copy bandwidth
  {
    size_t n = (size_t)1 << 30; double *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n));
    double ms = time_ms([&] { CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0)); }, 5);
    printf("D2D memcpy 1 GiB: %.3f ms -> %.2f TB/s (r+w)\n", ms, 2.0 * n / ms / 1e9);
    ms = time_ms([&] { CK(hipMemsetAsync(b, 0, n, 0)); }, 5);
    printf("memset 1 GiB: %.3f ms -> %.2f TB/s (w)\n", ms, 1.0 * n / ms / 1e9);
  }
  return 0;
}
