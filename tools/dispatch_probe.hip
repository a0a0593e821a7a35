// Workgroup dispatch rate on gfx950: time an (almost) empty kernel of W workgroups of 256 threads for several LDS
// sizes and register footprints.  Build: hipcc -O3 --offload-arch=gfx950 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNUSED>
__global__ void __launch_bounds__(256) probe(int* out, int live_every) {
  extern __shared__ unsigned char smem[];
  if (live_every > 0 && (blockIdx.x % live_every) == 0) {
    // a "live" workgroup: ~20 us of dependent work
    double v = threadIdx.x;
    for (int i = 0; i < 4000; ++i) v = v * 1.0000001 + 1e-9;
    if (v == 12345.678) out[0] = 1;
  }
}
int main() {
  int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int Ws[] = {1024, 16384, 65536, 262144};
  const int ldss[] = {0, 36 * 1024, 72 * 1024};
  for (int lds : ldss)
    for (int W : Ws) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<0>, dim3(W), dim3(256), lds, 0, d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("empty   lds %6d  W %7d : %8.1f us  -> %6.1f ns / workgroup\n", lds, W, ms * 1e3, ms * 1e6 / W);
    }
  // mixed: one live workgroup in every `every`
  for (int every : {1, 2, 4, 8, 16}) {
    const int W = 65536;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<0>, dim3(W), dim3(256), 36 * 1024, 0, d, every);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("1 live in %2d, lds 36K, W %d : %8.1f us (%d live)\n", every, W, ms * 1e3, W / every);
  }
  return 0;
}
