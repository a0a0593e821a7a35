// Which LDS sizes co-reside on a CU?  256 spinning workgroups with X bytes (X above half the LDS: one per CU, every CU taken), then
// ONE probe workgroup with Y bytes on a second stream: it starts at once if X + Y (each rounded to the allocation granule) fit in
// 160 KB, otherwise when the spinners leave (20 ms).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_granule_probe.hip -o tools/lds_granule_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }
__global__ __launch_bounds__(256) void spin(unsigned long long* t, unsigned long long ticks) {
  extern __shared__ int s[];
  s[threadIdx.x] = 1;
  const unsigned long long t0 = wall();
  if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = t0;
  while (wall() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (s[threadIdx.x] == 7) t[3] = 1;
}
__global__ __launch_bounds__(256) void probe(unsigned long long* t) {
  extern __shared__ int s[];
  s[threadIdx.x] = 1;
  if (threadIdx.x == 0) t[1] = wall();
  if (s[threadIdx.x] == 7) t[3] = 1;
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  unsigned long long* t; hipHostMalloc(&t, 64);
  hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
  const int xs[] = {81984, 82432, 82944, 83200};
  const int ys[] = {74240, 79872, 80384, 80512, 80640, 80896, 81408, 81536, 81856};
  for (int x : xs) {
    printf("spinners %6d bytes: probe starts after (ms)", x);
    for (int y : ys) {
      t[0] = t[1] = 0;
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), x, a, t, 2000000ull);   // 20 ms at 100 MHz
      hipLaunchKernelGGL(probe, dim3(1), dim3(256), y, b, t);
      hipDeviceSynchronize();
      printf("  %d: %.2f", y, (double)(long long)(t[1] - t[0]) / 1e5);
    }
    printf("\n");
  }
  return 0;
}
