// Probe: can a running kernel release another stream through hipStreamWaitValue64 on signal memory?  Measures the
// latency from the device-side atomic to the start of the dependent kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void producer(unsigned long long* sig, unsigned long long* stamps, int spin_us) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
  __threadfence();
  stamps[0] = wall_clock64();
  atomicAdd(sig, 5ull);
  // keep running: the consumer must start while this kernel is still resident
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 300) {}
  stamps[2] = wall_clock64();
}
__global__ void consumer(unsigned long long* stamps) { stamps[1] = wall_clock64(); }
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  unsigned long long* sig = nullptr; unsigned long long* stamps = nullptr;
  CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
  CK(hipMalloc((void**)&stamps, 64));
  hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
  for (int it = 0; it < 5; ++it) {
    CK(hipMemset(stamps, 0, 64));
    unsigned long long zero = 0; CK(hipMemcpy(sig, &zero, 8, hipMemcpyHostToDevice));
    CK(hipStreamWaitValue64(a, sig, 5, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, a, stamps);
    hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, b, sig, stamps, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[3]; CK(hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost));
    printf("signal -> consumer start: %.2f us   (producer still ran %.2f us after the consumer started)\n",
           (double)(long long)(h[1] - h[0]) / 100.0, (double)(long long)(h[2] - h[1]) / 100.0);
  }
  return 0;
}
