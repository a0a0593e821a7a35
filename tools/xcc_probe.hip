// Which XCD does workgroup i of a small launch run on?  Prints XCC_ID per workgroup for a few consecutive launches of
// 8 and 12 workgroups, on one stream and alternating between two streams.
// Build: hipcc -O3 --offload-arch=gfx950 tools/xcc_probe.hip -o tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // XCC_ID[3:0]
}
int main() {
  int* d; hipMalloc(&d, 64 * 4); int h[64];
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int rep = 0; rep < 6; ++rep) {
    const int n = rep < 3 ? 8 : 12;
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, s1, d);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    printf("stream 1, %2d workgroups:", n); for (int i = 0; i < n; ++i) printf(" %d", h[i]); printf("\n");
  }
  for (int rep = 0; rep < 4; ++rep) {
    hipStream_t s = (rep & 1) ? s2 : s1;
    hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h, d, 8 * 4, hipMemcpyDeviceToHost);
    printf("stream %d,  8 workgroups:", (rep & 1) + 1); for (int i = 0; i < 8; ++i) printf(" %d", h[i]); printf("\n");
  }
  return 0;
}
