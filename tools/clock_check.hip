// Diagnostic: rate of s_memtime against s_memrealtime (100 MHz) and the wall clock.   hipcc --offload-arch=gfx950 -o clock_check clock_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int spin) {
  const unsigned long long a = __builtin_amdgcn_s_memtime(), ra = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001f + 0.5f;
  const unsigned long long b = __builtin_amdgcn_s_memtime(), rb = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = b - a; out[1] = rb - ra; out[2] = (unsigned long long)x; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 4000000); hipEventRecord(e1); hipEventSynchronize(e1);
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.3f ms: s_memtime %llu ticks (%.1f MHz), s_memrealtime %llu ticks (%.1f MHz)\n", ms, h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
  }
  return 0;
}
