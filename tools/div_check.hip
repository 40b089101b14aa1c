// Exhaustive check (every pair of FP32 significands, 2^46 quotients) of the shared-reciprocal division
//   r = RN(1/b) (v_rcp_f32 + one Newton step, rt_math.h rcp_exact);  q0 = a * r;  e = fma(-b, q0, a);  q = fma(e, r, q0)
// against the correctly rounded a / b.  Exponents do not matter while nothing under- or overflows (every step scales with
// them), so a and b run over [1, 2).  build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/div_check.hip -o tools/div_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_newton1(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__global__ __launch_bounds__(256) void k(unsigned long long* out, unsigned int mb_begin, unsigned int mb_count) {
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= mb_count) return;
  const unsigned int mb = mb_begin + t;
  const float b = __uint_as_float(0x3f800000u | mb);
  const float r = rcp_newton1(b);
  unsigned long long bad = 0;
  for (unsigned int ma = 0; ma < (1u << 23); ++ma) {
    const float a = __uint_as_float(0x3f800000u | ma);
    const float ref = a / b;
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    const float q = __builtin_fmaf(e, r, q0);
    if (__float_as_uint(q) != __float_as_uint(ref)) {
      ++bad;
      const unsigned long long slot = atomicAdd(&out[1], 1ull);
      if (slot < 32) out[8 + slot] = ((unsigned long long)ma << 32) | mb;
    }
  }
  if (bad) atomicAdd(&out[0], bad);
}
int main(int argc, char** argv) {
  unsigned long long* d = nullptr;
  hipMalloc(&d, 64 * 8);
  hipMemset(d, 0, 64 * 8);
  const unsigned int total = 1u << 23, step = 1u << 19;      // 16 launches, a progress line each
  for (unsigned int begin = 0; begin < total; begin += step) {
    hipLaunchKernelGGL(k, dim3(step / 256), dim3(256), 0, 0, d, begin, step);
    unsigned long long h[64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("b significands [%u, %u): mismatches so far %llu\n", begin, begin + step, h[0]);
    fflush(stdout);
  }
  unsigned long long h[64];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("pairs 2^46, mismatches %llu\n", h[0]);
  for (int i = 0; i < 32 && i < (int)h[1]; ++i) printf("  a significand %06llx  b significand %06llx\n", h[8 + i] >> 32, h[8 + i] & 0xffffffffull);
  return 0;
}
