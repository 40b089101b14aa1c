// rt_selftest.hip — on-device self tests of the numerics building blocks (C ABI: rt_selftest_rcp).
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_math.h"

namespace uobrt {
void set_error(const char* fmt, ...);

// For every FP32 bit pattern x: compare the Newton-refined v_rcp_f32 (1 and 2 steps) with the correctly
// rounded 1.0f/x.  out[0]/out[1]: mismatches of the 1-/2-step form over the "safe" magnitudes
// 2^-100 <= |x| <= 2^100 (far from the flush/overflow ends); out[2]/out[3]: mismatches over every other
// finite non-zero x; out[4]: number of recorded examples; out[8..]: up to 56 mismatching patterns of
// the 1-step form inside the safe range (low 32 bits = x).
__global__ __launch_bounds__(256) void k_selftest_rcp(unsigned long long* out) {
  const unsigned long long total = 1ull << 32;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad1 = 0, bad2 = 0, obad1 = 0, obad2 = 0;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
    const uint32_t bits = (uint32_t)k;
    const float x = __uint_as_float(bits);
    const uint32_t mag = bits & 0x7fffffffu;
    if (mag == 0u || mag >= 0x7f800000u) continue;   // zero, inf, NaN
    const float ref = 1.0f / x;
    const float r1 = rcp_newton(x, 1), r2 = rcp_newton(x, 2);
    const bool safe = mag >= 0x0d800000u && mag <= 0x71800000u;   // 2^-100 .. 2^100
    const bool m1 = __float_as_uint(r1) != __float_as_uint(ref);
    const bool m2 = __float_as_uint(r2) != __float_as_uint(ref);
    if (safe) {
      bad1 += m1; bad2 += m2;
      if (m1) { const unsigned long long slot = atomicAdd(&out[4], 1ull); if (slot < 56) out[8 + slot] = bits; }
    } else { obad1 += m1; obad2 += m2; }
  }
  if (bad1) atomicAdd(&out[0], bad1);
  if (bad2) atomicAdd(&out[1], bad2);
  if (obad1) atomicAdd(&out[2], obad1);
  if (obad2) atomicAdd(&out[3], obad2);
}
}  // namespace uobrt

extern "C" int rt_selftest_rcp(uint64_t out[64]) {
  using namespace uobrt;
  if (!out) { set_error("NULL argument"); return RT_E_INVALID; }
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 64 * 8) != hipSuccess) { set_error("hipMalloc failed (no device?)"); return RT_E_DEVICE; }
  hipMemset(d, 0, 64 * 8);
  hipLaunchKernelGGL(k_selftest_rcp, dim3(16384), dim3(256), 0, 0, d);
  const hipError_t e = hipMemcpy(out, d, 64 * 8, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) { set_error("rt_selftest_rcp: %s", hipGetErrorString(e)); return RT_E_DEVICE; }
  return RT_OK;
}
