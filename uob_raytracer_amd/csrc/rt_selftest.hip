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
// normalize3's building blocks (rt_math.h): (a) the refined v_rsq_f32 against sqrtf for every FP32 pattern; (b) the quotient
// from the shared exact reciprocal against a / b for every significand of a and every `stride`-th significand of b (a, b in
// [1, 2): exponents do not matter while nothing under- or overflows).  out[0] = mismatches of (a) for 2^-60 <= x <= 2^60,
// out[1] = mismatches of (b), out[2] = pairs checked by (b), out[3] = a mismatching pattern of (a), out[4] = one of (b).
__global__ __launch_bounds__(256) void k_selftest_sqrt(unsigned long long* out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const uint32_t bits = (uint32_t)i;
    if (bits < 0x21800000u || bits > 0x5d800000u) continue;              // 2^-60 .. 2^60
    const float x = __uint_as_float(bits);
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    const float len = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    if (__float_as_uint(len) != __float_as_uint(sqrtf(x))) { ++bad; out[3] = bits; }
  }
  if (bad) atomicAdd(&out[0], bad);
}
__global__ __launch_bounds__(256) void k_selftest_div(unsigned long long* out, unsigned int stride, unsigned int count) {
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  const unsigned int mb = t * stride;
  const float b = __uint_as_float(0x3f800000u | mb);
  const float r = rcp_newton(b, 1);
  unsigned long long bad = 0;
  for (unsigned int ma = 0; ma < (1u << 23); ++ma) {
    const float a = __uint_as_float(0x3f800000u | ma);
    const float q0 = a * r;
    const float q = __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
    if (__float_as_uint(q) != __float_as_uint(a / b)) { ++bad; out[4] = ((unsigned long long)ma << 32) | mb; }
  }
  if (bad) atomicAdd(&out[1], bad);
}
}  // namespace uobrt

extern "C" int rt_selftest_normalize(uint64_t out[8], uint32_t b_stride) {
  using namespace uobrt;
  if (!out || b_stride == 0 || b_stride > (1u << 23)) { set_error("rt_selftest_normalize: NULL argument or stride outside 1 .. 2^23"); return RT_E_INVALID; }
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 8 * 8) != hipSuccess) { set_error("hipMalloc failed (no device?)"); return RT_E_DEVICE; }
  hipMemset(d, 0, 8 * 8);
  const unsigned int count = ((1u << 23) + b_stride - 1) / b_stride;
  hipLaunchKernelGGL(k_selftest_sqrt, dim3(16384), dim3(256), 0, 0, d);
  hipLaunchKernelGGL(k_selftest_div, dim3((count + 255) / 256), dim3(256), 0, 0, d, b_stride, count);
  const hipError_t e = hipMemcpy(out, d, 8 * 8, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) { set_error("rt_selftest_normalize: %s", hipGetErrorString(e)); return RT_E_DEVICE; }
  out[2] = (uint64_t)count << 23;
  return RT_OK;
}

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
