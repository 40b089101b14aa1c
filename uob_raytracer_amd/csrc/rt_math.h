// rt_math.h — scalar float3 helpers for the gfx950 kernels.
//
// Numerics contract: every translation unit that includes this header is compiled with
// -ffp-contract=off and without fast-math, so each '*', '+', '-' below is ONE correctly rounded FP32
// operation in the order written; '/' and sqrtf are correctly rounded (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).  The operation order follows the reference kernel
// (Source/kernels.cl) so that results are bit-identical with the strict oracle.
#pragma once
#include <hip/hip_runtime.h>

namespace uobrt {

struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 xyz(float4 v) { return f3{v.x, v.y, v.z}; }

// OpenCL dot as the oracle defines it: x*x + y*y + z*z, left to right
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// OpenCL normalize as the oracle defines it: v / sqrtf(dot(v,v))
// The 2x2 cofactors of rows (m1, m2) used by det (kernels.cl:31-35):
//   det(m0,m1,m2) = m0.x*c.x - m0.y*c.y + m0.z*c.z   with c = cof(m1,m2)
__device__ __forceinline__ f3 cof(f3 m1, f3 m2) {
  return f3{m1.y * m2.z - m1.z * m2.y, m1.x * m2.z - m1.z * m2.x, m1.x * m2.y - m1.y * m2.x};
}
__device__ __forceinline__ float detc(f3 m0, f3 c) { return m0.x * c.x - m0.y * c.y + m0.z * c.z; }

// v_rcp_f32 (1 ulp) refined by Newton steps in FMA arithmetic; rcp_exact below is the form the kernels use
__device__ __forceinline__ float rcp_newton(float x, int steps) {
  float r = __builtin_amdgcn_rcpf(x);
  for (int k = 0; k < steps; ++k) {
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  return r;
}

// The correctly rounded 1/x at a third of the cost of IEEE division (13 vs 39 issue cycles, profiles/
// r01_valu_issue_cost_8waves.txt): v_rcp_f32 + one Newton step in FMA arithmetic equals 1.0f/x bit for bit for
// EVERY x with 2^-126 <= |x| < 2^126 (rt_selftest_rcp sweeps all 2^32 patterns on the GPU; the only
// mismatches are denormal x and |x| >= 2^126).  For zero, denormal, infinite or NaN x the refinement
// returns NaN, which is the cue to fall back to the division; |x| >= 2^126 cannot occur for scenes that
// pass rt_init's coordinate bound (|coordinate| <= 2^16, rt_device.h kMaxCoordinate).
// The fallback sits behind a WAVE-UNIFORM branch (a ballot): written as a plain per-lane `if`, the compiler computes
// the division for every call and selects (12 more instructions per triangle test).
__device__ __forceinline__ float rcp_exact(float x) {
  float r = rcp_newton(x, 1);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(r != r) != 0ull, 0)) {
    if (r != r) r = 1.0f / x;
  }
  return r;
}

// normalize(): a / sqrt(a.a), the square root and the three quotients correctly rounded (the oracle's normalize3, strict C).
// IEEE sqrt + three IEEE divisions are ~160 issue cycles on this GPU; the same bits come from
//   len = v_rsq_f32 refined once in FMA arithmetic   (== sqrtf(x) for EVERY x in [2^-60, 2^60]: tools/sqrt_check.hip, all 2^32 patterns)
//   r   = RN(1 / len)                                 (rcp_newton, exact for every normal len: rt_selftest_rcp)
//   q0 = a * r,  e = fma(-len, q0, a),  q = fma(e, r, q0)      (== a / len for EVERY pair of significands: tools/div_check.hip, all 2^46
//        pairs, no mismatch — every step scales with the exponents, so the check holds wherever nothing under- or overflows:
//        len in [2^-30, 2^30], |q0| in [2^-30, 1], the residual e then a normal number or exactly 0)
// in ~70.  Anything else — a zero or tiny component, a NaN, a vector shorter than 2^-30 or longer than 2^30 — takes the IEEE
// operations behind a wave-uniform branch, as rcp_exact does.
__device__ __forceinline__ f3 normalize3(f3 a) {
  const float x = a.x * a.x + a.y * a.y + a.z * a.z;
  const float y = __builtin_amdgcn_rsqf(x);
  const float s = x * y, h = 0.5f * y;
  const float len = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  const float r = rcp_newton(len, 1);
  const float q0x = a.x * r, q0y = a.y * r, q0z = a.z * r;
  f3 q = f3{__builtin_fmaf(__builtin_fmaf(-len, q0x, a.x), r, q0x), __builtin_fmaf(__builtin_fmaf(-len, q0y, a.y), r, q0y),
            __builtin_fmaf(__builtin_fmaf(-len, q0z, a.z), r, q0z)};
  const float mn = fminf(fminf(fabsf(q0x), fabsf(q0y)), fabsf(q0z));
  const bool ok = x >= 0x1p-60f && x <= 0x1p60f && mn >= 0x1p-30f;        // (false for NaN)
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
    if (!ok) {
      const float l2 = sqrtf(x);
      q = f3{a.x / l2, a.y / l2, a.z / l2};
    }
  }
  return q;
}


// x / (float)n for a wave-uniform count n (kernels.cl:338 total / light_sources, :426 final / aa_rays).  When n is a
// power of two the host passes inv = 1/n (exact) and the quotient is the product: both are the correctly rounded
// value of the same real number x * 2^-k, for every x (denormal results included).  inv == 0: IEEE division.
__device__ __forceinline__ float div_count(float x, int n, float inv) {
  return inv != 0.0f ? x * inv : x / (float)n;
}

// xorshift32 (kernels.cl:42-47), one component
__device__ __forceinline__ uint32_t xorshift(uint32_t s) {
  s ^= s << 13;
  s ^= s >> 17;
  s ^= s << 5;
  return s;
}
// crush (kernels.cl:49-52), one component; (float)UINT_MAX rounds to 2^32
__device__ __forceinline__ float crush1(uint32_t v, float range) {
  return range * (float)v / 4294967296.0f - range / 2.f;
}

}  // namespace uobrt
