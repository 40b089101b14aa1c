// rt_wave_common.h — pieces shared by the wave-mapped kernels (rt_kernel_wave.hip, rt_kernel_mesh.hip):
// wave helpers, and the exact interval bounds of the reference's ray-triangle / ray-sphere tests.
// Numerics contract: rt_math.h.  See DESIGN.md 4.1 for the derivation of the bounds.
#pragma once
#include "rt_trace.h"

namespace uobrt {
namespace {

constexpr int kRngPixels = 4;               // pixels whose sample streams are generated together
constexpr int kRngStride = 64 * 4 + 4;      // 32-bit words per pixel in the scratch (+4: bank spread)

__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float shfl(float v, int lane) { return __shfl(v, lane, 64); }
// A copy of the lane id the optimiser cannot see through.  Addresses and constants derived from it are
// recomputed where they are used (2-3 cheap VALU instructions) instead of being hoisted out of the task loop
// and kept in registers for the whole kernel — hoisted LDS addresses were what pushed the kernel past 96
// VGPRs and into scratch spills, each reload a ~500-cycle dependent VMEM access inside the hot loops.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ float norm1(f3 a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z); }
// Square root for the BOUNDS only (never for a value of the reference's arithmetic): v_sqrt_f32, 1 ulp, one
// instruction instead of the ~12 of the correctly rounded sqrtf.  Every bound that uses it carries a relative
// slack of at least 1e-6.
__device__ __forceinline__ float bsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// |a|_inf.  (Issuing v_max_f32 / v_max3_f32 through inline asm — to avoid the `v_max x,x` with which the compiler quiets
// possible signalling NaNs in front of fmaxf operands that come from loads, lane reads and DPP moves, 94 of the kernel's
// 150 v_max — was built and measured: the asm blocks cost more in scheduling and hazard padding than the canonicalising
// instructions do; 3.59 -> 3.98 ms.)
__device__ __forceinline__ float norm_inf(f3 a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)); }
__device__ __forceinline__ float max_abs3(float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); }

// A 64-bit value that is the same in every lane, read back through the scalar unit: a loop over a wave-uniform mask that
// sits inside a divergent `if` otherwise keeps the mask (and runs its bookkeeping) in vector registers
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// Wave-wide reductions.  The result is the same in every lane; readfirstlane tells the compiler so
// (otherwise everything derived from it — the candidate masks, the loops over them — is treated as
// divergent and kept in VGPRs with exec-mask loops).
__device__ __forceinline__ float uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// DPP data movement: the value of another lane of the same row of 16 (row_shr / row_shl) or the last lane of the
// previous row(s) (row_bcast), inside the VALU — no LDS round trip.  Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
// Six dependent VALU steps (row_shr 1,2,4,8 leave each row's result in its lane 15; row_bcast:15 / :31 carry it on
// to lane 63) instead of six ds_bpermute round trips of ~100 cycles each: the reductions sit on the critical path
// of every task (level 1) and the kernel is latency-sensitive at 5 waves per SIMD.  Call with all 64 lanes active.
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp<0x111, 0xf>(v, v));
  v = fmaxf(v, dpp<0x112, 0xf>(v, v));
  v = fmaxf(v, dpp<0x114, 0xf>(v, v));
  v = fmaxf(v, dpp<0x118, 0xf>(v, v));
  v = fmaxf(v, dpp<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp<0x111, 0xf>(v, v));
  v = fminf(v, dpp<0x112, 0xf>(v, v));
  v = fminf(v, dpp<0x114, 0xf>(v, v));
  v = fminf(v, dpp<0x118, 0xf>(v, v));
  v = fminf(v, dpp<0x142, 0xa>(v, v));
  v = fminf(v, dpp<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The same for values that are known to be >= +0 (norms, lengths, pixel coordinates): non-negative floats order like
// their bit patterns, and an integer max needs no NaN quieting (fmaxf makes the compiler put a `v_max x,x` in front of
// every operand that comes out of a DPP move: twelve extra instructions per reduction).  A NaN (pattern above every
// finite value) wins the maximum, which every caller treats as "no bound".
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dppu(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_max_pos(float f) {
  unsigned v = __float_as_uint(f);
  v = max(v, dppu<0x111, 0xf>(v)); v = max(v, dppu<0x112, 0xf>(v)); v = max(v, dppu<0x114, 0xf>(v));
  v = max(v, dppu<0x118, 0xf>(v)); v = max(v, dppu<0x142, 0xa>(v)); v = max(v, dppu<0x143, 0xc>(v));
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float wave_min_pos(float f) {
  unsigned v = __float_as_uint(f);
  v = min(v, dppu<0x111, 0xf>(v)); v = min(v, dppu<0x112, 0xf>(v)); v = min(v, dppu<0x114, 0xf>(v));
  v = min(v, dppu<0x118, 0xf>(v)); v = min(v, dppu<0x142, 0xa>(v)); v = min(v, dppu<0x143, 0xc>(v));
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v, 63));
}

// Sum of the aa AA rays of a pixel in index order (final_color_total +=, kernels.cl:415-425), valid in the FIRST lane
// of each pixel's group of aa lanes [first, first + aa).  A ray without a contribution adds +0, which leaves the
// running sum unchanged bit for bit.  For the grids whose groups do not straddle a row of 16 lanes (aa = 1,2,4,8,16)
// lane first + r is `row_shl:r` away: one DPP add per ray; otherwise aa dependent ds_bpermute round trips.
template <int R>
__device__ __forceinline__ f3 add_shl(f3 acc, f3 c) {
  return mk(acc.x + dpp<0x100 + R, 0xf>(0.f, c.x), acc.y + dpp<0x100 + R, 0xf>(0.f, c.y), acc.z + dpp<0x100 + R, 0xf>(0.f, c.z));
}
__device__ __forceinline__ f3 aa_sum(f3 c, int aa, int first) {
  f3 acc = mk(0.f, 0.f, 0.f) + c;
  if (aa <= 16 && (aa & (aa - 1)) == 0) {
    if (aa > 1) acc = add_shl<1>(acc, c);
    if (aa > 2) { acc = add_shl<2>(acc, c); acc = add_shl<3>(acc, c); }
    if (aa > 4) { acc = add_shl<4>(acc, c); acc = add_shl<5>(acc, c); acc = add_shl<6>(acc, c); acc = add_shl<7>(acc, c); }
    if (aa > 8) {
      acc = add_shl<8>(acc, c); acc = add_shl<9>(acc, c); acc = add_shl<10>(acc, c); acc = add_shl<11>(acc, c);
      acc = add_shl<12>(acc, c); acc = add_shl<13>(acc, c); acc = add_shl<14>(acc, c); acc = add_shl<15>(acc, c);
    }
    return acc;
  }
  acc = mk(0.f, 0.f, 0.f);
  for (int r = 0; r < aa; ++r) acc = acc + mk(shfl(c.x, first + r), shfl(c.y, first + r), shfl(c.z, first + r));
  return acc;
}

// Per-lane registers of a "triangle lane" (lane i < ns holds shadow-casting triangle i)
struct TriLane {
  f3 v0, e1, e2, c;
  float c1, e1_1, e2_1;    // 1-norms |c|, |e1|, |e2| (interval bounds)
};

// ---- interval bounds ----------------------------------------------------------------------------------
// The bounds are certificates, not values of the reference's arithmetic: they may be evaluated with fused multiply-adds
// (one rounding instead of two — every slack below was sized for the unfused evaluation, so it still covers), which
// takes a third of their instructions away.  What must stay bit-identical to the reference's own evaluation is
// computed by the unfused helpers of rt_math.h (detc / cof): det(A0) and the cofactors p, q of point_bound, whose
// slack is relative to the computed values themselves.
#pragma clang fp contract(fast)
__device__ __forceinline__ f3 bcof(f3 m1, f3 m2) {
  return f3{m1.y * m2.z - m1.z * m2.y, m1.x * m2.z - m1.z * m2.x, m1.x * m2.y - m1.y * m2.x};
}
__device__ __forceinline__ float bdetc(f3 m0, f3 c) { return m0.x * c.x - m0.y * c.y + m0.z * c.z; }
__device__ __forceinline__ float bdot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Level 1 (lane = triangle): can ANY shadow sample of ANY lit surface point of the current task hit this
// triangle (`clear` = no), and do ALL of them hit it (`all_blocked`)?
//   s0, D0 : start / dir of a reference surface point;  es, ed : max |component| deviation of the other
//            points' start / dir from it;  hh : jitter half-width incl. rounding slack (max over points);
//   dlen_min/max : range of |dir| over the points.
struct Bound { bool clear, all_blocked; };
__device__ __forceinline__ Bound task_bound(const TriLane& T, f3 s0, f3 D0, float es, float ed, float hh,
                                            float dlen_min, float dlen_max) {
  const f3 b0 = s0 - T.v0;
  const float binf = norm_inf(b0);
  const float eb = 1.001f * es + 1e-6f * (binf + es);              // |b - b0| per component, any point
  const float nA0 = bdetc(b0, T.c);
  const f3 p0 = bcof(b0, T.e2), q0 = bcof(T.e1, b0);
  const float p1 = norm1(p0), q1 = norm1(q0);
  const float ep1 = 2.002f * eb * T.e2_1, eq1 = 2.002f * eb * T.e1_1;   // sum_k |p_k - p0_k|, |q_k - q0_k|
  const f3 md = -D0;
  const float A0 = bdetc(md, T.c), N1 = bdetc(md, p0), N2 = bdetc(md, q0);
  const float edd = 1.001f * ed + hh;                               // |d - D0| per component, any point, any sample
  const float E0 = eb * T.c1 * 1.0001f;
  const float EA = edd * T.c1 * 1.0001f;
  const float E1 = (dlen_max * ep1 + edd * (p1 + ep1)) * 1.0001f;
  const float E2 = (dlen_max * eq1 + edd * (q1 + eq1)) * 1.0001f;
  const float aD = fabsf(A0), hiD = aD + EA, loD = aD - EA;
  const bool robust = aD > EA + 1e-30f;
  const float sg = copysignf(1.0f, A0);
  const float tn = sg * nA0, un = sg * N1, vn = sg * N2;
  const float dmin = fmaxf(dlen_min - 1.7321f * hh, 0.0f), dmax = dlen_max + 1.7321f * hh;
  // the third edge as ONE linear function of the direction (see point_bound): det(-d, b - e1, e2 - e1)
  const f3 g = T.e2 - T.e1;
  const f3 w0 = bcof(b0 - T.e1, g);
  const float w1 = norm1(w0), ew1 = 2.002f * eb * norm1(g);
  const float W0 = sg * bdetc(md, w0);
  const float EW = (dlen_max * ew1 + edd * (w1 + ew1)) * 1.0001f;
  const float slackW = 4e-6f * (dlen_max + hh) * (((p1 + ep1) + (q1 + eq1) + T.c1) +
                                                  (norm1(b0) + 3.0f * eb + T.e1_1) * (T.e1_1 + T.e2_1));
  // A sample can only hit if det(A), det(A0), det(A1), det(A2) share one sign (t,u,v >= 0): cull when
  // neither the all-positive nor the all-negative combination is possible.  No condition on det(A): this
  // also settles rays that are nearly parallel to the triangle's plane, where det(A) changes sign.
  const bool can_pos = (A0 + EA > 0.0f) && (nA0 + E0 > -1e-18f) && (N1 + E1 > -1e-18f) && (N2 + E2 > -1e-18f);
  const bool can_neg = (A0 - EA < 0.0f) && (nA0 - E0 < 1e-18f) && (N1 - E1 < 1e-18f) && (N2 - E2 < 1e-18f);
  const bool cR = (fabsf(nA0) - E0) * dmin > hiD * (dlen_max * 1.000004f);   // |t d|^2 >= radius_sq everywhere
  const bool cW = fabsf(N1 + N2) - (E1 + E2) > hiD * 1.000004f;              // u+v > 1 wherever u,v >= 0
  const bool cE = robust && (W0 - EW > slackW);                              // u+v > 1 for every sample
  Bound r;
  r.clear = (!can_pos && !can_neg) || cR || cW || cE;
  r.all_blocked = robust && (tn - E0 > 1e-18f) && (un - E1 > 1e-18f) && (vn - E2 > 1e-18f) &&
                  (W0 + EW < -slackW) &&
                  ((fabsf(nA0) + E0) * dmax < loD * (dlen_min * 0.999996f));
  return r;
}

// The same question asked from the LIGHT's side (rt_kernel_mesh.hip: level 1 of a task, and rt_bin_shadow for
// the start points of a whole world cell).  The ray's start is tied to its direction — start = X + eps dir,
// X = light - dir (kernels.cl:323-324) — so
//   b = start - v0 = L - (1 - eps) dir + Delta,   L = light - v0,  |Delta| = a few ulps of the coordinates,
// and with d = dir + j (j the jitter), det(dir, dir, e) = 0:
//   det(A)  = -d.c                                  det(A0) = L.c - (1 - eps) dir.c + Delta.c
//   det(A1) = -dir.cof(L,e2) - j.cof(b,e2) + ...    det(A2) = -dir.cof(e1,L) - j.cof(e1,b) + ...
// i.e. linear in the direction alone, with coefficients fixed per frame, plus a jitter term.  task_bound
// must treat the start box and the direction box of an arbitrary point set as independent, which doubles
// the width; here the direction box (D0 +- ed) counts once, and the start box (s0 +- es) only scales the
// jitter term.   M : bound on the coordinates involved (|light|, |v0|, |dir|), for the rounding terms.
__device__ __forceinline__ Bound light_bundle_bound(const TriLane& T, f3 light, f3 s0, float es, f3 D0, float ed, float hh,
                                                    float dlen_min, float dlen_max, float M) {
  const f3 Lv = light - T.v0;
  const float Linf = norm_inf(Lv);
  const f3 pL = bcof(Lv, T.e2), qL = bcof(T.e1, Lv);
  const f3 b0 = s0 - T.v0;
  const float binf = norm_inf(b0);
  const float eb = 1.001f * es + 1e-6f * (binf + es);                // |b - b0| per component, any point
  const float pj = norm1(bcof(b0, T.e2)) + 2.002f * eb * T.e2_1;      // >= |bcof(b,e2)|_1, any point
  const float qj = norm1(bcof(T.e1, b0)) + 2.002f * eb * T.e1_1;
  const f3 md = -D0;
  const float A0 = bdetc(md, T.c), N1 = bdetc(md, pL), N2 = bdetc(md, qL);
  const float nA0 = bdetc(Lv, T.c) + 0.9999f * A0;                    // L.c - (1 - eps) D0.c
  const float ed1 = 1.001f * ed;                                     // |dir - D0| per component, any point
  const float Bmax = Linf + dlen_max;                                // |b| per component
  const float rnd = 8e-6f * dlen_max * (Bmax + M);                   // roundings of bcof(b,e), of the dots, Delta
  const float E0 = (ed1 + 4e-6f * (M + Bmax)) * T.c1 * 1.0001f;
  const float EA = (ed1 + hh) * T.c1 * 1.0001f;
  const float E1 = (ed1 * norm1(pL) + hh * pj + rnd * T.e2_1) * 1.0001f;
  const float E2 = (ed1 * norm1(qL) + hh * qj + rnd * T.e1_1) * 1.0001f;
  const float aD = fabsf(A0), hiD = aD + EA, loD = aD - EA;
  const bool robust = aD > EA + 1e-30f;
  const float sg = copysignf(1.0f, A0);
  const float tn = sg * nA0, un = sg * N1, vn = sg * N2;
  const float dmin = fmaxf(dlen_min - 1.7321f * hh, 0.0f), dmax = dlen_max + 1.7321f * hh;
  // the third edge as one linear function (see point_bound): det(A1)+det(A2)-det(A) = -dir.wL - j.wb,
  // wL = bcof(L - e1, e2 - e1), wb = bcof(b - e1, e2 - e1)
  const f3 g = T.e2 - T.e1;
  const float g1 = norm1(g);
  const f3 wL = bcof(Lv - T.e1, g);
  const float wj = norm1(bcof(b0 - T.e1, g)) + 2.002f * eb * g1;      // >= |wb|_1, any point
  const float W0 = sg * bdetc(md, wL);
  const float EW = (ed1 * norm1(wL) + hh * wj + rnd * g1) * 1.0001f;
  const float slackW = 4e-6f * (dlen_max + hh) * ((pj + qj + T.c1) +
                                                  (norm1(Lv) + norm1(b0) + 3.0f * eb + T.e1_1) * (T.e1_1 + T.e2_1));
  const bool can_pos = (A0 + EA > 0.0f) && (nA0 + E0 > -1e-18f) && (N1 + E1 > -1e-18f) && (N2 + E2 > -1e-18f);
  const bool can_neg = (A0 - EA < 0.0f) && (nA0 - E0 < 1e-18f) && (N1 - E1 < 1e-18f) && (N2 - E2 < 1e-18f);
  const bool cR = (fabsf(nA0) - E0) * dmin > hiD * (dlen_max * 1.000004f);   // |t d|^2 >= radius_sq everywhere
  const bool cW = fabsf(N1 + N2) - (E1 + E2) > hiD * 1.000004f;              // u+v > 1 wherever u,v >= 0
  const bool cE = robust && (W0 - EW > slackW);                              // u+v > 1 for every sample
  Bound r;
  r.clear = (!can_pos && !can_neg) || cR || cW || cE;
  r.all_blocked = robust && (tn - E0 > 1e-18f) && (un - E1 > 1e-18f) && (vn - E2 > 1e-18f) &&
                  (W0 + EW < -slackW) &&
                  ((fabsf(nA0) + E0) * dmax < loD * (dlen_min * 0.999996f));
  return r;
}

// Level 2 (lane = surface point): the same question for ONE point (this lane's) and a wave-uniform
// triangle.  hh >= h plus every rounding error of the per-sample evaluation; every sample's det(A) lies in
// D0 +- hh*|c|_1, det(A1) in N1 +- hh*|p|_1, det(A2) in N2 +- hh*|q|_1 (they are linear in the direction).
__device__ __forceinline__ Bound point_bound(f3 start, f3 dir, float hh, float dlen, float dminlen, float dk,
                                             f3 v0, f3 e1, f3 e2, f3 c) {
  const f3 b = start - v0;
  const f3 p = cof(b, e2), q = cof(e1, b);       // unfused: the reference's own p, q (their slack is relative to them)
  const float nA0 = detc(b, c);                   // unfused: the reference's det(A0), bit for bit
  const f3 md = -dir;
  const float D0 = bdetc(md, c), N1 = bdetc(md, p), N2 = bdetc(md, q);
  const float aD = fabsf(D0);
  const float c1 = norm1(c), p1 = norm1(p), q1 = norm1(q);
  const float Delta = hh * c1;
  const bool robust = aD > Delta + 1e-30f;           // every sample's det(A) has D0's sign and is normal
  const float sg = copysignf(1.0f, D0);
  const float tn = sg * nA0, un = sg * N1, vn = sg * N2;
  const float hp = hh * p1, hq = hh * q1;
  const float hiD = aD + Delta, loD = aD - Delta;
  // The third edge, exactly: det(A1) + det(A2) - det(A) = det(-d, b - e1, e2 - e1) is ONE linear function of
  // the sample direction, so its range over the jitter box is centre +- hh |w|_1.  (Bounding the three
  // determinants separately, as cW below must when det(A) may change sign, ignores that they move together:
  // a ray skimming along a face passes its top edge at u+v = 1.03 for every sample, yet the three separate
  // intervals overlap.)  slack: the reference rounds u, v and u+v (a few 2^-24 of |d| (|p|+|q|+|c|)), and w is not
  // formed the way the reference forms p, q and c (a few 2^-24 of |d| (|b|+|e1|) (|e1|+|e2|), whatever cancels).
  const f3 w = bcof(b - e1, e2 - e1);
  const float W0 = sg * bdetc(md, w), hw = hh * norm1(w);
  // second term: w is rounded like any product of (b - e1) and (e2 - e1) however small p, q and c come out
  const float e1n = norm1(e1);
  const float slackW = 4e-6f * (dlen + hh) * ((p1 + q1 + c1) + (norm1(b) + e1n) * (e1n + norm1(e2)));
  // sign consistency of det(A), det(A0), det(A1), det(A2) (see task_bound); det(A0) is exact here
  const bool can_pos = (D0 + Delta > 0.0f) && (nA0 > -1e-18f) && (N1 + hp > -1e-18f) && (N2 + hq > -1e-18f);
  const bool can_neg = (D0 - Delta < 0.0f) && (nA0 < 1e-18f) && (N1 - hp < 1e-18f) && (N2 - hq < 1e-18f);
  const bool cR = fabsf(nA0) * dminlen > hiD * dk;               // |t*d|^2 >= radius_sq for every sample
  const bool cW = fabsf(N1 + N2) - (hp + hq) > hiD * 1.000004f;  // u+v > 1 wherever u,v >= 0
  const bool cE = robust && (W0 - hw > slackW);                  // u+v > 1 for every sample
  Bound r;
  r.clear = (!can_pos && !can_neg) || cR || cW || cE;
  r.all_blocked = robust && (tn > 1e-18f) && (un - hp > 1e-18f) && (vn - hq > 1e-18f) &&
                  (W0 + hw < -slackW) &&
                  (fabsf(nA0) * (dlen + 1.7321f * hh) < loD * (dlen * 0.999996f));
  return r;
}

// Primary rays of one task (lane = triangle i < n): can ANY of the task's rays hit this triangle?
// The rays leave the camera through the sub-pixel rectangle [wc +- (hx,hy)] x {focal}; before
// normalisation their directions are du = R w, i.e. duc +- eu per component, and every determinant of
// the test is linear in the direction, so the sign/ratio conditions can be checked on du (they are
// invariant under the positive scale 1/|du|).  slack covers the roundings of R w, of the normalisation
// and of the per-ray determinant evaluation (each a few 2^-24 relative to |du|_max * |cofactors|_1).
__device__ __forceinline__ bool primary_clear(f3 duc, f3 eu, float dumax, f3 c, float nA0cam, f3 pc, f3 qc) {
  const f3 md = -duc;
  const float Ac = bdetc(md, c), N1 = bdetc(md, pc), N2 = bdetc(md, qc);
  const float sl = 4e-6f * dumax;
  const float EA = eu.x * fabsf(c.x) + eu.y * fabsf(c.y) + eu.z * fabsf(c.z) + sl * norm1(c);
  const float E1 = eu.x * fabsf(pc.x) + eu.y * fabsf(pc.y) + eu.z * fabsf(pc.z) + sl * norm1(pc);
  const float E2 = eu.x * fabsf(qc.x) + eu.y * fabsf(qc.y) + eu.z * fabsf(qc.z) + sl * norm1(qc);
  const bool can_pos = (Ac + EA > 0.0f) && (nA0cam > -1e-18f) && (N1 + E1 > -1e-30f) && (N2 + E2 > -1e-30f);
  const bool can_neg = (Ac - EA < 0.0f) && (nA0cam < 1e-18f) && (N1 - E1 < 1e-30f) && (N2 - E2 < 1e-30f);
  const bool cW = fabsf(N1 + N2) - (E1 + E2) > (fabsf(Ac) + EA) * 1.000004f;
  // the third edge as one linear function of the direction (see point_bound); w = pc + qc - c is formed here
  // from the staged cofactors, the rounding of that sum and of the reference's u, v, u+v goes into the slack
  const f3 w = (pc + qc) - c;
  const float cn = norm1(c) + norm1(pc) + norm1(qc);
  const float W = copysignf(1.0f, Ac) * bdetc(md, w);
  const float EW = eu.x * fabsf(w.x) + eu.y * fabsf(w.y) + eu.z * fabsf(w.z) + 2.0f * sl * cn;
  const bool cE = (fabsf(Ac) > EA) && (W - EW > 0.0f);           // u+v > 1 for every ray of the bundle
  return (!can_pos && !can_neg) || cW || cE;
}

// Can any ray of a bundle — origins o +- eo per component, directions dir + e with |e|_2 <= jm — touch a sphere?
// Conservative: the line misses sphere (c,R) when |L x d| > R |d|; bound both sides over the bundle
//   |L x d| >= |L0 x dir| - |L0| jm - sqrt(3) eo |d|,   |d| <= |dir| + jm
// and leave 0.2 % for the rounding of the reference's discriminant b*b - 4*a*c (kernels.cl:285, :214).  That rounding
// is at most ~50*2^-24 * |d|^2 (|L|^2 + R^2), while the margin is worth 8*0.002 * R^2 |d|^2: rigorous for |L|/R < 73,
// applied for |L|/R < 40; farther (or degenerate) spheres are simply always tested.  The condition is
// homogeneous in d, so it holds for the normalised directions of primary rays as well.
//   casters_only: skip glass spheres (they cast no shadow, :279); primary and bounce rays see every sphere.
__device__ __forceinline__ bool sphere_bundle_maybe(const FrameParams& P, f3 o, float eo, f3 dir, float dlen, float jm, bool casters_only) {
  bool maybe = false;
  const float eo2 = 1.7321f * 1.001f * eo;
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = RT_SPH(P, i);
    if (casters_only && sp.col[3] == -1.0f) continue;
    const f3 Lv = o - mk(sp.cx, sp.cy, sp.cz);
    const f3 cr = mk(Lv.y * dir.z - Lv.z * dir.y, Lv.z * dir.x - Lv.x * dir.z, Lv.x * dir.y - Lv.y * dir.x);
    const float crn = bsqrt(bdot3(cr, cr)), Ln = bsqrt(bdot3(Lv, Lv)), R = bsqrt(fmaxf(sp.r2, 0.0f));
    const bool miss = (crn - Ln * jm - eo2 * (dlen + jm) > R * (dlen + jm) * 1.002f) && (Ln + eo2 < 40.0f * R) && (sp.r2 > 0.0f);
    maybe = maybe || !miss;
  }
  return maybe;
}
// Shadow rays of ONE surface point against the shadow-casting spheres (kernels.cl:278-307): directions
// dir + e, e in the box [-hh,hh]^3 (hh includes the rounding slack of the sample direction).
//   returns maybe : some sample may touch a sphere (else the spheres need no evaluation for this point)
//   all_blocked   : every sample provably hits one sphere before the light
// With cr = L x dir and a = cr/|cr|:  |L x (dir+e)| >= (cr + L x e).a = |cr| + e.(a x L) >= |cr| - hh |a x L|_1
// (the box's support function in the one direction that matters; a ball of radius sqrt(3) hh is up to 1.7x
// wider).  Margins: 0.2 % / 1 % on the radius against the rounding of the reference's discriminant
// b*b - 4*a*c (at most ~50*2^-24 |d|^2 (|L|^2 + R^2)): rigorous for |L|/R < 73, applied for |L|/R < 40;
// farther or degenerate spheres are always evaluated.
struct SphereBound { bool maybe, all_blocked; };
__device__ __forceinline__ SphereBound spheres_point(const FrameParams& P, f3 start, f3 dir, float dlen, float hh) {
  SphereBound r;
  r.maybe = false; r.all_blocked = false;
  const float jm = 1.7321f * hh;
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = RT_SPH(P, i);
    if (sp.col[3] == -1.0f) continue;                               // glass casts no shadow, :279
    const f3 Lv = start - mk(sp.cx, sp.cy, sp.cz);
    const f3 cr = mk(Lv.y * dir.z - Lv.z * dir.y, Lv.z * dir.x - Lv.x * dir.z, Lv.x * dir.y - Lv.y * dir.x);
    const f3 cl = mk(cr.y * Lv.z - cr.z * Lv.y, cr.z * Lv.x - cr.x * Lv.z, cr.x * Lv.y - cr.y * Lv.x);     // cr x L
    const float crn = bsqrt(bdot3(cr, cr)), Ln2 = bdot3(Lv, Lv), Ln = bsqrt(Ln2), R = bsqrt(fmaxf(sp.r2, 0.0f));
    const bool near = (Ln < 40.0f * R) && (sp.r2 > 0.0f);
    const bool miss = near && (crn * crn - hh * norm1(cl) > R * (dlen + jm) * 1.002f * crn);
    r.maybe = r.maybe || !miss;
    // all samples: start outside the sphere, sphere ahead of the start and wholly nearer than the light
    // (then both roots are positive and the near one lies within |L| of the start), line through the sphere
    const bool hit = near && (Ln2 > 1.001f * sp.r2) && (bdot3(dir, Lv) + hh * norm1(Lv) < 0.0f) && (Ln < 0.999f * dlen) &&
                     (crn + Ln * jm < 0.99f * R * (dlen - jm)) && (dlen > jm);
    r.all_blocked = r.all_blocked || hit;
  }
  return r;
}

#pragma clang fp contract(off)

}  // namespace
}  // namespace uobrt
