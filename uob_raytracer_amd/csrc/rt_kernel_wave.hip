// rt_kernel_wave.hip — the headline kernel: 64 shadow samples == one 64-lane wavefront.
//
// The reference spends ~98 % of its ray-primitive tests in direct_light's shadow loop
// (Source/kernels.cl:313-340 -> in_shadow :243-311): for ONE surface point, S jittered rays towards the
// area light are tested against every triangle.  On gfx950 a wavefront is 64 lanes, so for S = 64 the
// sample-level work maps   lane = shadow sample   and surface points are processed one after another:
// all 64 lanes test the same triangle against the same surface point, so the two-stage test of in_shadow
// (t first, u/v only if t passes, :266) is a WAVE-UNIFORM branch, lane predicates are 64-bit wave masks
// (one v_cmp each, logic on the scalar unit) and the any-hit early-out is "blocked mask == all ones".
//
// Persistent waves pull JOBS from a multi-headed queue (see the job loop): a job is up to 64 consecutive pixels
// of one image row (its framebuffer store is one coalesced access), walked in tasks of 64 primary rays
// (64/aa pixels x aa AA rays):
//   phase 1  primary rays, lane = (pixel, AA sample);  phase 2  mirror/glass bounces (kernels.cl:342-365)
//   phase 3  shadows of the 64 surface points (below);  phase 4  shading, AA sum in the reference's order.
//
// Phase 3 with CULL (the shipped path) is a three-level exact hierarchy.  The determinants of the
// reference's test are linear in the ray direction, so over the jitter box of the area light
// (dir + [-h,h]^3) each is  centre value +- h*|cofactors|_1 ; with explicit slack for every FP32 rounding
// of the per-sample evaluation these intervals decide, for ALL 64 samples at once, whether the
// reference's comparisons (t>=0, |t d|^2<r^2, u>=0, v>=0, u+v<=1) can come out true / must come out true:
//   level 1 (lane = triangle, once per task): bounds over all lit surface points of the task -> the few
//            triangles K that may matter; or "one triangle blocks everything" -> task done;
//   level 2 (lane = surface point, loop over K): the same bounds per point -> per point: fully lit,
//            fully blocked, or the triangles whose samples must really be tested;
//   level 3 (lane = sample): the reference's test, operation for operation, for those (point, triangle)
//            pairs only, with the per-pixel xorshift streams (seeded by the GLOBAL pixel id, :319)
//            generated on demand into per-wave LDS.
// The result is bit-identical to testing every triangle for every sample (tests/test_gpu_cull.py); the
// brute-force form of phase 3 (CULL = false, RT_FLAG_NO_CULL) is kept for that comparison: there lanes
// 0..n-1 act as triangle lanes that compute the sample-independent terms once per surface point and hand
// them to the sample lanes through LDS records read at a uniform address (an LDS broadcast).
//
// Arithmetic is the reference's, operation for operation (rt_math.h): results are bit-identical to the
// generic kernel and to the CPU oracle.
#define RT_SPHERES_IN_LDS
#define RT_RNG_JUMP_IN_LDS
#include "rt_wave_common.h"

#ifndef RT_OPT_SPHJOB
#define RT_OPT_SPHJOB 1
#endif
#ifndef RT_OPT_LIGHTSIDE
#define RT_OPT_LIGHTSIDE 1
#endif
#ifndef RT_OPT_JOBK
#define RT_OPT_JOBK 1       // level 1 of a wider point set, reused by the job's next tasks
#endif
#ifndef RT_OPT_TASKSPH
#define RT_OPT_TASKSPH 1
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 4          // jobs per hand-out while the queue is long (2: 3.58 ms, 4: 3.50, 8: 3.64 on the headline frame)
#endif

#include <type_traits>

namespace uobrt {

namespace {

// Per-wave LDS.
//   r0..r2 (brute-force path): sample-independent terms of (surface point, triangle i), written by
//          triangle lane i, read by all sample lanes at the same address (LDS broadcast into VGPRs;
//          broadcasting through SGPRs with v_readlane_b32 costs 4.3 issue cycles per value and makes every
//          consuming VALU instruction half rate: profiles/r01_valu_issue_cost_8waves.txt)
//   h0,h1: the 64 surface points of the current task, written lane-parallel, read as broadcasts
//   rng  : xorshift streams of kRngPixels pixels
struct WaveLds {
  float4* r0;   // c.x c.y c.z | det(A0) = det(b,e1,e2)
  float4* r1;   // p.x p.y p.z | q.x        p = cof(b,e2), q = cof(e1,b)
  float2* r2;   // q.y q.z
  float4* h0;   // start.xyz | radius_sq
  float4* h1;   // dir.xyz   | hh (jitter half-width incl. rounding slack)
  uint32_t* rng;
};
// the r0..r2 records exist only in the brute-force build
__host__ __device__ constexpr int wave_lds_bytes(bool cull) {
  return (cull ? 0 : 64 * (16 + 16 + 8)) + 64 * 32 + kRngPixels * kRngStride * 4;
}

__device__ __forceinline__ WaveLds wave_lds(char* base, bool cull) {
  WaveLds L;
  L.r0 = reinterpret_cast<float4*>(base);
  L.r1 = reinterpret_cast<float4*>(base + 64 * 16);
  L.r2 = reinterpret_cast<float2*>(base + 64 * 32);
  if (!cull) base += 64 * 40;
  L.h0 = reinterpret_cast<float4*>(base);
  L.h1 = reinterpret_cast<float4*>(base + 64 * 16);
  L.rng = reinterpret_cast<uint32_t*>(base + 64 * 32);
  return L;
}

// ---- the reference's sample test, lane = sample ----------------------------------------------------
// One triangle against the 64 jittered rays of one surface point: kernels.cl:249-275.  c, nA0 = det(A0),
// p = cof(b,e2), q = cof(e1,b) are wave-uniform values held in VGPRs.  Returns the lanes that hit.
// `todo` = sample lanes that exist (lane < S) and are not blocked yet.
template <bool COUNT>
__device__ __forceinline__ unsigned long long sample_test(f3 d, f3 nd, float radius_sq, f3 c, float nA0, f3 p, f3 q,
                                                          unsigned long long todo, Work& wk) {
  const float detA = detc(nd, c);
  float rr = rcp_newton(detA, 1);          // == 1.0f/detA, or NaN when detA is 0/denormal/inf (rt_math.h)
  float t = nA0 * rr;
  f3 dv = t * d;
  float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
  // first stage, :266.  The negated compares are also true for NaN, so a lane whose reciprocal needs the
  // division fallback reaches the second stage, where it is recomputed exactly.
  unsigned long long pass = ballot(!(t < 0.0f)) & ballot(!(dist >= radius_sq));
  if (COUNT) wk.v[1] += 1;
  if ((pass & todo) == 0ull) return 0ull;
  if (COUNT) wk.v[2] += 1;
  if (ballot(rr != rr) != 0ull) {                                    // rare: reciprocal outside v_rcp's range
    rr = 1.0f / detA;
    t = nA0 * rr;
    dv = t * d;
    dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
    pass = ballot(t >= 0) & ballot(dist < radius_sq);
  }
  const float u = detc(nd, p) * rr;
  const float v = detc(nd, q) * rr;
  return pass & ballot(u >= 0) & ballot(v >= 0) & ballot((u + v) <= 1);   // :272
}

// ---- brute force (CULL = false): every triangle for every surface point ---------------------------
template <bool COUNT>
__device__ __forceinline__ int wave_unshadowed_all(const FrameParams& P, const TriLane& T, const WaveLds& L, int lane,
                                                   int ns, int j, f3 jit, unsigned long long active, Work& wk) {
  const float4 h0 = L.h0[j], h1 = L.h1[j];                 // LDS broadcasts
  const f3 start = mk(h0.x, h0.y, h0.z), dir = mk(h1.x, h1.y, h1.z);
  const float radius_sq = h0.w;
  const f3 d = dir + jit;       // shadow_ray.direction + crush(rand_vec, light_spread), :333
  const f3 nd = -d;
  {                             // once per surface point, lane i = triangle i
    const f3 b = start - T.v0;
    const f3 p = cof(b, T.e2), q = cof(T.e1, b);
    reinterpret_cast<float*>(&L.r0[lane])[3] = detc(b, T.c);
    L.r1[lane] = make_float4(p.x, p.y, p.z, q.x);
    L.r2[lane] = make_float2(q.y, q.z);
    if (COUNT) wk.v[0] += 1;
  }
  __builtin_amdgcn_wave_barrier();
  unsigned long long shadowed = 0ull;
  float4 r0 = L.r0[0];
  for (int i = 0; i < ns; ++i) {
    const float4 nxt = L.r0[i + 1 < 64 ? i + 1 : 63];     // prefetch the next record
    const float4 r1 = L.r1[i];
    const float2 r2 = L.r2[i];
    shadowed |= active & sample_test<COUNT>(d, nd, radius_sq, mk(r0.x, r0.y, r0.z), r0.w, mk(r1.x, r1.y, r1.z),
                                            mk(r1.w, r2.x, r2.y), active & ~shadowed, wk);
    if (shadowed == active) break;                         // every sample blocked: any-hit early-out
    r0 = nxt;
  }
  __builtin_amdgcn_wave_barrier();
  bool sh = (shadowed >> lane) & 1ull;
  if (P.nsph > 0 && shadowed != active) {
    Work unused;
    if (COUNT) wk.v[3] += 1;
    if (!sh) sh = shadow_spheres<false>(P, start, d, radius_sq, unused);
  }
  return __popcll(active & ballot(!sh));
}

// Level 3 (lane = sample): TWO surface points ja, jb of one pixel (same jitter) per pass against the
// triangles of K whose bit is set in `need` (the union of the two points' sets: testing a triangle a point
// does not need cannot produce a hit for it, that is what the bound proved).  Two points per pass give the
// dependent chain  LDS record -> determinants -> reciprocal -> compares  two independent instances to
// interleave: with one point per pass this level ran at ~20 % VALU utilisation, bound by latency (7.8 ms
// of the headline frame; 5.5 ms with two; four or eight points cost more in registers than they gain).
// SC = the shadow-casting triangles' records (v0,e1,e2,c) in K's bit order.  Returns both counts.
struct ShadowCasters { const float4 *v0, *e1, *e2, *c; };
struct Count2 { int a, b; };
template <bool COUNT>
__device__ __forceinline__ Count2 wave_unshadowed_pair(const FrameParams& P, const ShadowCasters& SC, const WaveLds& L,
                                                       int lane, int ja, int jb, unsigned long long K,
                                                       unsigned long long need, bool sph_a, bool sph_b, f3 jit,
                                                       unsigned long long active, Work& wk) {
  const float4 ha0 = L.h0[ja], ha1 = L.h1[ja], hb0 = L.h0[jb], hb1 = L.h1[jb];     // LDS broadcasts
  const f3 sa = mk(ha0.x, ha0.y, ha0.z), sb = mk(hb0.x, hb0.y, hb0.z);
  const float ra = ha0.w, rb = hb0.w;
  const f3 da = mk(ha1.x, ha1.y, ha1.z) + jit, db = mk(hb1.x, hb1.y, hb1.z) + jit;   // dir + crush(...), :333
  const f3 nda = -da, ndb = -db;
  if (COUNT) wk.v[0] += (ja == jb) ? 1 : 2;
  unsigned long long sha = 0ull, shb = 0ull;
  int pos = 0;
  for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull, ++pos) {
    if (((need >> pos) & 1ull) == 0ull) continue;
    const int k = __builtin_ctzll(kk);
    const f3 v0 = xyz(SC.v0[k]), e1 = xyz(SC.e1[k]), e2 = xyz(SC.e2[k]), c = xyz(SC.c[k]);
    // first stage for both points, :251-266 (negated compares: also true for NaN, see rcp_newton)
    const f3 ba = sa - v0, bb = sb - v0;
    const float nA0a = detc(ba, c), nA0b = detc(bb, c);
    const float detAa = detc(nda, c), detAb = detc(ndb, c);
    float rra = rcp_newton(detAa, 1), rrb = rcp_newton(detAb, 1);
    float ta = nA0a * rra, tb = nA0b * rrb;
    f3 dva = ta * da, dvb = tb * db;
    float dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
    float distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
    unsigned long long passa = ballot(!(ta < 0.0f)) & ballot(!(dista >= ra));
    unsigned long long passb = ballot(!(tb < 0.0f)) & ballot(!(distb >= rb));
    if (COUNT) wk.v[1] += (ja == jb) ? 1 : 2;
    if (((passa & active & ~sha) | (passb & active & ~shb)) == 0ull) continue;
    if (COUNT) wk.v[2] += 1;
    if ((ballot(rra != rra) | ballot(rrb != rrb)) != 0ull) {         // rare: reciprocal outside v_rcp's range
      rra = 1.0f / detAa; rrb = 1.0f / detAb;
      ta = nA0a * rra; tb = nA0b * rrb;
      dva = ta * da; dvb = tb * db;
      dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
      distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
      passa = ballot(ta >= 0) & ballot(dista < ra);
      passb = ballot(tb >= 0) & ballot(distb < rb);
    }
    // second stage, :268-272
    const float ua = detc(nda, cof(ba, e2)) * rra, va = detc(nda, cof(e1, ba)) * rra;
    const float ub = detc(ndb, cof(bb, e2)) * rrb, vb = detc(ndb, cof(e1, bb)) * rrb;
    sha |= active & passa & ballot(ua >= 0) & ballot(va >= 0) & ballot((ua + va) <= 1);
    shb |= active & passb & ballot(ub >= 0) & ballot(vb >= 0) & ballot((ub + vb) <= 1);
    if (sha == active && shb == active) break;             // every sample of both points blocked
  }
  bool sh_a = (sha >> lane) & 1ull, sh_b = (shb >> lane) & 1ull;
  if ((sph_a && sha != active) || (sph_b && shb != active)) {
    Work unused;
    if (COUNT) wk.v[3] += 1;
    if (sph_a && !sh_a) sh_a = shadow_spheres<false>(P, sa, da, ra, unused);
    if (sph_b && !sh_b) sh_b = shadow_spheres<false>(P, sb, db, rb, unused);
  }
  Count2 r;
  r.a = __popcll(active & ballot(!sh_a));
  r.b = __popcll(active & ballot(!sh_b));
  return r;
}

// Level 3 for FEW samples (NS <= 32): with lane = sample a pass of wave_unshadowed_pair keeps NS of 64 lanes busy.  Here a lane
// is a (surface point, sample): SP = 16 or 32 sample slots per point (the slots >= NS idle), 64 / SP points per instance, and
// two independent instances a / b per pass as before (the dependent chain LDS record -> determinants -> reciprocal -> compares
// needs a second one to interleave with) — 8 points per pass at up to 16 samples, 4 at up to 32, instead of 2.  An instance's
// points belong to ONE pixel (they share its jitter stream `rng`: word 4 s + component of sample s); pm = their lanes, bit i =
// lane base + i.  Every lane tests the casters any point of its pass needs (testing a triangle a point does not need cannot
// produce a hit for it).  `unshadowed` of the points' own lanes receives the counts.
template <int SP>
__device__ __forceinline__ void wave_unshadowed_packed(const FrameParams& P, const ShadowCasters& SC, const WaveLds& L, int lane,
                                                       unsigned long long pmA, int baseA, const uint32_t* rngA,
                                                       unsigned long long pmB, int baseB, const uint32_t* rngB,
                                                       unsigned long long K, unsigned long long need_l, unsigned long long sphmask,
                                                       int NS, int& unshadowed) {
  constexpr int PK = 64 / SP;
  const int q = lane / SP, sidx = lane - q * SP;
  int ja[PK], jb[PK];
  {
    unsigned long long m = pmA;
#pragma unroll
    for (int i = 0; i < PK; ++i) { ja[i] = m != 0ull ? baseA + __builtin_ctzll(m) : -1; m &= m - 1ull; }
    m = pmB;
#pragma unroll
    for (int i = 0; i < PK; ++i) { jb[i] = m != 0ull ? baseB + __builtin_ctzll(m) : -1; m &= m - 1ull; }
  }
  int myA = ja[0], myB = jb[0];
#pragma unroll
  for (int i = 1; i < PK; ++i) { myA = q == i ? ja[i] : myA; myB = q == i ? jb[i] : myB; }
  const bool actA = myA >= 0 && sidx < NS, actB = myB >= 0 && sidx < NS;
  unsigned long long need = 0ull;
#pragma unroll
  for (int i = 0; i < PK; ++i) {
    if (ja[i] >= 0) need |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(need_l >> 32), ja[i]) << 32) | (unsigned)__builtin_amdgcn_readlane((int)need_l, ja[i]);
    if (jb[i] >= 0) need |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(need_l >> 32), jb[i]) << 32) | (unsigned)__builtin_amdgcn_readlane((int)need_l, jb[i]);
  }
  const float4 ha0 = L.h0[actA ? myA : 0], ha1 = L.h1[actA ? myA : 0], hb0 = L.h0[actB ? myB : 0], hb1 = L.h1[actB ? myB : 0];
  const f3 sa = mk(ha0.x, ha0.y, ha0.z), sb = mk(hb0.x, hb0.y, hb0.z);
  const float ra = ha0.w, rb = hb0.w;
  const int sw = (sidx < NS ? sidx : 0) * 4;
  const f3 jitA = mk(crush1(rngA[sw], P.spread), crush1(rngA[sw + 1], P.spread), crush1(rngA[sw + 2], P.spread));
  const f3 jitB = mk(crush1(rngB[sw], P.spread), crush1(rngB[sw + 1], P.spread), crush1(rngB[sw + 2], P.spread));
  const f3 da = mk(ha1.x, ha1.y, ha1.z) + jitA, db = mk(hb1.x, hb1.y, hb1.z) + jitB;   // dir + crush(...), :333
  const f3 nda = -da, ndb = -db;
  bool blkA = false, blkB = false;
  int pos = 0;
  for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull, ++pos) {
    if (((need >> pos) & 1ull) == 0ull) continue;
    const int k = __builtin_ctzll(kk);
    const f3 v0 = xyz(SC.v0[k]), e1 = xyz(SC.e1[k]), e2 = xyz(SC.e2[k]), c = xyz(SC.c[k]);
    // first stage, :251-266 (negated compares: also true for NaN, see rcp_newton)
    const f3 ba = sa - v0, bb = sb - v0;
    const float nA0a = detc(ba, c), nA0b = detc(bb, c);
    const float detAa = detc(nda, c), detAb = detc(ndb, c);
    float rra = rcp_newton(detAa, 1), rrb = rcp_newton(detAb, 1);
    float ta = nA0a * rra, tb = nA0b * rrb;
    f3 dva = ta * da, dvb = tb * db;
    float dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
    float distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
    bool passa = !(ta < 0.0f) && !(dista >= ra), passb = !(tb < 0.0f) && !(distb >= rb);
    if (ballot((actA && !blkA && passa) || (actB && !blkB && passb)) == 0ull) continue;
    if (ballot((actA && rra != rra) || (actB && rrb != rrb)) != 0ull) {        // rare: reciprocal outside v_rcp's range
      rra = 1.0f / detAa; rrb = 1.0f / detAb;
      ta = nA0a * rra; tb = nA0b * rrb;
      dva = ta * da; dvb = tb * db;
      dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
      distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
      passa = ta >= 0 && dista < ra; passb = tb >= 0 && distb < rb;
    }
    // second stage, :268-272
    const float ua = detc(nda, cof(ba, e2)) * rra, va = detc(nda, cof(e1, ba)) * rra;
    const float ub = detc(ndb, cof(bb, e2)) * rrb, vb = detc(ndb, cof(e1, bb)) * rrb;
    blkA = blkA || (passa && ua >= 0 && va >= 0 && (ua + va) <= 1);
    blkB = blkB || (passb && ub >= 0 && vb >= 0 && (ub + vb) <= 1);
    if (ballot((actA && !blkA) || (actB && !blkB)) == 0ull) break;          // every sample of every point blocked
  }
  const bool sphA = actA && ((sphmask >> (myA >= 0 ? myA : 0)) & 1ull) != 0ull, sphB = actB && ((sphmask >> (myB >= 0 ? myB : 0)) & 1ull) != 0ull;
  if (ballot((sphA && !blkA) || (sphB && !blkB)) != 0ull) {
    Work unused;
    if (sphA && !blkA) blkA = shadow_spheres<false>(P, sa, da, ra, unused);
    if (sphB && !blkB) blkB = shadow_spheres<false>(P, sb, db, rb, unused);
  }
  const unsigned long long mA = ballot(actA && !blkA), mB = ballot(actB && !blkB);      // samples that reach the light
#pragma unroll
  for (int i = 0; i < PK; ++i) {
    if (jb[i] >= 0 && lane == jb[i]) unshadowed = __popcll((mB >> (i * SP)) & ((1ull << SP) - 1ull));
    if (ja[i] >= 0 && lane == ja[i]) unshadowed = __popcll((mA >> (i * SP)) & ((1ull << SP) - 1ull));
  }
}


}  // namespace

// Persistent waves: the grid is what fits the chip at once (CUs x RT_MIN_WAVES workgroups of 4 waves); each
// wave pulls 64-pixel segments (jobs) from an atomic counter until none is left, so a wave slot is never idle
// while work remains.  (With one workgroup per 4 segments the cost spread between fully lit and penumbra
// segments left on average 2 of 4-5 possible waves per SIMD resident: workgroup resources are only released
// when the slowest wave of the workgroup ends.)  The exit condition is reached by every wave: the counter
// only grows.
// (one wave per workgroup measured 20 % slower than four: 22.1 vs 18.3 ms on the headline frame)
#ifndef RT_WAVES_PER_BLOCK
#define RT_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = RT_WAVES_PER_BLOCK;
// PROF builds (diagnostic only, never timed): per-phase s_memtime shares, summed over waves, in counters[0..7]
#define RT_STAMP(slot)                                                              \
  if (PROF) {                                                                       \
    unsigned long long now_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                              \
    prof[slot] += now_ - tlast;                                                     \
    tlast = now_;                                                                   \
  }
// STRIDE > 0: the LDS record arrays have a fixed stride of STRIDE triangles (n <= STRIDE) and the whole LDS
// layout is static, so every array base is an immediate of the ds_read instead of a VGPR; 0 = packed by n.
template <int STRIDE, bool CULL>
__host__ __device__ constexpr int wave_fixed_lds_float4() {
  return kLdsRecords * STRIDE + STRIDE / 4 + 4 * STRIDE + kWavesPerBlock * wave_lds_bytes(CULL) / 16;
}

// MULTI: more than 64 shadow samples per surface point, worked off in passes of 64 sample lanes
// AA_X, AA_Y, SS > 0: the AA grid and the sample count are compile-time constants ("specialise on (AA,
// samples)"): the lane <-> (pixel, AA sample) arithmetic becomes shifts, the 64 adds of direct_light's sum and the AA sum
// straight-line code, and a dozen wave-uniform conditions (and the scalar registers they were spilled from) disappear.
// Same operations on the same values: the frame is bit-identical to the generic instantiation's (tests/test_gpu_cull.py).
// BIGAA: 65..256 AA samples per pixel (the reference's grid is a pair of constants, kernels.cl:12-14): a pixel's samples are
// worked off in chunks of 64 — a task is one chunk of ONE pixel, a job's tasks run pixel by pixel, chunk by chunk, and the
// pixel's running sum (final_color_total +=, :415-425, in sample order) is carried from chunk to chunk.
template <bool CULL, bool COUNT, bool PROF = false, int STRIDE = 0, bool MULTI = false, int AA_X = 0, int AA_Y = 0, int SS = 0, bool BIGAA = false>
// 5 waves per SIMD (<= 96 VGPRs; what spills is written once per wave, outside the loops): the kernel is bound by
// instruction issue and needs the waves — 5 per SIMD measured 3.76 ms against 4.10 ms at 4 (128 VGPRs)
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 5
#endif
__global__ __launch_bounds__(64 * kWavesPerBlock, RT_MIN_WAVES) void rt_draw_wave(const FrameParams P) {
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
  extern __shared__ float4 lds_dyn[];
  __shared__ float4 lds_fix[STRIDE ? wave_fixed_lds_float4<STRIDE ? STRIDE : 4, CULL>() : 1];
  float4* const lds = STRIDE ? lds_fix : lds_dyn;
  const int tid = threadIdx.x;
  // (The compiler cannot know that tid >> 6 is the same in all 64 lanes: the queue head, the job id and everything derived
  // from them count as per-lane values and live in vector registers.  Declaring the wave index uniform with readfirstlane
  // moves them to the scalar file, which is the fuller one: 18 more scalar spills, 3.23 -> 3.25 ms.  Left as it is.)
  const int wave = tid >> 6, lane = tid & 63;
  const int n = P.n, ns = P.n_shadow;
  const int st = STRIDE ? STRIDE : n, sst = STRIDE ? STRIDE : ns;     // record-array strides

  // ---- stage the triangle list once per workgroup ---------------------------------------------------
  stage_triangles(P, lds, tid, 64 * kWavesPerBlock, st);
  stage_spheres(P, tid);
  stage_rng_jump(tid);
  int* sidx = reinterpret_cast<int*>(lds + kLdsRecords * st);         // shadow-casting triangles, in order
  if (wave == 0) {                                                    // n <= 64 on this path (supports())
    const bool casts = (lane < n) && (P.colors[lane < n ? lane : 0].w != -1.0f);   // glass casts no shadow, :247
    const unsigned long long m = ballot(casts);
    if (casts) sidx[__popcll(m & ((1ull << lane) - 1ull))] = lane;
  }
  RT_STAMP(7)                               // 7: staging work before the workgroup barrier
  __syncthreads();
  // the shadow-casting triangles' (v0,e1,e2,c) once more, in caster order: levels 2 and 3 index them by
  // the bit position of the candidate mask, with no index indirection in their dependent chains
  float4* scbase = reinterpret_cast<float4*>(sidx + ((st + 3) & ~3));
  for (int kq = tid; kq < ns; kq += 64 * kWavesPerBlock) {
    const int ti = sidx[kq];
    scbase[kq] = lds[ti]; scbase[sst + kq] = lds[st + ti]; scbase[2 * sst + kq] = lds[2 * st + ti]; scbase[3 * sst + kq] = lds[3 * st + ti];
  }
  __syncthreads();
  const ShadowCasters SC{scbase, scbase + sst, scbase + 2 * sst, scbase + 3 * sst};
  const WaveLds L = wave_lds(reinterpret_cast<char*>(scbase + 4 * sst) + wave * wave_lds_bytes(CULL), CULL);

  const LdsScene S = lds_scene(lds, n, st);
  const int aa_x = AA_X ? AA_X : P.aa_x, aa_y = AA_Y ? AA_Y : P.aa_y;
  const float sy = (AA_X && AA_Y) ? (float)AA_X / (float)AA_Y : P.sy;
  const int aa_full = aa_x * aa_y;                                    // AA samples per pixel: 1..64, BIGAA: 65..256 (supports())
  const int chunks = BIGAA ? (aa_full + 63) >> 6 : 1;                 // tasks per pixel
  const int aa = BIGAA ? 64 : aa_full;                                // lanes of one pixel in a task (BIGAA: one chunk of its samples)
  const int PT = BIGAA ? 1 : 64 / aa;                                 // pixels per task (lanes >= PT * aa idle)
  f3 aa_run = mk(0.f, 0.f, 0.f);                                      // BIGAA: the current pixel's running AA sum
  const int GP = PT < kRngPixels ? PT : kRngPixels;                   // pixels per RNG group
  const int gp_magic = (65536 + GP - 1) / GP;                         // q / GP == (q * gp_magic) >> 16 for q < 64
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const float hbox = P.hbox;                                          // |crush()| <= range/2, :51
  const unsigned long long tri_lanes = ns == 64 ? ~0ull : ((1ull << ns) - 1ull);
  const int NS = SS ? SS : P.S;                                       // shadow samples = sample lanes, <= 64
  const float inv_S = SS ? ((SS & (SS - 1)) == 0 ? 1.0f / (float)(SS ? SS : 1) : 0.0f) : P.inv_S;
  const float inv_aa = (AA_X && AA_Y) ? (((AA_X * AA_Y) & (AA_X * AA_Y - 1)) == 0 ? 1.0f / (float)(AA_X * AA_Y ? AA_X * AA_Y : 1) : 0.0f) : P.inv_aa;
  const int n_pass = MULTI ? (NS + 63) >> 6 : 1;                      // passes of 64 sample lanes
  const unsigned long long active = NS >= 64 ? ~0ull : ((1ull << NS) - 1ull);     // sample lanes of a full pass

  // triangle lane i < ns holds shadow-casting triangle i: resident in registers in the brute-force build,
  // re-read from LDS once per task by level 1 in the culled build (registers are what limits occupancy)
  const int tl = lane < ns ? lane : 0;
  TriLane T;
  if (!CULL) {
    T.v0 = xyz(SC.v0[tl]); T.e1 = xyz(SC.e1[tl]); T.e2 = xyz(SC.e2[tl]); T.c = xyz(SC.c[tl]);
    T.c1 = norm1(T.c); T.e1_1 = norm1(T.e1); T.e2_1 = norm1(T.e2);
    L.r0[lane] = make_float4(T.c.x, T.c.y, T.c.z, 0.f);   // static part of record 0
  }

  RT_STAMP(0)                               // 0: staging + set-up
  Work wk, xw;                               // xw: executed-work counters of this wave (COUNT builds only)
  if (COUNT) for (int q = 0; q < 8; ++q) xw.v[q] = 0;
  // ---- job loop.  The queue has kJobHeads heads; a wave pulls from its home head until that one's jobs are
  // gone, then moves on to the next head that still has jobs (all heads peeked at with one load), and leaves when
  // none has: every head only grows, so every wave reaches the exit.
  int head = (int)((blockIdx.x * kWavesPerBlock + wave) % kJobHeads);
  // Longest jobs first: the jobs that cost more than heavy_factor4/4 x the average in the previous frame of this context
  // are pulled (phase A, second set of heads) before the plain sequence (phase B, which skips them by their flag).
  // (Handing the expensive jobs out one TASK at a time, to different waves, was built and measured: no gain at 512 rows
  // per rank, 2 % lost on the whole frame.)
  const bool lpt = !COUNT && !PROF && PC(heavy_new) != nullptr;
  unsigned int n_heavy = 0u;
  unsigned long long heavy_thr = ~0ull;
  // this wave's sum of job costs and job count live in the padding words of its RNG scratch (kept in registers
  // they were spilled: 768 B of scratch traffic per job)
  unsigned long long* const cost_acc = reinterpret_cast<unsigned long long*>(L.rng + 256);
  if (lpt && lane == 0) { cost_acc[0] = 0ull; cost_acc[1] = 0ull; }
  if (lpt) {
    n_heavy = PC(heavy_prev_state)[0] < (unsigned int)PC(heavy_cap) ? PC(heavy_prev_state)[0] : (unsigned int)PC(heavy_cap);
    const unsigned long long psum = ((unsigned long long)PC(heavy_prev_state)[3] << 32) | PC(heavy_prev_state)[2];
    const unsigned int pjobs = PC(heavy_prev_state)[4];
    if (pjobs != 0u) heavy_thr = (unsigned long long)PC(heavy_factor4) * (psum / pjobs) / 4ull;    // per TASK
  }
  // diagnostic (UOB_RT_TIMELINE): when this wave starts asking for jobs and how many it gets; kept in the padding words of
  // the second pixel's RNG scratch, not in registers
  const bool timeline = !COUNT && !PROF && PC(counters) != nullptr;
  unsigned long long* const tls = reinterpret_cast<unsigned long long*>(L.rng + kRngStride + 256);
  if (timeline && lane == 0) { tls[0] = __builtin_amdgcn_s_memrealtime(); tls[1] = 0ull; }
  bool phase_a = n_heavy != 0u;
  // Several jobs per hand-out (a hand-out is a returning device-scope atomic that stalls its wave for microseconds), fewer
  // as the queue runs out: RT_CHUNK while more than 2 T jobs per wave remain, two while more than T, then one, with T = 8 for
  // 64-pixel jobs and T = 1 for the halved jobs of short frames, which hold ~16 jobs per wave in all and would otherwise
  // never see a full hand-out (measured, ms: 1024 rows, 64-pixel jobs: 0.92 with T = 8, 1.00 with T = 1; 512 rows, 32-pixel
  // jobs: 0.570 against 0.540).
  const int grid_waves = (int)gridDim.x * kWavesPerBlock;
  const int per_wave = PC(job_tasks) * PT > 32 ? 8 * grid_waves : grid_waves;
  int next_job = -1, chunk_left = 0;                    // the rest of the last hand-out, still to do
  unsigned int listed = 0u;                             // bit i: job i of the rest of the hand-out is on last frame's list
  int chunk = PC(njobs) > 2 * per_wave ? RT_CHUNK : (PC(njobs) > per_wave ? 2 : 1);      // size of the next hand-out
  for (;;) {
  int job = 0;
  const int jt = PC(job_tasks);                 // tasks of this hand-out
  int k0 = 0, k1 = jt;                        // the tasks of the job this hand-out covers
  if (phase_a) {
    if (lane == 0) job = (int)atomicAdd(PC(job_counter) + (kJobHeads + head) * kJobHeadStride, 1u);
    unsigned int unit = (unsigned int)__builtin_amdgcn_readfirstlane(job) * kJobHeads + (unsigned int)head;
    if (PC(split_listed)) {
      // UOB_RT_SPLIT_LISTED=1: the listed jobs go out one TASK at a time (unit u = task u / n of listed job u % n, all
      // first tasks before all second ones) — for frames so short that one job's 64 undecided surface points per task
      // (25x an ordinary task, profiles/r02_wave_timeline.txt) could be the critical path.  Measured: they are not.
      const unsigned int q = unit / n_heavy;
      if (q >= (unsigned int)jt) { phase_a = false; continue; }
      unit -= q * n_heavy; k0 = (int)q; k1 = k0 + 1;
    } else if (unit >= n_heavy) { phase_a = false; continue; }
    job = (int)PC(heavy_prev)[unit];
    if (job < 0 || job >= PC(njobs)) continue;
  } else {
    if (chunk_left > 0) {
      job = next_job; next_job += kJobHeads; --chunk_left; listed >>= 1;
    } else {
      if (lane == 0) job = (int)atomicAdd(PC(job_counter) + head * kJobHeadStride, (unsigned int)chunk);
      job = __builtin_amdgcn_readfirstlane(job) * kJobHeads + head;
      next_job = job + kJobHeads; chunk_left = chunk - 1;
      // which jobs of this hand-out are on last frame's list (phase A takes care of those): ONE load for the hand-out,
      // lane i asking for job i, instead of a dependent load in front of every job
      listed = 0u;
      if (n_heavy != 0u) {
        const int jl = job + lane * kJobHeads;
        listed = (unsigned int)ballot(lane <= chunk_left && jl < PC(njobs) && PC(heavy_flags)[jl] == PC(heavy_gen));
      }
      // what this head has left decides the next hand-out
      const int left = PC(njobs) - next_job;
      chunk = left > 2 * per_wave ? RT_CHUNK : (left > per_wave ? 2 : 1);
    }
    if (job >= PC(njobs)) {
      // This head is empty: look at ALL heads at once (lane h reads head h: one memory round trip, not one per head — walking
      // them one by one kept every wave ~30 us in the kernel after its last job, a fifth of a 1024^2 frame) and move on to the
      // next one that still has jobs; none: leave.  The heads only grow, so what is empty stays empty: every wave gets here.
      chunk_left = 0;
      const unsigned int at = lane < kJobHeads ? __hip_atomic_load(PC(job_counter) + lane * kJobHeadStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      const unsigned long long avail = ballot(lane < kJobHeads && (long long)at * kJobHeads + lane < (long long)PC(njobs));
      if (avail == 0ull) break;
      const unsigned long long above = avail & ~((2ull << head) - 1ull);            // cyclically after this head
      head = __builtin_ctzll(above != 0ull ? above : avail);
      continue;
    }
    if ((listed & 1u) != 0u) continue;                                    // listed: taken care of by phase A
  }
  k0 = __builtin_amdgcn_readfirstlane(k0); k1 = __builtin_amdgcn_readfirstlane(k1);   // (the same in every lane; see `wave` above)
  RT_STAMP(7)                               // 7: waiting for the hand-out (PROF builds)
  if (timeline && lane == 0) tls[1] += 1ull;
  const unsigned long long job_t0 = lpt ? __builtin_amdgcn_s_memtime() : 0ull;
  // Rows are handed out from the middle of the rank's rows outwards: segments differ 10x in cost, and the kernel
  // ends when the last job does, so the last jobs should be cheap ones — the top and bottom rows of a view
  // usually are (background, plain walls).
  const int jrow = (int)div_magic((uint32_t)job, PC(nseg_magic));
  const int mid = (PC(owned_rows) + 1) >> 1;
  const int lr = (jrow & 1) ? mid + (jrow >> 1) : mid - 1 - (jrow >> 1);
  const int JP = BIGAA ? jt / chunks : jt * PT;    // pixels of this hand-out
  const int x0 = (job - jrow * PC(nseg)) * JP;
  const int y = band_global_row_cold(P, lr);
  f3 outc = mk(0.f, 0.f, 0.f);
  // Level 1 of a WIDER point set than the task's, reused by the job's next tasks while their points stay inside it (below)
  bool jk_valid = false, jk_sph = false, jk_blocked = false;
  f3 jk_D0 = mk(0.f, 0.f, 0.f);
  float jk_ed = 0.0f;
  unsigned long long jk_K = 0ull;
  // Triangles a primary ray of this job may hit, bounded once for the job's 64 x 1 pixels (all AA samples): the
  // rays leave the camera through a sub-pixel rectangle, see primary_clear.  (Per task the rectangle is 8x
  // narrower and a triangle or so fewer survives, but the bound itself costs more than that triangle's tests.)
  unsigned long long Kp_job = n == 64 ? ~0ull : ((1ull << n) - 1ull);
  bool sph_job = P.nsph > 0;
  if (CULL) {
    const int lnJ = opaque(lane);
    const float Xlo = (float)(x0 * aa_x) - P.half_wx;
    const float Ylo = ((float)(y * aa_y) - P.half_hy) * sy;
    // a hand-out narrower than a job (none at present) would only make the box generous
    const f3 wc = mk(Xlo + PC(job_hx), Ylo + PC(job_hy), P.focal);
    const f3 duc = mk(P.rot[0] * wc.x + P.rot[1] * wc.y + P.rzf[0], P.rot[4] * wc.x + P.rot[5] * wc.y + P.rzf[1],
                      P.rot[8] * wc.x + P.rot[9] * wc.y + P.rzf[2]);
    const f3 eu = mk(PC(job_eu[0]), PC(job_eu[1]), PC(job_eu[2]));
    const float dumax = fmaxf(fmaxf(fabsf(duc.x) + eu.x, fabsf(duc.y) + eu.y), fabsf(duc.z) + eu.z);
    const int ti = lnJ < n ? lnJ : 0;
    const float4 c4 = S.c[ti];
    const bool clear = primary_clear(duc, eu, dumax, xyz(c4), c4.w, xyz(S.pc[ti]), xyz(S.qc[ti]));
    if (dumax < 1e30f) Kp_job &= ~ballot(clear);
    // ... and whether any of them can touch a sphere at all (else the two quadratic tests per ray are skipped)
    if (RT_OPT_SPHJOB && P.nsph > 0 && dumax < 1e30f)
      sph_job = ballot(sphere_bundle_maybe(P, lane, mk(P.cam[0], P.cam[1], P.cam[2]), 0.0f, duc, bsqrt(dot3(duc, duc)),
                                           1.0001f * bsqrt(dot3(eu, eu)), false)) != 0ull;
  }
  RT_STAMP(0)                               // 0: job set-up (primary-ray bounds of the job)
  for (int k = k0; k < k1; ++k) {
    const int lnA = opaque(lane);
    // ---- phase 1: 64 primary rays, lnA = (pixel, AA sample) -----------------------------------------
    const int kp = BIGAA ? k / chunks : k;     // BIGAA: the pixel (within the job) this task belongs to, and which chunk of its samples
    const int chunk = BIGAA ? k - kp * chunks : 0;
    const int pA = BIGAA ? 0 : ((AA_X && AA_Y) ? lnA / (AA_X * AA_Y ? AA_X * AA_Y : 1) : (lnA * P.aa_magic) >> 16);   // lnA / aa
    const int pj = kp * PT + pA;               // pixel of this lnA within the job
    const int a = (lnA - pA * aa) + 64 * chunk;   // AA sample index dy*rx+dx, kernels.cl:395
    const int x = x0 + pj;
    const bool valid = x < P.W && pA < PT && (!BIGAA || a < aa_full);
    const int ay = AA_X ? a / (AA_X ? AA_X : 1) : (a * P.aax_magic) >> 16;        // a / aa_x (a < 256)
    Ray ray = primary_ray(P, x, y, a - ay * aa_x, ay, aa_x, aa_y, sy);
    bool lit = false, secondary = false;
    // triangles a primary ray of this job may hit (read back through readfirstlane: the set is the same in every lane, and
    // the loop over it then runs on the scalar unit)
    const unsigned long long Kp = uniform64(Kp_job);
    if (valid) {
      if (CULL) closest_hit_primary_masked(S, P, ray, Kp, sph_job);
      else closest_hit_primary<false>(S, P, ray, wk);
      if (ray.tri != -1) {
        // ---- phase 2: mirror / glass bounces (kernels.cl:342-365) -----------------------------------
        if (ray.col.w <= 0.0f) { secondary = true; if (!CULL) lit = bounce_to_diffuse<false>(S, P, ray, wk); }
        else lit = true;
      }
    }
    if (CULL) {
      // The bounce loop of secondary_light (:342-365) run by the whole wave, so that each round's closest-hit
      // search visits only the triangles its rays can reach: the bounce rays of neighbouring pixels leave a
      // smooth surface nearby in similar directions, and one lane = triangle bound over their origin box and
      // direction box (task_bound with the distance rule off) usually leaves a handful of the scene's triangles.
      bool bouncing = secondary;
      for (int b = 0; b < P.bounces; ++b) {
        const bool act = bouncing && ray.col.w <= 0.0f;               // this lane's loop condition, :345
        const unsigned long long actm = ballot(act);
        if (actm == 0ull) break;
        if (act) ray = (ray.col.w == 0.0f) ? reflect_ray(ray) : refract_ray(ray);
        unsigned long long Kb = n == 64 ? ~0ull : ((1ull << n) - 1ull);
        {
          const int lnK = opaque(lane);
          const f3 o = ray.start, d = ray.dir;
          const float mag = fmaxf(norm_inf(o), norm_inf(d));
          const bool fin = act && mag < 1e30f;                         // false for NaN as well
          const bool isnan_ = act && !(mag == mag);                    // a NaN ray hits nothing whatever the set
          const unsigned long long finm = ballot(fin);
          if (finm != 0ull && ballot(act && !fin && !isnan_) == 0ull) {
            const float big = 3.0e38f;
            const f3 olo = mk(wave_min(fin ? o.x : big), wave_min(fin ? o.y : big), wave_min(fin ? o.z : big));
            const f3 ohi = mk(wave_max(fin ? o.x : -big), wave_max(fin ? o.y : -big), wave_max(fin ? o.z : -big));
            const f3 dlo = mk(wave_min(fin ? d.x : big), wave_min(fin ? d.y : big), wave_min(fin ? d.z : big));
            const f3 dhi = mk(wave_max(fin ? d.x : -big), wave_max(fin ? d.y : -big), wave_max(fin ? d.z : -big));
            const f3 s0 = 0.5f * (olo + ohi), D0 = 0.5f * (dlo + dhi);
            const float es = 0.5001f * fmaxf(fmaxf(ohi.x - olo.x, ohi.y - olo.y), ohi.z - olo.z) + 1e-6f * norm1(s0);
            const float ed = 0.5001f * fmaxf(fmaxf(dhi.x - dlo.x, dhi.y - dlo.y), dhi.z - dlo.z) + 1e-6f * norm1(D0);
            const float dmx = fmaxf(fmaxf(fmaxf(fabsf(dlo.x), fabsf(dhi.x)), fmaxf(fabsf(dlo.y), fabsf(dhi.y))), fmaxf(fabsf(dlo.z), fabsf(dhi.z)));
            TriLane Tb;
            const int tb_i = lnK < n ? lnK : 0;
            Tb.v0 = xyz(S.v0[tb_i]); Tb.e1 = xyz(S.e1[tb_i]); Tb.e2 = xyz(S.e2[tb_i]); Tb.c = xyz(S.c[tb_i]);
            Tb.c1 = norm1(Tb.c); Tb.e1_1 = norm1(Tb.e1); Tb.e2_1 = norm1(Tb.e2);
            const float dl = 1.7321f * dmx * 1.0001f;                   // >= |d|_2 of every ray
            const Bound bb = task_bound(Tb, s0, D0, es, ed, 2e-6f * dl, 0.0f, dl);
            Kb &= ~ballot(bb.clear);
          } else if (finm == 0ull) {
            Kb = 0ull;                                                  // only NaN rays: they hit nothing
          }
        }
        if (act) {
          closest_hit_masked(S, P, ray, Kb);
          if (ray.tri != -1 && ray.col.w > 0.0f) { lit = true; bouncing = false; }
        }
      }
    }
    RT_STAMP(1)                             // 1: primary rays + bounces
    const int lnB = opaque(lane);
    // per-lnB light set-up of direct_light, :323-326
    const f3 dir = light - ray.P;
    const f3 start = ray.P + 0.0001f * dir;
    const float radius_sq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    const float term = (16.0f * fmaxf(dot3(dir, ray.N), 0.0f)) / (4.0f * 3.14159274f * radius_sq);

    // A surface point that faces away from the light — term == 0: max(dot, 0) is 0 (:335) — adds (mask * 0) / (4 pi r^2) = +-0 per
    // sample whatever its masks are, and the sum stays +0: its samples need no test at all.  (Exactly zero only: a NaN or
    // infinite term compares unequal and takes the full path.)  Half the surface of any closed mesh, the far faces of the blocks.
    const bool slit = lit && !(term == 0.0f);
    // ---- phase 3: shadows of the lit surface points ---------------------------------------------------
    int unshadowed = NS;                        // samples of this lnB's surface point that reach the light
    unsigned long long work = ballot(slit);      // lanes whose samples must really be tested (level 3)
    // need: per surface point, the casters (as positions among K's set bits) whose samples must be tested; with at most 32
    // casters (the static-layout build) it is one register, and one lane read per point in level 3
    using need_t = typename std::conditional<STRIDE == 32, uint32_t, unsigned long long>::type;
    unsigned long long K = tri_lanes, sphmask = P.nsph > 0 ? ~0ull : 0ull;
    need_t need = (need_t)~(need_t)0;
    if (CULL && work != 0ull) {
      // bounds used by the cull (never by the shading): |dir|, jitter half-width with rounding slack
      const float dlen = bsqrt(radius_sq);
      const float hh = 1.002f * hbox + 2e-6f * (dlen + hbox);
      float dminlen = dlen - 1.7321f * hh;
      const bool sane = slit && (radius_sq > 1e-18f) && (radius_sq < 1e30f);
      if (!sane || !(dminlen > 0.0f)) dminlen = 0.0f;                  // disables the distance rule
      const float dk = dlen * 1.000004f;
      // level 1: all such points of the task at once, lnB = triangle
      bool task_blocked = false;
      SphereBound sb;
      sb.maybe = false; sb.all_blocked = false;
      bool task_sph = P.nsph > 0;                  // may any shadow ray of the task touch a shadow-casting sphere?
      {
        const bool all_sane = ballot(slit && !sane) == 0ull;
        // A bound is a statement about a SET of shadow rays and holds for every subset.  The set bounded here is wider than the
        // task's own — every direction within PC(l1_inflate) x the task's spread of a reference direction — and the job's next
        // tasks (neighbouring pixels of the same row, mostly on the same surface) reuse its outcome while their points'
        // directions stay inside it: one wave reduction instead of the whole of level 1 (the start points follow the
        // directions, see below).  A wider set keeps a few more casters for level 2, whose first clause is cheap: headline frame
        // 2.68 ms without, 2.60 with the set around the task's first lane (x4), 2.53 around its LAST lane (x3.5: the next tasks
        // lie further right, so the same width reaches one task further); widths of 2, 2.5, 3, 4, 5: 2.58, 2.55, 2.57, 2.56, 2.60.
        // (Per-component widths — the points of a row on a plane differ along one line — keep fewer casters still, but three
        // reductions and the weighted sums cost more than that: 2.58; profiles/r03_l1_width.txt.)
        bool reuse = false;
        if (RT_OPT_JOBK && jk_valid && all_sane) {
          const f3 dj = dir - jk_D0;
          reuse = wave_max_pos(slit ? norm_inf(dj) : 0.0f) <= jk_ed;
        }
        if (reuse) {
          K = jk_K; task_sph = jk_sph; task_blocked = jk_blocked;
        } else {
        const int jr = RT_OPT_JOBK ? 63 - __builtin_clzll(work) : __builtin_ctzll(work);
        const f3 s0 = mk(rl(start.x, jr), rl(start.y, jr), rl(start.z, jr));
        const f3 D0 = mk(rl(dir.x, jr), rl(dir.y, jr), rl(dir.z, jr));
        // One wave reduction instead of four: the points' start and dir move together (start = X + 1e-4 dir,
        // dir = light - X, kernels.cl:323-324), so |start - s0| <= (1 + 1e-4) |dir - D0| + roundings of the
        // coordinates, and |dir| lies within sqrt(3) ed of the reference point's.
        const f3 dd = dir - D0;
        const float ed_task = wave_max_pos(slit ? norm_inf(dd) : 0.0f);
        // (no wider than the tasks this job still has to come can use: its last task — and every task of a one-task job, the
        // 16-pixel jobs of a 1024^2 frame — bounds its own set)
        const float left = (float)(k1 - 1 - k);
        const float ed = RT_OPT_JOBK ? ed_task * (left > 0.0f ? fminf(PC(l1_inflate), left + 0.5f) : 1.0f) : ed_task;
        const float dlen0 = rl(dlen, jr);
        const float dlen_max = (dlen0 + 1.7321f * ed) * 1.000001f;
        const float dlen_min = fmaxf(dlen0 - 1.7321f * ed, 0.0f) * 0.999999f;
        const float es = 1.0002f * ed + 2e-6f * (P.light_inf + dlen_max);
        jk_valid = false;
        if (all_sane && es < 1e30f && ed < 1e30f) {                    // finite, non-degenerate
          const float hh_task = 1.002f * hbox + 2e-6f * (dlen_max + hbox);
          if (RT_OPT_TASKSPH && P.nsph > 0)       // every sample direction lies within sqrt(3) (ed + hh) of D0, every start within es of s0
            task_sph = ballot(sphere_bundle_maybe(P, lane, s0, es, D0, dlen0, 1.7321f * (1.001f * ed + hh_task), true)) != 0ull;
          TriLane T1;
          T1.v0 = xyz(SC.v0[(lnB < ns ? lnB : 0)]); T1.e1 = xyz(SC.e1[(lnB < ns ? lnB : 0)]); T1.e2 = xyz(SC.e2[(lnB < ns ? lnB : 0)]); T1.c = xyz(SC.c[(lnB < ns ? lnB : 0)]);
          T1.c1 = norm1(T1.c); T1.e1_1 = norm1(T1.e1); T1.e2_1 = norm1(T1.e2);
          // The bound from the LIGHT's side (rt_wave_common.h light_bundle_bound, the mesh kernel's level 1): start and direction
          // of a shadow ray are tied together (start = X + 1e-4 dir, X = light - dir), so the direction box counts once instead
          // of widening both the start box and the direction box as task_bound must for an arbitrary point set: fewer
          // survivors K for level 2.
#if RT_OPT_LIGHTSIDE
          const Bound tb = light_bundle_bound(T1, light, s0, es, D0, ed, hh_task, dlen_min, dlen_max, P.light_inf + dlen_max + norm1(T1.v0));
#else
          const Bound tb = task_bound(T1, s0, D0, es, ed, hh_task, dlen_min, dlen_max);
#endif
          K = tri_lanes & ~ballot(tb.clear);
          task_blocked = (tri_lanes & ballot(tb.all_blocked)) != 0ull;
          if (RT_OPT_JOBK) {
            // what the next tasks compare with: directions within 0.9999 ed of D0 (the bound itself allows 1.001 ed and more)
            jk_valid = true; jk_D0 = D0; jk_ed = uniform(ed * 0.9999f); jk_K = K; jk_sph = task_sph; jk_blocked = task_blocked;
          }
        }
        }
      }
      if (task_sph && !task_blocked && sane) sb = spheres_point(P, start, dir, dlen, hh);
      sphmask = ballot(slit && P.nsph > 0 && (sb.maybe || !sane));
      const bool sph_blocked = sane && sb.all_blocked;
      RT_STAMP(2)                           // 2: light set-up + level 1
      if (task_blocked) {
        unshadowed = 0; work = 0ull;
        if (COUNT) xw.v[5] += 1;
      } else {
        // level 2: per surface point, lnB = point, over the triangles K that survived level 1

        bool blocked = sph_blocked;
        need = (need_t)0;
        int pos = 0;
        for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull, ++pos) {
          const int kq = __builtin_ctzll(kk);
          // two thirds of the pairs that reach level 2 are a surface against its own plane: t < 0 for every
          // sample, by the signs of det(A0) (exact) and det(A) alone — the first clause of point_bound
          const f3 c_ = xyz(SC.c[kq]);
          const float nA0_ = detc(start - xyz(SC.v0[kq]), c_), D0_ = detc(-dir, c_), Dl_ = hh * norm1(c_);
          const bool tneg = sane && ((D0_ - Dl_ > 0.0f && nA0_ < -1e-18f) || (D0_ + Dl_ < 0.0f && nA0_ > 1e-18f));
          if (ballot(slit && !tneg) == 0ull) continue;
          const Bound pb = point_bound(start, dir, hh, dlen, dminlen, dk, xyz(SC.v0[kq]), xyz(SC.e1[kq]), xyz(SC.e2[kq]),
                                       xyz(SC.c[kq]), slit && !tneg && !blocked);
          if (!pb.clear || !sane) need |= (need_t)((need_t)1 << pos);
          blocked = blocked || (sane && pb.all_blocked);
        }
        if (blocked) unshadowed = 0;
        work = ballot(slit && !blocked && (need != (need_t)0 || ((sphmask >> lnB) & 1ull) != 0ull));
        if (COUNT) { xw.v[4] += (unsigned)__popcll(ballot(slit && !blocked && need == (need_t)0)); if (work == 0ull) xw.v[5] += 1; }
      }
    }

    RT_STAMP(3)                             // 3: level 2
    const int lnC = opaque(lane);
    // level 3 reads the surface points lane = sample: they go to LDS only for the tasks that get there
    if (work != 0ull) {
      __builtin_amdgcn_wave_barrier();
      L.h0[lnC] = make_float4(start.x, start.y, start.z, radius_sq);
      L.h1[lnC] = make_float4(dir.x, dir.y, dir.z, 0.f);
      __builtin_amdgcn_wave_barrier();
    }
    // level 3 / brute force: the reference's sample test, one surface point at a time
    const int GL = GP * aa;                     // lanes per RNG group
    for (int g = 0; g * GP < PT && work != 0ull; ++g) {
      const unsigned long long gm = (work >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
      if (gm == 0ull) continue;
      for (int pass = 0; pass < n_pass; ++pass) {
      const int first_s = pass << 6;                                   // first sample of this pass
      const int cnt_s = NS - first_s < 64 ? NS - first_s : 64;         // samples in it
      const unsigned long long act = (!MULTI || cnt_s == 64) ? active : ((1ull << cnt_s) - 1ull);
      // xorshift streams of the GP pixels of this group: lnC c -> (pixel c/3, component c%3), :319,:331
      // lane -> (segment, pixel, component): kRngSegs lanes share a stream, lane `seg` writing samples [13 seg, 13 seg + 13)
      const int q3 = lnC / 3, comp = lnC - 3 * q3;
      const int seg = (q3 * gp_magic) >> 16, pp = q3 - seg * GP;         // q3 / GP, q3 % GP
      if (seg < kRngSegs) {
        uint32_t* dst = L.rng + pp * kRngStride + comp;
        uint32_t s;
        if (!MULTI || pass == 0) {
          const int gid = pixel_global_id(P, x0 + kp * PT + g * GP + pp, y);
          const uint32_t seed = comp == 0 ? (uint32_t)gid : (uint32_t)((float)gid * (comp == 1 ? 91.0f : 19.0f));
          s = xorshift(seed);
        } else {
          s = dst[63 * 4];                                               // the stream goes on where the last pass left it
        }
        if (seg > 0) s = rng_jump(s, seg);
        const int it0 = seg * kRngSegLen;
#pragma unroll 1
        for (int j = 0; j < kRngSegLen; ++j) { s = xorshift(s); if (it0 + j < cnt_s) dst[(it0 + j) * 4] = s; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      RT_STAMP(4)                           // 4: xorshift streams
      if (CULL && !COUNT && !MULTI && ((SS > 0 && SS <= 32) || (SS == 0 && STRIDE == 32 && !BIGAA && NS <= 16))) {
        // few samples: lane = (surface point, sample), 64 / SP points of one pixel per instance, two instances per pass
        // (the instantiations specialised on the sample count; the static-layout generic one for up to 16 samples)
        constexpr int SP = (SS > 16) ? 32 : 16, PK = 64 / SP;
        const unsigned long long aam = aa == 64 ? ~0ull : ((1ull << aa) - 1ull);
        unsigned long long rest = gm;
        while (rest != 0ull) {
          unsigned long long pmv[2] = {0ull, 0ull};
          int basev[2] = {0, 0}, ppv[2] = {0, 0};
#pragma unroll
          for (int inst = 0; inst < 2; ++inst) {
            if (rest == 0ull) break;
            const int b0 = __builtin_ctzll(rest);
            const int pp = (b0 * (aa == 1 ? 65536 : P.aa_magic)) >> 16;     // b0 / aa (b0 < 64)
            unsigned long long pm = (rest >> (pp * aa)) & aam, take = 0ull;
            for (int i = 0; i < PK && pm != 0ull; ++i) { take |= pm & (0ull - pm); pm &= pm - 1ull; }   // its lowest PK points
            pmv[inst] = take; basev[inst] = g * GL + pp * aa; ppv[inst] = pp;
            rest &= ~(take << (pp * aa));
          }
          wave_unshadowed_packed<SP>(P, SC, L, lnC, pmv[0], basev[0], L.rng + ppv[0] * kRngStride, pmv[1], basev[1],
                                     L.rng + ppv[1] * kRngStride, K, (unsigned long long)need, sphmask, NS, unshadowed);
        }
      } else
      for (int pp = 0; pp < GP; ++pp) {
        unsigned long long pm = (gm >> (pp * aa)) & (aa == 64 ? ~0ull : ((1ull << aa) - 1ull));
        if (pm == 0ull) continue;
        const uint32_t* src = L.rng + pp * kRngStride + lnC * 4;     // lnC = sample index within the pass
        const f3 jit = mk(crush1(src[0], P.spread), crush1(src[1], P.spread), crush1(src[2], P.spread));
        const int base = g * GL + pp * aa;
        while (pm != 0ull) {
          const int j = base + __builtin_ctzll(pm);
          pm &= pm - 1ull;
          if (CULL) {
            const int j2 = pm != 0ull ? base + __builtin_ctzll(pm) : j;   // second point of the pair (or j again)
            pm &= pm - 1ull;                                             // 0 & anything stays 0
            unsigned long long n1, n2;
            if (STRIDE == 32) {
              n1 = (unsigned)__builtin_amdgcn_readlane((int)(uint32_t)need, j);
              n2 = (unsigned)__builtin_amdgcn_readlane((int)(uint32_t)need, j2);
            } else {
              const unsigned long long nd64 = (unsigned long long)need;
              n1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(nd64 >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)nd64, j);
              n2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(nd64 >> 32), j2) << 32) | (unsigned)__builtin_amdgcn_readlane((int)nd64, j2);
            }
            const Count2 c2 = wave_unshadowed_pair<COUNT>(P, SC, L, lnC, j, j2, K, n1 | n2, (sphmask >> j) & 1ull,
                                                          (sphmask >> j2) & 1ull, jit, act, xw);
            const int before = (MULTI && pass > 0) ? unshadowed : 0;     // j2 may be j again: add once
            if (lnC == j2) unshadowed = before + c2.b;
            if (lnC == j) unshadowed = before + c2.a;
          } else {
            const int cnt = wave_unshadowed_all<COUNT>(P, T, L, lnC, ns, j, jit, act, xw);
            if (lnC == j) unshadowed = (MULTI && pass > 0 ? unshadowed : 0) + cnt;
          }
        }
      }
      if (MULTI) __builtin_amdgcn_wave_barrier();      // the next pass rewrites the scratch
      }                                                 // passes
      __builtin_amdgcn_wave_barrier();          // scratch is rewritten by the next group
      RT_STAMP(5)                           // 5: level 3 sample tests
    }
    RT_STAMP(5)

    const int lnD = opaque(lane);
    // ---- phase 4: shade lnD-parallel (direct_light's sum :335, then :354 / :421-422) ----------------
    if (COUNT) {      // outcome of the sampled points: fully lit / fully blocked after all (the rest is true penumbra)
      xw.v[6] += (unsigned)__popcll(work & ballot(lit && unshadowed == NS));
      xw.v[7] += (unsigned)__popcll(work & ballot(lit && unshadowed == 0));
    }
    f3 contrib = mk(0.f, 0.f, 0.f);
    {
      // direct_light's running sum (:335): the same term added once per unblocked sample, in sequence.
      // Most tasks have every lit lnD fully lit: then the adds need no per-lnD predicate.
      float total = 0.0f;
      // (eight adds per trip: one add per trip is a taken branch per add, and the chain is on every task's critical path)
      if (ballot(lit && unshadowed != NS) == 0ull) {
#pragma unroll 8
        for (int i = 0; i < NS; ++i) total += term;
      } else {
        if (unshadowed < NS) total += 0.0f * term;        // a blocked sample adds 0*term (NaN/inf-faithful)
#pragma unroll 8
        for (int i = 0; i < NS; ++i) if (i < unshadowed) total += term;
      }
      if (lit) {
        const float l = 0.5f + div_count(total, NS, inv_S);
        if (secondary) { const float kk = 0.9f * l; contrib = mk(kk * ray.col.x, kk * ray.col.y, kk * ray.col.z); }
        else contrib = mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
      }
    }
    // sum the AA rays of each pixel in index order (final_color_total +=, :415-425); a ray without a
    // contribution adds +0, which leaves the running sum unchanged bit for bit
    if (BIGAA) {
      // the pixel's samples of this chunk in index order onto its running sum (a lane without a ray adds +0: a no-op)
      if (chunk == 0) aa_run = mk(0.f, 0.f, 0.f);
#pragma unroll 8
      for (int r = 0; r < 64; ++r) aa_run = aa_run + mk(rl(contrib.x, r), rl(contrib.y, r), rl(contrib.z, r));
      if (chunk == chunks - 1 && lnD == kp) outc = aa_run;
    } else {
    const f3 acc = aa_sum(contrib, aa, ((AA_X && AA_Y) ? lnD / (AA_X * AA_Y ? AA_X * AA_Y : 1) : (lnD * P.aa_magic) >> 16) * aa);
      // pixel pj's sum lives in lanes [pj*aa, pj*aa+aa); hand it to output lnD (k*PT + pj)
      {
        const int rel = lnD - k * PT;
        const int srcl = (rel >= 0 && rel < PT) ? rel * aa : 0;
        const f3 v = mk(shfl(acc.x, srcl), shfl(acc.y, srcl), shfl(acc.z, srcl));
        if (rel >= 0 && rel < PT) outc = v;
      }
    }
    RT_STAMP(6)                             // 6: shading + AA sum
  }
  // ---- store: the job's consecutive pixels, one coalesced access per wave ------------------------------
  const int x = x0 + lane;
  if (!COUNT && !PROF && (BIGAA ? lane < JP : (lane >= k0 * PT && lane < k1 * PT)) && x < P.W) {
    const f3 c = mk(div_count(outc.x, aa_full, inv_aa), div_count(outc.y, aa_full, inv_aa), div_count(outc.z, aa_full, inv_aa));
    const size_t o = (size_t)(PC(out_global) ? y : lr) * P.W + x;
    PC(out_argb)[o] = pack_argb(c);
    if (PC(out_rgb)) PC(out_rgb)[o] = make_float4(c.x, c.y, c.z, 1.0f);
  }
  if (lpt) {
    const unsigned long long cost = __builtin_amdgcn_s_memtime() - job_t0;
    if (lane == 0) { cost_acc[0] += cost; cost_acc[1] += (unsigned long long)(k1 - k0); }
    // Listed once per frame, by whichever wave finds one of its tasks expensive first — together with its two neighbours in
    // the row (lanes 1, 2): next frame's light or camera has moved, and with them the penumbra, by less than a job's width
    // (update() moves the light by at most 0.025 per frame, skeleton.cpp:290-298); a neighbour that turns out cheap only
    // starts early.  (animated light, 20 frames: 3.39 -> ms mean with the static frame unchanged)
    if (cost > heavy_thr * (unsigned long long)(k1 - k0) && lane < (PC(heavy_dilate) ? 3 : 1)) {
      const int jn = job + (lane == 1 ? -1 : (lane == 2 ? 1 : 0));
      const int col = jn - jrow * PC(nseg);
      if (col >= 0 && col < PC(nseg)) {
        // heavy_flags_new[j] == gen + 1: job j is on the list this frame builds (each job once)
        const unsigned int was = atomicMax(PC(heavy_flags_new) + jn, PC(heavy_gen) + 1u);
        if (was < PC(heavy_gen) + 1u) {
          const unsigned int at = atomicAdd(PC(heavy_new_state), 1u);
          if (at < (unsigned int)PC(heavy_cap)) PC(heavy_new)[at] = (unsigned int)jn;
          else atomicExch(PC(heavy_flags_new) + jn, was);     // list full: not listed after all
        }
      }
    }
  }
  }                                          // ---- end of the job loop ---------------------------------------
  if (lpt && lane == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(PC(heavy_new_state) + 2), cost_acc[0]);
    atomicAdd(PC(heavy_new_state) + 4, (unsigned int)cost_acc[1]);
  }
  if (timeline && lane == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), t0 = tls[0];
    unsigned long long* const rec = PC(counters) + 3 * (size_t)(blockIdx.x * kWavesPerBlock + wave);   // one record per wave
    rec[0] = t0; rec[1] = t1; rec[2] = tls[1] | ((unsigned long long)n_heavy << 32);   // (+ the length of the list this frame started from)
  }
  if (PROF) {
    if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&PC(counters)[q], prof[q]);
    return;
  }
  if (COUNT) {                               // wave-uniform counters: lane 0 publishes them
    if (lane == 0) for (int q = 0; q < 8; ++q) if (xw.v[q]) atomicAdd(&PC(counters)[q], xw.v[q]);
  }
}

template __global__ void rt_draw_wave<false, false>(const FrameParams);
template __global__ void rt_draw_wave<true, false>(const FrameParams);
template __global__ void rt_draw_wave<false, true>(const FrameParams);
template __global__ void rt_draw_wave<true, true>(const FrameParams);
template __global__ void rt_draw_wave<true, false, true>(const FrameParams);
template __global__ void rt_draw_wave<true, false, false, 32>(const FrameParams);
template __global__ void rt_draw_wave<true, false, false, 0, true>(const FrameParams);
// specialised on (AA grid, samples): BASELINE.json's configurations and the reference as shipped
template __global__ void rt_draw_wave<true, false, false, 32, false, 4, 2, 64>(const FrameParams);   // the headline frame
template __global__ void rt_draw_wave<true, false, false, 32, false, 2, 2, 64>(const FrameParams);
template __global__ void rt_draw_wave<true, false, false, 32, false, 2, 2, 16>(const FrameParams);   // configs[1]
template __global__ void rt_draw_wave<true, false, false, 32, false, 2, 2, 10>(const FrameParams);   // the reference's own constants, configs[2]
template __global__ void rt_draw_wave<false, false, false, 0, true>(const FrameParams);
template __global__ void rt_draw_wave<true, false, false, 0, false, 0, 0, 0, true>(const FrameParams);   // 65..256 AA samples per pixel

bool wave_kernel_supports(const FrameParams& P) {
  const int aa = P.aa_x * P.aa_y;
  // (more than 64 AA samples per pixel: the chunked instantiation exists for <= 64 shadow samples, with the cull)
  return P.S >= 1 && P.S <= 4096 && aa >= 1 && (aa <= 64 || (aa <= 256 && P.S <= 64)) && P.n >= 1 && P.n <= 64 && P.n_shadow >= 1 &&
         P.spread >= 0.0f;
}

static size_t wave_kernel_lds(const FrameParams& P, bool cull) {
  return (size_t)P.n * kLdsRecords * sizeof(float4) + (size_t)((P.n + 3) & ~3) * sizeof(int) +
         (size_t)P.n_shadow * 4 * sizeof(float4) + kWavesPerBlock * (size_t)wave_lds_bytes(cull);
}

int wave_blocks_per_cu(bool leave_room) { return leave_room ? RT_MIN_WAVES - 1 : RT_MIN_WAVES; }

static dim3 wave_grid(const FrameParams& P) {
  const int resident = P.wave_blocks > 0 ? P.wave_blocks : 256 * RT_MIN_WAVES;     // set per device by the host API
  const int needed = (P.njobs + kWavesPerBlock - 1) / kWavesPerBlock;
  return dim3(needed < resident ? (needed > 0 ? needed : 1) : resident);
}

void launch_wave_prof(const FrameParams& P, hipStream_t stream) {
  hipMemsetAsync(P.job_counter, 0, 2 * kJobHeads * kJobHeadStride * sizeof(unsigned int), stream);
  hipLaunchKernelGGL((rt_draw_wave<true, false, true>), wave_grid(P), dim3(64 * kWavesPerBlock), wave_kernel_lds(P, true), stream, P);
}

void launch_wave(const FrameParams& P, bool cull, bool count, hipStream_t stream) {
  const dim3 block(64 * kWavesPerBlock);
  const dim3 grid = wave_grid(P);
  const size_t lds_bytes = wave_kernel_lds(P, cull);
  // one memset: the queue heads, and this frame's HeavyState, which lies directly before or behind them
  const size_t heads_bytes = 2 * kJobHeads * kJobHeadStride * sizeof(unsigned int), state_bytes = kJobHeadStride * sizeof(unsigned int);
  if (P.heavy_new != nullptr && !count) {
    const bool before = P.heavy_new_state < P.job_counter;
    hipMemsetAsync(before ? (void*)P.heavy_new_state : (void*)P.job_counter, 0, heads_bytes + state_bytes, stream);
  } else {
    hipMemsetAsync(P.job_counter, 0, heads_bytes, stream);
  }
  if (P.aa_x * P.aa_y > 64) {               // (callers: cull on, no counting — launch_frame / rt_count_executed check)
    hipLaunchKernelGGL((rt_draw_wave<true, false, false, 0, false, 0, 0, 0, true>), grid, block, lds_bytes, stream, P);
  } else if (count) {
    if (cull) hipLaunchKernelGGL((rt_draw_wave<true, true>), grid, block, lds_bytes, stream, P);
    else hipLaunchKernelGGL((rt_draw_wave<false, true>), grid, block, lds_bytes, stream, P);
  } else if (P.S > 64) {
    if (cull) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 0, true>), grid, block, lds_bytes, stream, P);
    else hipLaunchKernelGGL((rt_draw_wave<false, false, false, 0, true>), grid, block, lds_bytes, stream, P);
  } else {
    const bool spec = cull && P.n <= 32 && !P.no_specialise;
    if (spec && P.aa_x == 4 && P.aa_y == 2 && P.S == 64) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 32, false, 4, 2, 64>), grid, block, 0, stream, P);
    else if (spec && P.aa_x == 2 && P.aa_y == 2 && P.S == 64) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 32, false, 2, 2, 64>), grid, block, 0, stream, P);
    else if (spec && P.aa_x == 2 && P.aa_y == 2 && P.S == 16) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 32, false, 2, 2, 16>), grid, block, 0, stream, P);
    else if (spec && P.aa_x == 2 && P.aa_y == 2 && P.S == 10) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 32, false, 2, 2, 10>), grid, block, 0, stream, P);
    else if (cull && P.n <= 32) hipLaunchKernelGGL((rt_draw_wave<true, false, false, 32>), grid, block, 0, stream, P);
    else if (cull) hipLaunchKernelGGL((rt_draw_wave<true, false>), grid, block, lds_bytes, stream, P);
    else hipLaunchKernelGGL((rt_draw_wave<false, false>), grid, block, lds_bytes, stream, P);
  }
}

}  // namespace uobrt
