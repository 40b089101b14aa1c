// rt_kernel_wave.hip — the headline kernel: 64 shadow samples == one 64-lane wavefront.
//
// The reference spends ~98 % of its ray-primitive tests in direct_light's shadow loop
// (Source/kernels.cl:313-340 -> in_shadow :243-311): for ONE surface point, S jittered rays towards the
// area light are tested against every triangle.  On gfx950 a wavefront is 64 lanes, so for S = 64 this
// kernel maps   lane = shadow sample   and processes surface points one after another per wave:
//
//   * everything that depends only on (surface point, triangle) — b = start - v0, det(b,e1,e2),
//     cof(b,e2), cof(e1,b): 26 of the ~45 FP32 operations of one test — is computed ONCE per surface
//     point, lane-parallel over triangles (lane i = triangle i), and handed to the 64 sample lanes
//     through per-wave LDS records that every lane reads at the same address (an LDS broadcast);
//   * all 64 lanes test the same triangle against the same surface point, so the two-stage test of
//     in_shadow (t first, u/v only if t passes, :266) becomes a WAVE-UNIFORM branch: the u/v stage runs
//     only when some lane's t passes, with no divergence;
//   * any-hit early-out is a wave ballot: the triangle loop ends as soon as every lane is shadowed;
//   * (CULL) while the triangle lanes hold the per-surface-point terms they also bound them over the
//     whole jitter box of the area light (interval arithmetic on the same determinants, with explicit
//     slack for every FP32 rounding the test performs): a triangle for which NO sample can pass the
//     reference's own comparisons is dropped before the sample loop.  The surviving set is a wave
//     ballot; the result is bit-identical to testing all triangles (tests/test_gpu_parity.py).
//
// A wave owns 64 consecutive pixels of one image row (so its framebuffer store is one coalesced 256-B
// ARGB / 1-KiB float4 access).  It walks them in `aa` tasks of 64 primary rays (64/aa pixels x aa AA
// rays): phase 1 traces the 64 primary rays lane-parallel, phase 2 follows mirror/glass bounces, phase 3
// runs the wave-wide shadow test for each lit lane, phase 4 shades lane-parallel and sums the AA rays of
// a pixel in the reference's order.  The per-pixel xorshift streams (seeded by the GLOBAL pixel id,
// :319) are generated for 4 pixels at a time by 12 lanes into a per-wave LDS scratch.
//
// Arithmetic is the reference's, operation for operation (rt_math.h): results are bit-identical to the
// generic kernel and to the CPU oracle.
#include "rt_trace.h"

namespace uobrt {

namespace {

constexpr int kRngPixels = 4;               // pixels whose sample streams are generated together
constexpr int kRngStride = 64 * 4 + 4;      // 32-bit words per pixel in the scratch (+4: bank spread)

__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float shfl(float v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Per-lane registers of a "triangle lane" (lane i < ns holds shadow-casting triangle i)
struct TriLane {
  f3 v0, e1, e2, c;
  float c1;          // |c.x|+|c.y|+|c.z|
};

// Per-wave LDS.  Records r0..r2: the triangle lanes hand their per-surface-point terms to the 64 sample
// lanes; every sample lane reads the SAME record, which LDS serves as a broadcast into VGPRs.
// (Broadcasting through SGPRs instead — v_readlane_b32 — costs 4.3 issue cycles per value and makes
// every VALU instruction that consumes the SGPR half rate: profiles/r01_valu_issue_cost_8waves.txt.)
// Records h0,h1: the 64 surface points of the current task, written lane-parallel, read as broadcasts.
struct WaveLds {
  float4* r0;   // c.x c.y c.z | det(A0) = det(b,e1,e2)
  float4* r1;   // p.x p.y p.z | q.x        p = cof(b,e2), q = cof(e1,b)
  float2* r2;   // q.y q.z
  float4* h0;   // start.xyz | radius_sq
  float4* h1;   // dir.xyz   | hh (jitter half-width incl. rounding slack)
  uint32_t* rng;
};
constexpr int kWaveLdsBytes = 64 * (16 + 16 + 8) + 64 * 32 + kRngPixels * kRngStride * 4;

__device__ __forceinline__ WaveLds wave_lds(char* base) {
  WaveLds L;
  L.r0 = reinterpret_cast<float4*>(base);
  L.r1 = reinterpret_cast<float4*>(base + 64 * 16);
  L.r2 = reinterpret_cast<float2*>(base + 64 * 32);
  L.h0 = reinterpret_cast<float4*>(base + 64 * 40);
  L.h1 = reinterpret_cast<float4*>(base + 64 * 56);
  L.rng = reinterpret_cast<uint32_t*>(base + 64 * 72);
  return L;
}

// Number of the 64 jittered shadow rays (lane = sample) from surface point j of the current task
// towards the light that are NOT blocked: kernels.cl:243-311 evaluated for 64 samples at once.
//   dminlen, dk : lower bound of |d| over the jitter box and sqrt(radius_sq)*(1+slack), for the cull
//   sph_maybe   : false when no sample's ray can reach a shadow-casting sphere
template <bool CULL, bool COUNT>
__device__ __forceinline__ int wave_unshadowed(const FrameParams& P, const TriLane& T, const WaveLds& L, int lane, int ns,
                                               int j, float dminlen, float dk, bool sph_maybe, f3 jit, Work& wk) {
  const float4 h0 = L.h0[j], h1 = L.h1[j];                 // LDS broadcasts
  const f3 start = mk(h0.x, h0.y, h0.z), dir = mk(h1.x, h1.y, h1.z);
  const float radius_sq = h0.w;
  const f3 d = dir + jit;       // shadow_ray.direction + crush(rand_vec, light_spread), :333
  const f3 nd = -d;
  // ---- once per surface point, lane i = triangle i -------------------------------------------------
  unsigned long long cand;
  {
    const f3 b = start - T.v0;
    const f3 p = cof(b, T.e2);             // cofactors of det(A1) = det(-d, b, e2), :269
    const f3 q = cof(T.e1, b);             // cofactors of det(A2) = det(-d, e1, b), :270
    const float nA0 = detc(b, T.c);        // det(A0), :257-259
    reinterpret_cast<float*>(&L.r0[lane])[3] = nA0;
    L.r1[lane] = make_float4(p.x, p.y, p.z, q.x);
    L.r2[lane] = make_float2(q.y, q.z);
    bool keep = lane < ns;
    if (CULL) {
      // Interval bounds of the three determinants over the jitter box d = dir + [-h,h]^3.  hh >= h plus
      // every rounding error of the per-sample evaluation (see DESIGN.md "exact culling").
      const float hh = h1.w;
      const f3 md = -dir;
      const float D0 = detc(md, T.c), N1 = detc(md, p), N2 = detc(md, q);
      const float aD = fabsf(D0);
      const float Delta = hh * T.c1;
      const bool robust = aD > Delta + 1e-30f;           // every sample's det(A) has D0's sign and is normal
      const float sg = copysignf(1.0f, D0);
      const float tn = sg * nA0, un = sg * N1, vn = sg * N2;
      const float hp = hh * (fabsf(p.x) + fabsf(p.y) + fabsf(p.z));
      const float hq = hh * (fabsf(q.x) + fabsf(q.y) + fabsf(q.z));
      const float hiD = aD + Delta;
      const bool cS = tn < -1e-18f;                                  // t < 0 for every sample
      const bool cR = fabsf(nA0) * dminlen > hiD * dk;               // |t*d|^2 >= radius_sq for every sample
      const bool cU = un < -(hp + 1e-18f);                           // u < 0 for every sample
      const bool cV = vn < -(hq + 1e-18f);                           // v < 0 for every sample
      const bool cW = (un + vn) - (hp + hq) > hiD * 1.000004f;       // u+v > 1 wherever u,v >= 0
      keep = keep && !(robust && (cS || cR || cU || cV || cW));
    }
    cand = ballot(keep);
    if (COUNT) { wk.v[0] += 1; wk.v[4] += (unsigned)(ns - __popcll(cand)); }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- 64 samples against each candidate triangle ---------------------------------------------------
  // Lane predicates are kept as explicit 64-bit wave masks: each ballot below is ONE v_cmp writing an
  // SGPR pair, and all the and/or logic runs on the scalar unit.
  unsigned long long shadowed = 0ull;
  if (cand != 0ull) {
    int i = __builtin_ctzll(cand);
    cand &= cand - 1ull;
    float4 r0 = L.r0[i];
    for (;;) {
      const int inext = cand ? __builtin_ctzll(cand) : 0;
      const float4 nxt = L.r0[inext];          // prefetch the next candidate's record
      const float detA = detc(nd, mk(r0.x, r0.y, r0.z));
      float rr = rcp_newton(detA, 1);          // == 1.0f/detA, or NaN when detA is 0/denormal/inf (rt_math.h)
      float t = r0.w * rr;
      f3 dv = t * d;
      float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
      // first stage, :266.  The negated compares are also true for NaN, so a lane whose reciprocal needs
      // the division fallback reaches the second stage, where it is recomputed exactly.
      unsigned long long pass = ballot(!(t < 0.0f)) & ballot(!(dist >= radius_sq));
      if (COUNT) wk.v[1] += 1;
      if ((pass & ~shadowed) != 0ull) {                                // wave-uniform second stage
        if (COUNT) wk.v[2] += 1;
        if (ballot(rr != rr) != 0ull) {                                // rare: reciprocal outside v_rcp's range
          rr = 1.0f / detA;
          t = r0.w * rr;
          dv = t * d;
          dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
          pass = ballot(t >= 0) & ballot(dist < radius_sq);
        }
        const float4 r1 = L.r1[i];
        const float2 r2 = L.r2[i];
        const float u = detc(nd, mk(r1.x, r1.y, r1.z)) * rr;
        const float v = detc(nd, mk(r1.w, r2.x, r2.y)) * rr;
        shadowed |= pass & ballot(u >= 0) & ballot(v >= 0) & ballot((u + v) <= 1);   // :272
        if (shadowed == ~0ull) break;                                  // every sample blocked: any-hit early-out
      }
      if (cand == 0ull) break;
      cand &= cand - 1ull;
      i = inext;
      r0 = nxt;
    }
  }
  __builtin_amdgcn_wave_barrier();
  bool sh = (shadowed >> lane) & 1ull;
  if (sph_maybe && shadowed != ~0ull) {
    Work unused;
    if (COUNT) wk.v[3] += 1;
    if (!sh) sh = shadow_spheres<false>(P, start, d, radius_sq, unused);
  }
  return __popcll(ballot(!sh));
}

// Can any jittered ray from `start` towards `dir` (+- jitter of half-width hh per axis) touch a
// shadow-casting sphere?  Conservative: the line misses sphere (c,R) when |L x d| > R |d|; bound both
// sides over the jitter box and leave 0.2 % + rounding slack for the reference's discriminant (:285).
__device__ __forceinline__ bool spheres_maybe(const FrameParams& P, f3 start, f3 dir, float dlen, float hh) {
  bool maybe = false;
  const float jm = 1.7321f * hh;
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = P.sph[i];
    if (sp.col[3] == -1.0f) continue;                    // glass casts no shadow, :279
    const f3 Lv = start - mk(sp.cx, sp.cy, sp.cz);
    const f3 cr = mk(Lv.y * dir.z - Lv.z * dir.y, Lv.z * dir.x - Lv.x * dir.z, Lv.x * dir.y - Lv.y * dir.x);
    const float crn = sqrtf(dot3(cr, cr)), Ln = sqrtf(dot3(Lv, Lv)), R = sqrtf(fmaxf(sp.r2, 0.0f));
    const bool miss = (crn - Ln * jm > R * (dlen + jm) * 1.002f) && (Ln < 1000.0f * R) && (sp.r2 > 0.0f);
    maybe = maybe || !miss;
  }
  return maybe;
}

}  // namespace

// Grid: x = ceil(W/64) row segments, y = ceil(owned_rows/4); block = 256 threads = 4 waves = 4 rows.
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(256) void rt_draw_wave(const FrameParams P) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int n = P.n, ns = P.n_shadow;

  // ---- stage the triangle list once per workgroup ---------------------------------------------------
  stage_triangles(P, lds, tid, 256);
  int* sidx = reinterpret_cast<int*>(lds + kLdsRecords * n);          // shadow-casting triangles, in order
  if (wave == 0) {                                                    // n <= 64 on this path (supports())
    const bool casts = (lane < n) && (P.colors[lane < n ? lane : 0].w != -1.0f);   // glass casts no shadow, :247
    const unsigned long long m = ballot(casts);
    if (casts) sidx[__popcll(m & ((1ull << lane) - 1ull))] = lane;
  }
  __syncthreads();
  const WaveLds L = wave_lds(reinterpret_cast<char*>(sidx + ((n + 3) & ~3)) + wave * kWaveLdsBytes);

  const int lr = blockIdx.y * 4 + wave;
  if (lr >= P.owned_rows) return;                                     // whole wave; no block barrier follows
  const LdsScene S = lds_scene(lds, n);
  const int x0 = blockIdx.x * 64;
  const int y = band_global_row(lr, P.band_rows, P.band_index, P.band_count);
  const int aa = P.aa_x * P.aa_y;                                     // a power of two <= 64 (supports())
  const int la = __builtin_ctz(aa);
  const int PT = 64 >> la;                                            // pixels per task
  const int GP = PT < kRngPixels ? PT : kRngPixels;                   // pixels per RNG group
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const float hbox = P.spread / 2.f;                                  // |crush()| <= range/2, :51

  TriLane T;
  {
    const int ti = sidx[lane < ns ? lane : 0];
    T.v0 = xyz(S.v0[ti]); T.e1 = xyz(S.e1[ti]); T.e2 = xyz(S.e2[ti]); T.c = xyz(S.c[ti]);
    T.c1 = fabsf(T.c.x) + fabsf(T.c.y) + fabsf(T.c.z);
    L.r0[lane] = make_float4(T.c.x, T.c.y, T.c.z, 0.f);       // static part of record 0
  }

  f3 outc = mk(0.f, 0.f, 0.f);
  Work wk, xw;                               // xw: executed-work counters of this wave (COUNT builds only)
  if (COUNT) for (int q = 0; q < 8; ++q) xw.v[q] = 0;
  for (int k = 0; k < aa; ++k) {
    // ---- phase 1: 64 primary rays, lane = (pixel, AA sample) -----------------------------------------
    const int pj = k * PT + (lane >> la);       // pixel of this lane within the 64-pixel job
    const int a = lane & (aa - 1);              // AA sample index dy*rx+dx, kernels.cl:395
    const int x = x0 + pj;
    const bool valid = x < P.W;
    Ray ray = primary_ray(P, x, y, a % P.aa_x, a / P.aa_x);
    bool lit = false, secondary = false;
    if (valid) {
      closest_hit_primary<false>(S, P, ray, wk);
      if (ray.tri != -1) {
        // ---- phase 2: mirror / glass bounces (kernels.cl:342-365) -----------------------------------
        if (ray.col.w <= 0.0f) { secondary = true; lit = bounce_to_diffuse<false>(S, P, ray, wk); }
        else lit = true;
      }
    }
    // per-lane light set-up of direct_light, :323-326
    const f3 dir = light - ray.P;
    const f3 start = ray.P + 0.0001f * dir;
    const float radius_sq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    const float term = (16.0f * fmaxf(dot3(dir, ray.N), 0.0f)) / (4.0f * 3.14159274f * radius_sq);
    // bounds used by the cull (never by the shading): |dir|, jitter half-width with rounding slack
    const float dlen = sqrtf(radius_sq);
    const float hh = 1.002f * hbox + 2e-6f * (dlen + hbox);
    float dminlen = dlen - 1.7321f * hh;
    if (!(radius_sq > 1e-18f) || !(dminlen > 0.0f)) dminlen = 0.0f;   // disables the distance rule
    const float dk = dlen * 1.000004f;
    const unsigned long long sphmask = CULL ? ballot(lit && P.nsph > 0 && spheres_maybe(P, start, dir, dlen, hh))
                                            : (P.nsph > 0 ? ~0ull : 0ull);
    __builtin_amdgcn_wave_barrier();
    L.h0[lane] = make_float4(start.x, start.y, start.z, radius_sq);
    L.h1[lane] = make_float4(dir.x, dir.y, dir.z, hh);
    __builtin_amdgcn_wave_barrier();

    // ---- phase 3: wave-wide shadow test, one lit lane at a time ---------------------------------------
    int unshadowed = 0;
    const unsigned long long litmask = ballot(lit);
    const int GL = GP * aa;                     // lanes per RNG group
    for (int g = 0; g * GL < 64; ++g) {
      const unsigned long long gm = (litmask >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
      if (gm == 0ull) continue;
      // xorshift streams of the GP pixels of this group: lane c -> (pixel c/3, component c%3), :319,:331
      if (lane < 3 * GP) {
        const int pp = lane / 3, comp = lane % 3;
        const int gid = pixel_global_id(P, x0 + k * PT + g * GP + pp, y);
        const uint32_t seed = comp == 0 ? (uint32_t)gid : (uint32_t)((float)gid * (comp == 1 ? 91.0f : 19.0f));
        uint32_t s = xorshift(seed);
        uint32_t* dst = L.rng + pp * kRngStride + comp;
        for (int it = 0; it < 64; ++it) { s = xorshift(s); dst[it * 4] = s; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int pp = 0; pp < GP; ++pp) {
        unsigned long long pm = (gm >> (pp * aa)) & (aa == 64 ? ~0ull : ((1ull << aa) - 1ull));
        if (pm == 0ull) continue;
        const uint32_t* src = L.rng + pp * kRngStride + lane * 4;     // lane = sample index
        const f3 jit = mk(crush1(src[0], P.spread), crush1(src[1], P.spread), crush1(src[2], P.spread));
        const int base = g * GL + pp * aa;
        while (pm != 0ull) {
          const int j = base + __builtin_ctzll(pm);
          pm &= pm - 1ull;
          const int cnt = wave_unshadowed<CULL, COUNT>(P, T, L, lane, ns, j, rl(dminlen, j), rl(dk, j), (sphmask >> j) & 1ull, jit, xw);
          if (lane == j) unshadowed = cnt;
        }
      }
      __builtin_amdgcn_wave_barrier();          // scratch is rewritten by the next group
    }

    // ---- phase 4: shade lane-parallel (direct_light's sum :335, then :354 / :421-422) ----------------
    f3 contrib = mk(0.f, 0.f, 0.f);
    if (lit) {
      float total = 0.0f;
      if (unshadowed < 64) total += 0.0f * term;          // a blocked sample adds 0*term (NaN/inf-faithful)
      for (int i = 0; i < 64; ++i) if (i < unshadowed) total += term;
      const float l = 0.5f + total / 64.0f;
      if (secondary) { const float kk = 0.9f * l; contrib = mk(kk * ray.col.x, kk * ray.col.y, kk * ray.col.z); }
      else contrib = mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
    }
    // sum the AA rays of each pixel in index order (final_color_total +=, :415-425); a ray without a
    // contribution adds +0, which leaves the running sum unchanged bit for bit
    f3 acc = mk(0.f, 0.f, 0.f);
    const int first = (lane >> la) << la;
    for (int r = 0; r < aa; ++r) {
      acc = acc + mk(shfl(contrib.x, first + r), shfl(contrib.y, first + r), shfl(contrib.z, first + r));
    }
    // pixel pj's sum lives in lanes [pj*aa, pj*aa+aa); hand it to output lane (k*PT + pj)
    {
      const int rel = lane - k * PT;
      const int srcl = (rel >= 0 && rel < PT) ? (rel << la) : 0;
      const f3 v = mk(shfl(acc.x, srcl), shfl(acc.y, srcl), shfl(acc.z, srcl));
      if (rel >= 0 && rel < PT) outc = v;
    }
  }

  if (COUNT) {                               // wave-uniform counters: lane 0 publishes them
    if (lane == 0) for (int q = 0; q < 5; ++q) if (xw.v[q]) atomicAdd(&P.counters[q], xw.v[q]);
    return;
  }
  // ---- store: 64 consecutive pixels, one coalesced access per wave ------------------------------------
  const int x = x0 + lane;
  if (x < P.W) {
    const float inv = (float)aa;
    const f3 c = mk(outc.x / inv, outc.y / inv, outc.z / inv);
    const size_t o = (size_t)lr * P.W + x;
    P.out_argb[o] = pack_argb(c);
    if (P.out_rgb) P.out_rgb[o] = make_float4(c.x, c.y, c.z, 1.0f);
  }
}

template __global__ void rt_draw_wave<false, false>(const FrameParams);
template __global__ void rt_draw_wave<true, false>(const FrameParams);
template __global__ void rt_draw_wave<false, true>(const FrameParams);
template __global__ void rt_draw_wave<true, true>(const FrameParams);

bool wave_kernel_supports(const FrameParams& P) {
  const int aa = P.aa_x * P.aa_y;
  return P.S == 64 && aa >= 1 && aa <= 64 && (64 % aa) == 0 && P.n >= 1 && P.n <= 64 && P.n_shadow >= 1 &&
         P.spread >= 0.0f;
}

void launch_wave(const FrameParams& P, bool cull, bool count, hipStream_t stream) {
  const dim3 block(256);
  const dim3 grid((P.W + 63) / 64, (P.owned_rows + 3) / 4);
  const size_t lds_bytes = (size_t)P.n * kLdsRecords * sizeof(float4) + (size_t)((P.n + 3) & ~3) * sizeof(int) +
                           4 * (size_t)kWaveLdsBytes;
  if (count) {
    if (cull) hipLaunchKernelGGL((rt_draw_wave<true, true>), grid, block, lds_bytes, stream, P);
    else hipLaunchKernelGGL((rt_draw_wave<false, true>), grid, block, lds_bytes, stream, P);
  } else {
    if (cull) hipLaunchKernelGGL((rt_draw_wave<true, false>), grid, block, lds_bytes, stream, P);
    else hipLaunchKernelGGL((rt_draw_wave<false, false>), grid, block, lds_bytes, stream, P);
  }
}

}  // namespace uobrt
