// rt_kernel_wave.hip — the headline kernel: 64 shadow samples == one 64-lane wavefront.
//
// The reference spends ~98 % of its ray-primitive tests in direct_light's shadow loop
// (Source/kernels.cl:313-340 -> in_shadow :243-311): for ONE surface point, S jittered rays towards the
// area light are tested against every triangle.  On gfx950 a wavefront is 64 lanes, so for S = 64 this
// kernel maps   lane = shadow sample   and processes surface points one after another per wave:
//
//   * everything that depends only on (surface point, triangle) — b = start - v0, det(b,e1,e2),
//     cof(b,e2), cof(e1,b): 26 of the ~45 FP32 operations of one test — is computed ONCE per surface
//     point, lane-parallel over triangles (lane i = triangle i), and broadcast to the 64 sample lanes
//     through SGPRs (v_readlane_b32), instead of once per (sample, triangle) as the reference does;
//   * all 64 lanes test the same triangle against the same surface point, so the two-stage test of
//     in_shadow (t first, u/v only if t passes, :266) becomes a WAVE-UNIFORM branch: the u/v stage runs
//     only when some lane's t passes (s_cbranch on the ballot), with no divergence;
//   * any-hit early-out is a wave ballot: the triangle loop ends as soon as every lane is shadowed.
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
__device__ __forceinline__ f3 rl3(f3 v, int lane) { return mk(rl(v.x, lane), rl(v.y, lane), rl(v.z, lane)); }
__device__ __forceinline__ float shfl(float v, int lane) { return __shfl(v, lane, 64); }

// Per-lane registers of a "triangle lane" (lane i < ns holds shadow-casting triangle i)
struct TriLane {
  f3 v0, e1, e2, c;
};

// Per-wave LDS records through which the triangle lanes hand their per-surface-point terms to the 64
// sample lanes: every sample lane reads the SAME record, which LDS serves as a broadcast into VGPRs.
// (Broadcasting through SGPRs instead — v_readlane_b32 — costs 4.3 issue cycles per value and makes
// every VALU instruction that consumes the SGPR half rate: profiles/r01_valu_issue_cost_8waves.txt.)
struct WaveRecs {
  float4* r0;   // c.x c.y c.z | det(A0) = det(b,e1,e2)
  float4* r1;   // p.x p.y p.z | q.x        p = cof(b,e2), q = cof(e1,b)
  float2* r2;   // q.y q.z
};
constexpr int kWaveRecBytes = 64 * (16 + 16 + 8);
constexpr int kWaveLdsBytes = kWaveRecBytes + kRngPixels * kRngStride * 4;

__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Number of the 64 jittered shadow rays (lane = sample) from `start` towards the light that are NOT
// blocked: kernels.cl:243-311 evaluated for 64 samples at once.  start/dir/radius_sq are wave-uniform.
__device__ __forceinline__ int wave_unshadowed(const FrameParams& P, const TriLane& T, const WaveRecs& R, int lane,
                                               int ns, f3 start, f3 dir, float radius_sq, f3 jit) {
  const f3 d = dir + jit;       // shadow_ray.direction + crush(rand_vec, light_spread), :333
  const f3 nd = -d;
  // ---- once per surface point, lane i = triangle i -------------------------------------------------
  {
    const f3 b = start - T.v0;
    const f3 p = cof(b, T.e2);             // cofactors of det(A1) = det(-d, b, e2), :269
    const f3 q = cof(T.e1, b);             // cofactors of det(A2) = det(-d, e1, b), :270
    reinterpret_cast<float*>(&R.r0[lane])[3] = detc(b, T.c);     // det(A0), :257-259
    R.r1[lane] = make_float4(p.x, p.y, p.z, q.x);
    R.r2[lane] = make_float2(q.y, q.z);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- 64 samples against triangle i ----------------------------------------------------------------
  // Lane predicates are kept as explicit 64-bit wave masks: each ballot below is ONE v_cmp writing an
  // SGPR pair, and all the and/or logic runs on the scalar unit.
  unsigned long long shadowed = 0ull;
  float4 r0 = R.r0[0];
  for (int i = 0; i < ns; ++i) {
    const float4 nxt = R.r0[i + 1];          // prefetch the next record (slot ns <= 63 is valid memory)
    const float detA = detc(nd, mk(r0.x, r0.y, r0.z));
    float rr = rcp_newton(detA, 1);          // == 1.0f/detA, or NaN when detA is 0/denormal/inf (rt_math.h)
    float t = r0.w * rr;
    f3 dv = t * d;
    float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
    // first stage, :266.  The negated compares are also true for NaN, so a lane whose reciprocal needs
    // the division fallback reaches the second stage, where it is recomputed exactly.
    unsigned long long pass = ballot(!(t < 0.0f)) & ballot(!(dist >= radius_sq));
    if ((pass & ~shadowed) != 0ull) {                                // wave-uniform second stage
      if (ballot(rr != rr) != 0ull) {                                // rare: reciprocal outside v_rcp's range
        rr = 1.0f / detA;
        t = r0.w * rr;
        dv = t * d;
        dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
        pass = ballot(t >= 0) & ballot(dist < radius_sq);
      }
      const float4 r1 = R.r1[i];
      const float2 r2 = R.r2[i];
      const float u = detc(nd, mk(r1.x, r1.y, r1.z)) * rr;
      const float v = detc(nd, mk(r1.w, r2.x, r2.y)) * rr;
      shadowed |= pass & ballot(u >= 0) & ballot(v >= 0) & ballot((u + v) <= 1);   // :272
      if (shadowed == ~0ull) break;                                  // every sample blocked: any-hit early-out
    }
    r0 = nxt;
  }
  __builtin_amdgcn_wave_barrier();
  bool sh = (shadowed >> lane) & 1ull;
  if (P.nsph > 0 && shadowed != ~0ull) {
    Work wk;
    if (!sh) sh = shadow_spheres<false>(P, start, d, radius_sq, wk);
  }
  return __popcll(ballot(!sh));
}

}  // namespace

// Grid: x = ceil(W/64) row segments, y = ceil(owned_rows/4); block = 256 threads = 4 waves = 4 rows.
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
    const unsigned long long m = __ballot(casts);
    if (casts) sidx[__popcll(m & ((1ull << lane) - 1ull))] = lane;
  }
  __syncthreads();
  char* wave_lds = reinterpret_cast<char*>(sidx + ((n + 3) & ~3)) + wave * kWaveLdsBytes;
  const WaveRecs R{reinterpret_cast<float4*>(wave_lds), reinterpret_cast<float4*>(wave_lds + 64 * 16),
                   reinterpret_cast<float2*>(wave_lds + 64 * 32)};
  uint32_t* rng = reinterpret_cast<uint32_t*>(wave_lds + kWaveRecBytes);

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

  TriLane T;
  {
    const int ti = sidx[lane < ns ? lane : 0];
    T.v0 = xyz(S.v0[ti]); T.e1 = xyz(S.e1[ti]); T.e2 = xyz(S.e2[ti]); T.c = xyz(S.c[ti]);
    R.r0[lane] = make_float4(T.c.x, T.c.y, T.c.z, 0.f);       // static part of record 0
  }

  f3 outc = mk(0.f, 0.f, 0.f);
  Work wk;
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
        uint32_t* dst = rng + pp * kRngStride + comp;
        for (int it = 0; it < 64; ++it) { s = xorshift(s); dst[it * 4] = s; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int pp = 0; pp < GP; ++pp) {
        unsigned long long pm = (gm >> (pp * aa)) & (aa == 64 ? ~0ull : ((1ull << aa) - 1ull));
        if (pm == 0ull) continue;
        const uint32_t* src = rng + pp * kRngStride + lane * 4;       // lane = sample index
        const f3 jit = mk(crush1(src[0], P.spread), crush1(src[1], P.spread), crush1(src[2], P.spread));
        const int base = g * GL + pp * aa;
        while (pm != 0ull) {
          const int j = base + __builtin_ctzll(pm);
          pm &= pm - 1ull;
          const int cnt = wave_unshadowed(P, T, R, lane, ns, rl3(start, j), rl3(dir, j), rl(radius_sq, j), jit);
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

bool wave_kernel_supports(const FrameParams& P) {
  const int aa = P.aa_x * P.aa_y;
  return P.S == 64 && aa >= 1 && aa <= 64 && (64 % aa) == 0 && P.n >= 1 && P.n <= 64 && P.n_shadow >= 1;
}

void launch_wave(const FrameParams& P, hipStream_t stream) {
  const dim3 block(256);
  const dim3 grid((P.W + 63) / 64, (P.owned_rows + 3) / 4);
  const size_t lds_bytes = (size_t)P.n * kLdsRecords * sizeof(float4) + (size_t)((P.n + 3) & ~3) * sizeof(int) +
                           4 * (size_t)kWaveLdsBytes;
  hipLaunchKernelGGL(rt_draw_wave, grid, block, lds_bytes, stream, P);
}

}  // namespace uobrt
