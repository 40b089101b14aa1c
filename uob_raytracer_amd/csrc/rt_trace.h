// rt_trace.h — device functions shared by the gfx950 kernels: LDS scene records, closest hit,
// any-hit shadow test, reflect/refract, in the reference's operation order (Source/kernels.cl).
// Numerics contract: rt_math.h.
#pragma once
#include "rt_device.h"
#include "rt_math.h"

namespace uobrt {

// LDS scene: float4 SoA records, one pass of staging per workgroup (stage_triangles).
//   v0.xyz , w = colour.w | e1.xyz , w = original index (bits) | e2.xyz | c = cof(e1,e2).xyz , w = det(cam-v0, e1, e2)
//   normal | colour (w = material) | pc = cof(cam-v0, e2) | qc = cof(e1, cam-v0)
// The last two (and c.w) are the camera-dependent terms of the PRIMARY-ray test: every primary ray
// starts at the camera, so b = cam - v0 is the same for all of them (kernels.cl:106 with start = cam).
struct LdsScene {
  const float4 *v0, *e1, *e2, *c, *nrm, *col, *pc, *qc;
  int n;
};
constexpr int kLdsRecords = 8;

// `stride` = distance between the record arrays in float4 (n when packed; a compile-time constant >= n lets
// the compiler fold every array base into the ds_read offset field)
__device__ __forceinline__ LdsScene lds_scene(const float4* lds, int n, int stride) {
  const int s = stride;
  return LdsScene{lds, lds + s, lds + 2 * s, lds + 3 * s, lds + 4 * s, lds + 5 * s, lds + 6 * s, lds + 7 * s, n};
}
__device__ __forceinline__ LdsScene lds_scene(const float4* lds, int n) { return lds_scene(lds, n, n); }

// Triangles that fit one LDS stage (8 records of 16 B each, 64 KB)
constexpr int kLdsMaxTriangles = 512;

__device__ __forceinline__ void stage_triangles(const FrameParams& P, float4* lds, long tid, long nthreads, int stride = 0) {
  const int n = P.n;
  const int s = stride ? stride : n;
  const f3 cam = mk(P.cam[0], P.cam[1], P.cam[2]);
  for (long i = tid; i < n; i += nthreads) {
    const float4 a = P.verts[3 * i], b = P.verts[3 * i + 1], c = P.verts[3 * i + 2];
    const f3 v0 = xyz(a), e1 = xyz(b) - v0, e2 = xyz(c) - v0;
    const f3 cf = cof(e1, e2);
    const f3 bc = cam - v0;
    const f3 pc = cof(bc, e2), qc = cof(e1, bc);
    lds[i] = make_float4(v0.x, v0.y, v0.z, P.colors[i].w);      // w: material flag (-1 = glass casts no shadow)
    lds[s + i] = make_float4(e1.x, e1.y, e1.z, __int_as_float(P.orig ? P.orig[i] : (int)i));   // w: original index
    lds[2 * s + i] = make_float4(e2.x, e2.y, e2.z, 0.f);
    lds[3 * s + i] = make_float4(cf.x, cf.y, cf.z, detc(bc, cf));
    lds[4 * s + i] = P.normals[i];
    lds[5 * s + i] = P.colors[i];
    lds[6 * s + i] = make_float4(pc.x, pc.y, pc.z, 0.f);
    lds[7 * s + i] = make_float4(qc.x, qc.y, qc.z, 0.f);
  }
}

// The sphere table: kernel argument (scalar registers) by default; a translation unit that defines RT_SPHERES_IN_LDS
// reads it from LDS instead, where its kernels stage it once per workgroup (stage_spheres) — the culls leave few rays
// that look at a sphere, and the scalar registers are the scarcer resource of the wave-mapped kernels.
#ifdef RT_SPHERES_IN_LDS
__shared__ DevSphere g_sph[RT_MAX_SPHERES];
#define RT_SPH(P, i) g_sph[i]
__device__ __forceinline__ void stage_spheres(const FrameParams& P, int tid) {
  if (tid < P.nsph * 8) reinterpret_cast<float*>(g_sph)[tid] = reinterpret_cast<const float*>(P.sph_dev)[tid];
}
#else
#define RT_SPH(P, i) P.sph[i]
#endif

struct Ray {           // kernels.cl:21-29
  f3 start, dir, P, N;
  float4 col;
  float medium;
  int tri;             // -1 none, -2 sphere, >= 0 triangle
};

struct Work {
  unsigned long long v[8];
};
enum { W_PRIMARY, W_BOUNCE, W_SHADOW, W_CTRI, W_CSPH, W_STRI, W_SSPH, W_LIT };

#define RT_AIR 1.0f
#define RT_GLASS 1.52f
#define RT_MAXFLOAT 3.402823466e+38f

// Sphere part of the closest-hit search, kernels.cl:208-239 (== :132-163)
template <bool COUNT>
__device__ __forceinline__ void closest_spheres(const FrameParams& P, Ray& ray, float& current_t, Work& wk) {
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = RT_SPH(P, i);
    const f3 ctr = mk(sp.cx, sp.cy, sp.cz);
    const f3 L = ray.start - ctr;
    const float a = dot3(ray.dir, ray.dir);
    const float b = 2 * dot3(ray.dir, L);
    const float cc = dot3(L, L) - sp.r2;
    const float disc = b * b - 4.0f * a * cc;
    if (COUNT) wk.v[W_CSPH]++;
    if (disc < 0.0f) continue;
    // the reference's -0.5 literal is double; x0.5 is exact, so the float product is identical
    const float q = (b > 0) ? -0.5f * (b + sqrtf(disc)) : -0.5f * (b - sqrtf(disc));
    const float x0 = q / a, x1 = cc / q;
    const float x_min = fminf(x0, x1), x_max = fmaxf(x0, x1);
    float x;
    if (x_min >= 0.0f && x_min < current_t) x = x_min;
    else if (x_max >= 0.0f && x_max < current_t) x = x_max;
    else continue;
    ray.tri = -2;
    ray.P = ray.start + x * ray.dir;
    ray.N = normalize3(ray.P - ctr);
    ray.col = make_float4(sp.col[0], sp.col[1], sp.col[2], sp.col[3]);
    current_t = x;
  }
}

// Closest hit of an arbitrary ray: kernels.cl:168-241 (single_ray_intersections)
template <bool COUNT>
__device__ void closest_hit(const LdsScene& S, const FrameParams& P, Ray& ray, Work& wk) {
  float current_t = RT_MAXFLOAT;
  const f3 nd = -ray.dir;
  for (int i = 0; i < S.n; ++i) {
    const f3 v0 = xyz(S.v0[i]), e1 = xyz(S.e1[i]), e2 = xyz(S.e2[i]), c = xyz(S.c[i]);
    const f3 b = ray.start - v0;
    const float detA_recip = rcp_exact(detc(nd, c));
    const float t = detc(b, c) * detA_recip;
    const float u = detc(nd, cof(b, e2)) * detA_recip;
    const float v = detc(nd, cof(e1, b)) * detA_recip;
    if (COUNT) wk.v[W_CTRI]++;
    if (t < current_t && u >= 0 && v >= 0 && (u + v) <= 1 && t >= 0) {
      ray.tri = i;
      ray.P = (v0 + u * e1) + v * e2;
      ray.N = xyz(S.nrm[i]);
      ray.col = S.col[i];
      current_t = t;
    }
  }
  closest_spheres<COUNT>(P, ray, current_t, wk);
}

// Closest hit of a PRIMARY ray (start == camera): same arithmetic as closest_hit, with the
// camera-dependent determinant terms read from the staged records instead of recomputed per ray.
template <bool COUNT>
__device__ void closest_hit_primary(const LdsScene& S, const FrameParams& P, Ray& ray, Work& wk) {
  float current_t = RT_MAXFLOAT;
  const f3 nd = -ray.dir;
  float bu = 0.f, bv = 0.f;
  int best = -1;
  for (int i = 0; i < S.n; ++i) {
    const float4 c4 = S.c[i];
    const float detA_recip = rcp_exact(detc(nd, xyz(c4)));
    const float t = c4.w * detA_recip;
    const float u = detc(nd, xyz(S.pc[i])) * detA_recip;
    const float v = detc(nd, xyz(S.qc[i])) * detA_recip;
    if (COUNT) wk.v[W_CTRI]++;
    if (t < current_t && u >= 0 && v >= 0 && (u + v) <= 1 && t >= 0) {
      best = i; bu = u; bv = v; current_t = t;
    }
  }
  if (best >= 0) {
    ray.tri = best;
    ray.P = (xyz(S.v0[best]) + bu * xyz(S.e1[best])) + bv * xyz(S.e2[best]);
    ray.N = xyz(S.nrm[best]);
    ray.col = S.col[best];
  }
  closest_spheres<COUNT>(P, ray, current_t, wk);
}

// closest_hit_primary restricted to the triangles whose bit is set in `mask` (a wave-uniform 64-bit set,
// n <= 64), visited in increasing index order so that ties on t resolve exactly as in the full loop
// (strict '<', kernels.cl:120).  The caller guarantees that no lane's ray can hit a triangle outside it.
// `spheres` (wave-uniform): false when the caller has shown that no ray of the wave can touch any sphere.
__device__ inline void closest_hit_primary_masked(const LdsScene& S, const FrameParams& P, Ray& ray,
                                                  unsigned long long mask, bool spheres = true) {
  float current_t = RT_MAXFLOAT;
  const f3 nd = -ray.dir;
  float bu = 0.f, bv = 0.f;
  int best = -1;
  for (; mask != 0ull; mask &= mask - 1ull) {
    const int i = __builtin_ctzll(mask);
    const float4 c4 = S.c[i];
    const f3 pc = xyz(S.pc[i]), qc = xyz(S.qc[i]);     // all three records requested before the first is used
    const float detA_recip = rcp_exact(detc(nd, xyz(c4)));
    const float t = c4.w * detA_recip;
    const float u = detc(nd, pc) * detA_recip;
    const float v = detc(nd, qc) * detA_recip;
    // (& instead of &&: five compares and four selects, no branch around them)
    const bool hit = (t < current_t) & (u >= 0) & (v >= 0) & ((u + v) <= 1) & (t >= 0);
    best = hit ? i : best; bu = hit ? u : bu; bv = hit ? v : bv; current_t = hit ? t : current_t;
  }
  if (best >= 0) {
    ray.tri = best;
    ray.P = (xyz(S.v0[best]) + bu * xyz(S.e1[best])) + bv * xyz(S.e2[best]);
    ray.N = xyz(S.nrm[best]);
    ray.col = S.col[best];
  }
  Work wk;
  if (spheres) closest_spheres<false>(P, ray, current_t, wk);
}

// closest_hit restricted to the triangles in `mask`, visited in index order (ties on t resolve as in the full loop)
__device__ inline void closest_hit_masked(const LdsScene& S, const FrameParams& P, Ray& ray, unsigned long long mask) {
  float current_t = RT_MAXFLOAT;
  const f3 nd = -ray.dir;
  float bu = 0.f, bv = 0.f;
  int best = -1;
  for (; mask != 0ull; mask &= mask - 1ull) {
    const int i = __builtin_ctzll(mask);
    const f3 v0 = xyz(S.v0[i]), e1 = xyz(S.e1[i]), e2 = xyz(S.e2[i]), c = xyz(S.c[i]);
    const f3 b = ray.start - v0;
    const float detA_recip = rcp_exact(detc(nd, c));
    const float t = detc(b, c) * detA_recip;
    const float u = detc(nd, cof(b, e2)) * detA_recip;
    const float v = detc(nd, cof(e1, b)) * detA_recip;
    if (t < current_t && u >= 0 && v >= 0 && (u + v) <= 1 && t >= 0) {
      best = i; bu = u; bv = v; current_t = t;
    }
  }
  if (best >= 0) {
    ray.tri = best;
    ray.P = (xyz(S.v0[best]) + bu * xyz(S.e1[best])) + bv * xyz(S.e2[best]);
    ray.N = xyz(S.nrm[best]);
    ray.col = S.col[best];
  }
  Work wk;
  closest_spheres<false>(P, ray, current_t, wk);
}

// Sphere part of the shadow test, kernels.cl:278-307
template <bool COUNT>
__device__ __forceinline__ bool shadow_spheres(const FrameParams& P, f3 start, f3 dir, float radius_sq, Work& wk) {
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = RT_SPH(P, i);
    if (sp.col[3] == -1.0f) continue;
    const f3 L = start - mk(sp.cx, sp.cy, sp.cz);
    const float a = dot3(dir, dir);
    const float b = 2 * dot3(dir, L);
    const float cc = dot3(L, L) - sp.r2;
    const float disc = b * b - 4.0f * a * cc;
    if (COUNT) wk.v[W_SSPH]++;
    if (disc < 0.0f) continue;
    const float q = (b > 0) ? -0.5f * (b + sqrtf(disc)) : -0.5f * (b - sqrtf(disc));
    const float x0 = q / a, x1 = cc / q;
    const float x_min = fminf(x0, x1), x_max = fmaxf(x0, x1);
    const f3 dmin = x_min * dir, dmax = x_max * dir;
    const float min_dist = dot3(dmin, dmin), max_dist = dot3(dmax, dmax);
    if (x_min >= 0.0f && min_dist < radius_sq) return true;
    else if (x_max >= 0.0f && max_dist < radius_sq) return true;
  }
  return false;
}

// kernels.cl:243-311, literal loop order (used by the generic kernel and the work counters)
template <bool COUNT>
__device__ bool in_shadow(const LdsScene& S, const FrameParams& P, f3 start, f3 dir, float radius_sq, Work& wk) {
  const f3 nd = -dir;
  for (int i = 0; i < S.n; ++i) {
    if (S.col[i].w == -1.0f) continue;
    const f3 v0 = xyz(S.v0[i]), c = xyz(S.c[i]);
    const f3 b = start - v0;
    const float detA_recip = rcp_exact(detc(nd, c));
    const float t = detc(b, c) * detA_recip;
    const f3 dv = t * dir;
    const float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
    if (COUNT) wk.v[W_STRI]++;
    if (t >= 0 && dist < radius_sq) {
      const f3 e1 = xyz(S.e1[i]), e2 = xyz(S.e2[i]);
      const float u = detc(nd, cof(b, e2)) * detA_recip;
      const float v = detc(nd, cof(e1, b)) * detA_recip;
      if (u >= 0 && v >= 0 && (u + v) <= 1) return true;
    }
  }
  return shadow_spheres<COUNT>(P, start, dir, radius_sq, wk);
}

// kernels.cl:54-65
__device__ inline Ray reflect_ray(const Ray& ray) {
  Ray o;
  o.tri = -1;
  o.col = make_float4(0.f, 0.f, 0.f, 1.0f);
  o.P = mk(0.f, 0.f, 0.f); o.N = mk(0.f, 0.f, 0.f);
  const float dn = dot3(ray.dir, ray.N);
  o.dir = ray.dir - 2.0f * (dn * ray.N);
  o.start = ray.P + 0.0001f * o.dir;
  o.medium = RT_AIR;
  o.dir = normalize3(o.dir);
  return o;
}

// kernels.cl:67-88 (total internal reflection is unreachable: sqrt of a negative is NaN, :77-80)
__device__ inline Ray refract_ray(const Ray& ray) {
  f3 normal = ray.N;
  const bool air = (ray.medium == RT_AIR);
  const float n1 = air ? RT_AIR : RT_GLASS, n2 = air ? RT_GLASS : RT_AIR;
  float c1 = dot3(normal, ray.dir);
  if (c1 < 0.0f) normal = -1.0f * normal;
  c1 = fabsf(c1);
  const float n = n1 / n2;
  const float c2 = sqrtf(1 - (n * n) * (1 - (c1 * c1)));
  if (c2 < 0.0f) return reflect_ray(ray);
  Ray o;
  o.tri = -1;
  o.col = make_float4(1.0f, 0.f, 0.f, 1.0f);
  o.P = mk(0.f, 0.f, 0.f); o.N = mk(0.f, 0.f, 0.f);
  o.dir = n * ray.dir + (n * c1 - c2) * (-normal);
  o.start = ray.P + 0.0001f * o.dir;
  o.medium = n2;
  o.dir = normalize3(o.dir);
  return o;
}

// The bounce loop of secondary_light (kernels.cl:342-365) up to, not including, the lighting of the
// first diffuse surface found: returns true and leaves that hit in `p`.
template <bool COUNT>
__device__ bool bounce_to_diffuse(const LdsScene& S, const FrameParams& P, Ray& p, Work& wk) {
  for (int b = 0; b < P.bounces && p.col.w <= 0.0f; ++b) {
    p = (p.col.w == 0.0f) ? reflect_ray(p) : refract_ray(p);
    if (COUNT) wk.v[W_BOUNCE]++;
    closest_hit<COUNT>(S, P, p, wk);
    if (p.tri != -1 && p.col.w > 0.0f) return true;
  }
  return false;
}

// Primary ray through AA sample (dx,dy) of pixel (x,y): kernels.cl:384-407.  Units are AA sub-pixels
// along x; sy = aa_x/aa_y rescales the y pitch for non-square grids (1 for the reference's square ones).
// aa_x, aa_y, sy: P.aa_x, P.aa_y, P.sy — passed separately so that a kernel specialised on the grid hands in constants
__device__ __forceinline__ Ray primary_ray(const FrameParams& P, int x, int y, int dx, int dy, int aa_x, int aa_y, float sy) {
  // (Wf*rx)/2, (Hf*ry)/2, focal + 0 and r_k.z * d.z are frame invariants, evaluated on the host with the same operations
  const float bx = (float)(x * aa_x) - P.half_wx;
  const float by = (float)(y * aa_y) - P.half_hy;
  Ray ray;
  ray.start = mk(P.cam[0], P.cam[1], P.cam[2]);
  const float dxs = bx + (float)dx, dys = (by + (float)dy) * sy;
  // dot(r_k, d) = r_k.x*d.x + r_k.y*d.y + r_k.z*d.z, left to right
  ray.dir = normalize3(mk(P.rot[0] * dxs + P.rot[1] * dys + P.rzf[0], P.rot[4] * dxs + P.rot[5] * dys + P.rzf[1],
                          P.rot[8] * dxs + P.rot[9] * dys + P.rzf[2]));
  ray.tri = -1;
  ray.medium = RT_AIR;
  ray.col = make_float4(0.f, 0.f, 0.f, 1.0f);
  ray.P = mk(0.f, 0.f, 0.f);
  ray.N = mk(0.f, 0.f, 0.f);
  return ray;
}

__device__ __forceinline__ Ray primary_ray(const FrameParams& P, int x, int y, int dx, int dy) {
  return primary_ray(P, x, y, dx, dy, P.aa_x, P.aa_y, P.sy);
}

// color_pixel, kernels.cl:37-40
__device__ __forceinline__ uint32_t pack_argb(f3 c) {
  const uint32_t R = (uint32_t)fminf(fmaxf(255 * c.x, 0.f), 255.f);
  const uint32_t G = (uint32_t)fminf(fmaxf(255 * c.y, 0.f), 255.f);
  const uint32_t B = (uint32_t)fminf(fmaxf(255 * c.z, 0.f), 255.f);
  return (255u << 24) + (R << 16) + (G << 8) + B;
}

// kernels.cl:380 — the pixel id is formed in FP32
__device__ __forceinline__ int pixel_global_id(const FrameParams& P, int x, int y) {
  return (int)((float)y * P.w_f + (float)x);
}

}  // namespace uobrt
