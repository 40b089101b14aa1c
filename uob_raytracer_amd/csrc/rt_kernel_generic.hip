// rt_kernel_generic.hip — one-thread-per-pixel gfx950 ray tracer for ANY rt_config.
//
// This is the general path (any AA grid, any sample count, spheres, bounces, band partitions) and the
// instrumented work counter.  It keeps the reference's loop structure (kernel `draw`,
// Source/kernels.cl:368-428) so the early-exit semantics of in_shadow (:243-311) are reproduced
// literally, which the exact work counters need.  The headline configuration (64 shadow samples) runs
// on the wave-per-hit-point kernel in rt_kernel_wave.hip instead.
//
// Layout: a workgroup is 64x4 pixels (4 waves, one image row segment of 64 pixels per wave, so every
// framebuffer store is a fully coalesced 256-B / 1-KiB wave access).  The triangle list is staged ONCE
// per workgroup into LDS as float4 SoA records {v0, e1, e2, cof(e1,e2), normal, colour}: all lanes of a
// wave read the same record in the intersection loops, which LDS serves as a broadcast.
// Compiled with -ffp-contract=off: see rt_math.h for the numerics contract.
#include "rt_device.h"
#include "rt_math.h"

namespace uobrt {

namespace {

struct LdsScene {
  const float4 *v0, *e1, *e2, *c, *nrm, *col;
  int n;
};

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

#define AIRF 1.0f
#define GLASSF 1.52f

// kernels.cl:168-241 (single_ray_intersections) == :92-166 per ray
template <bool COUNT>
__device__ void closest_hit(const LdsScene& S, const FrameParams& P, Ray& ray, Work& wk) {
  float current_t = 3.402823466e+38f;
  const f3 nd = -ray.dir;
  for (int i = 0; i < S.n; ++i) {
    const f3 v0 = xyz(S.v0[i]), e1 = xyz(S.e1[i]), e2 = xyz(S.e2[i]), c = xyz(S.c[i]);
    const f3 b = ray.start - v0;
    const float detA_recip = 1.0f / detc(nd, c);
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
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = P.sph[i];
    const f3 ctr = mk(sp.cx, sp.cy, sp.cz);
    const f3 L = ray.start - ctr;
    const float a = dot3(ray.dir, ray.dir);
    const float b = 2 * dot3(ray.dir, L);
    const float cc = dot3(L, L) - sp.r2;
    const float disc = b * b - 4.0f * a * cc;
    if (COUNT) wk.v[W_CSPH]++;
    if (disc < 0.0f) continue;
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

// kernels.cl:243-311
template <bool COUNT>
__device__ bool in_shadow(const LdsScene& S, const FrameParams& P, f3 start, f3 dir, float radius_sq, Work& wk) {
  const f3 nd = -dir;
  for (int i = 0; i < S.n; ++i) {
    if (S.col[i].w == -1.0f) continue;
    const f3 v0 = xyz(S.v0[i]), c = xyz(S.c[i]);
    const f3 b = start - v0;
    const float detA_recip = 1.0f / detc(nd, c);
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
  for (int i = 0; i < P.nsph; ++i) {
    const DevSphere& sp = P.sph[i];
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

// kernels.cl:313-340 (the three channels of light_color are equal: one float carries the sum)
template <bool COUNT>
__device__ float direct_light(const LdsScene& S, const FrameParams& P, const Ray& ray, int global_id, Work& wk) {
  float total = 0.0f;
  uint32_t r0 = xorshift((uint32_t)global_id);
  uint32_t r1 = xorshift((uint32_t)((float)global_id * 91.0f));
  uint32_t r2 = xorshift((uint32_t)((float)global_id * 19.0f));
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const f3 dir = light - ray.P;
  const f3 start = ray.P + 0.0001f * dir;
  const float radius_sq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
  const float term = (16.0f * fmaxf(dot3(dir, ray.N), 0.0f)) / (4.0f * 3.14159274f * radius_sq);
  if (COUNT) wk.v[W_LIT]++;
  for (int i = 0; i < P.S; ++i) {
    r0 = xorshift(r0); r1 = xorshift(r1); r2 = xorshift(r2);
    const f3 jit = mk(crush1(r0, P.spread), crush1(r1, P.spread), crush1(r2, P.spread));
    const bool sh = in_shadow<COUNT>(S, P, start, dir + jit, radius_sq, wk);
    if (COUNT) wk.v[W_SHADOW]++;
    // mask*(light_color*max(dot,0)) / (4 pi r^2): 1.0f*x == x exactly; a shadowed sample adds +0
    total += sh ? 0.0f * term : term;
  }
  return total / (float)P.S;
}

// kernels.cl:54-65
__device__ Ray reflect_ray(const Ray& ray) {
  Ray o;
  o.tri = -1;
  o.col = make_float4(0.f, 0.f, 0.f, 1.0f);
  o.P = mk(0.f, 0.f, 0.f); o.N = mk(0.f, 0.f, 0.f);
  const float dn = dot3(ray.dir, ray.N);
  o.dir = ray.dir - 2.0f * (dn * ray.N);
  o.start = ray.P + 0.0001f * o.dir;
  o.medium = AIRF;
  o.dir = normalize3(o.dir);
  return o;
}

// kernels.cl:67-88 (total internal reflection is unreachable: sqrt of a negative is NaN, :77-80)
__device__ Ray refract_ray(const Ray& ray) {
  f3 normal = ray.N;
  const bool air = (ray.medium == AIRF);
  const float n1 = air ? AIRF : GLASSF, n2 = air ? GLASSF : AIRF;
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

// kernels.cl:342-365
template <bool COUNT>
__device__ f3 secondary_light(const LdsScene& S, const FrameParams& P, const Ray& ray, int global_id, Work& wk) {
  Ray p = ray;
  for (int b = 0; b < P.bounces && p.col.w <= 0.0f; ++b) {
    p = (p.col.w == 0.0f) ? reflect_ray(p) : refract_ray(p);
    if (COUNT) wk.v[W_BOUNCE]++;
    closest_hit<COUNT>(S, P, p, wk);
    if (p.tri != -1 && p.col.w > 0.0f) {
      const float l = 0.5f + direct_light<COUNT>(S, P, p, global_id, wk);
      const float k = 0.9f * l;
      return mk(k * p.col.x, k * p.col.y, k * p.col.z);
    }
  }
  return mk(0.f, 0.f, 0.f);
}

}  // namespace

// Stage the packed triangle list into the LDS records (one pass per workgroup).
__device__ __forceinline__ void stage_triangles(const FrameParams& P, float4* lds, int tid, int nthreads) {
  const int n = P.n;
  for (int i = tid; i < n; i += nthreads) {
    const float4 a = P.verts[3 * i], b = P.verts[3 * i + 1], c = P.verts[3 * i + 2];
    const f3 v0 = xyz(a), e1 = xyz(b) - v0, e2 = xyz(c) - v0;
    const f3 cf = cof(e1, e2);
    lds[i] = make_float4(v0.x, v0.y, v0.z, 0.f);
    lds[n + i] = make_float4(e1.x, e1.y, e1.z, 0.f);
    lds[2 * n + i] = make_float4(e2.x, e2.y, e2.z, 0.f);
    lds[3 * n + i] = make_float4(cf.x, cf.y, cf.z, 0.f);
    lds[4 * n + i] = P.normals[i];
    lds[5 * n + i] = P.colors[i];
  }
}

// kernels.cl:368-428.  Grid: x = ceil(W/64), y = ceil(owned_rows/4); block 64x4.
template <bool COUNT>
__global__ __launch_bounds__(256) void rt_draw_generic(const FrameParams P) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  stage_triangles(P, lds, tid, 256);
  __syncthreads();

  const int x = blockIdx.x * 64 + threadIdx.x;
  const int lr = blockIdx.y * 4 + threadIdx.y;
  Work wk;
  if (COUNT) for (int k = 0; k < 8; ++k) wk.v[k] = 0;

  if (x < P.W && lr < P.owned_rows) {
    const int n = P.n;
    LdsScene S{lds, lds + n, lds + 2 * n, lds + 3 * n, lds + 4 * n, lds + 5 * n, n};
    const int y = band_global_row(lr, P.band_rows, P.band_index, P.band_count);
    const float Wf = (float)P.W, Hf = (float)P.H;
    const int global_id = (int)((float)y * Wf + (float)x);
    const int rx = P.aa_x, ry = P.aa_y;
    const float bx = (float)(x * rx) - (Wf * (float)rx) / 2.0f;
    const float by = (float)(y * ry) - (Hf * (float)ry) / 2.0f;
    const f3 r0 = mk(P.rot[0], P.rot[1], P.rot[2]), r1 = mk(P.rot[4], P.rot[5], P.rot[6]),
             r2 = mk(P.rot[8], P.rot[9], P.rot[10]);
    f3 total = mk(0.f, 0.f, 0.f);
    for (int dy = 0; dy < ry; ++dy) {
      for (int dx = 0; dx < rx; ++dx) {
        Ray ray;
        ray.start = mk(P.cam[0], P.cam[1], P.cam[2]);
        const f3 d = mk(bx + (float)dx, (by + (float)dy) * P.sy, P.focal + 0.0f);
        ray.dir = normalize3(mk(dot3(r0, d), dot3(r1, d), dot3(r2, d)));
        ray.tri = -1; ray.medium = AIRF;
        ray.col = make_float4(0.f, 0.f, 0.f, 1.0f);
        ray.P = mk(0.f, 0.f, 0.f); ray.N = mk(0.f, 0.f, 0.f);
        if (COUNT) wk.v[W_PRIMARY]++;
        closest_hit<COUNT>(S, P, ray, wk);
        if (ray.tri != -1) {
          if (ray.col.w <= 0.0f) {
            total = total + secondary_light<COUNT>(S, P, ray, global_id, wk);
          } else {
            const float l = 0.5f + direct_light<COUNT>(S, P, ray, global_id, wk);
            total = total + mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
          }
        }
      }
    }
    const float inv = (float)(rx * ry);
    const f3 c = mk(total.x / inv, total.y / inv, total.z / inv);
    // color_pixel, kernels.cl:37-40
    const uint32_t R = (uint32_t)fminf(fmaxf(255 * c.x, 0.f), 255.f);
    const uint32_t G = (uint32_t)fminf(fmaxf(255 * c.y, 0.f), 255.f);
    const uint32_t B = (uint32_t)fminf(fmaxf(255 * c.z, 0.f), 255.f);
    const size_t o = (size_t)lr * P.W + x;
    if (!COUNT) {
      P.out_argb[o] = (255u << 24) + (R << 16) + (G << 8) + B;
      if (P.out_rgb) P.out_rgb[o] = make_float4(c.x, c.y, c.z, 1.0f);
    }
  }
  if (COUNT) {
    // one atomic per counter per wave
    for (int k = 0; k < 8; ++k) {
      unsigned long long v = wk.v[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0 && v) atomicAdd(&P.counters[k], v);
    }
  }
}

template __global__ void rt_draw_generic<false>(const FrameParams);
template __global__ void rt_draw_generic<true>(const FrameParams);

void launch_generic(const FrameParams& P, bool count, hipStream_t stream) {
  const dim3 block(64, 4);
  const dim3 grid((P.W + 63) / 64, (P.owned_rows + 3) / 4);
  const size_t lds_bytes = (size_t)P.n * 6 * sizeof(float4);
  if (count) hipLaunchKernelGGL(rt_draw_generic<true>, grid, block, lds_bytes, stream, P);
  else hipLaunchKernelGGL(rt_draw_generic<false>, grid, block, lds_bytes, stream, P);
}

}  // namespace uobrt
