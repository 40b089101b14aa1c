/* rt_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's per-pixel ray tracer.
 *
 * This is the checker for the HIP path: a plain-C restatement of kernel `draw` and everything it
 * calls (/root/reference/Source/kernels.cl:31-428), with every hard-coded constant of the reference
 * turned into an rt_config field.  Each function cites the reference lines it follows and keeps the
 * reference's floating-point operation ORDER; build with -ffp-contract=off (no FMA), IEEE division and
 * square root (the "strict" semantics of oracle/ref_shim.cpp), so that on settings the reference can
 * express it reproduces the compiled reference kernel (oracle/_ref) bit for bit — that is how it is
 * pinned (tests/test_oracle_vs_ref.py, fixtures in tests/golden/).  It additionally covers what the
 * reference cannot express: non-square AA grids, band-partitioned frames, arbitrary sphere tables.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the product
 * library (libuob_rt.so) never links or loads it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/uob_rt.h"

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* OpenCL dot as oracle/ref_shim.cpp defines it: x*x + y*y + z*z, left to right */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* OpenCL normalize as the shim defines it: v / sqrtf(dot(v,v)) */
static inline v3 normalize(v3 a) {
  const float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  return V(a.x / len, a.y / len, a.z / len);
}

/* kernels.cl:31-35 — cofactor expansion on row 0 */
static inline float det3(v3 m0, v3 m1, v3 m2) {
  return m0.x * (m1.y * m2.z - m1.z * m2.y) - m0.y * (m1.x * m2.z - m1.z * m2.x) + m0.z * (m1.x * m2.y - m1.y * m2.x);
}

/* kernels.cl:21-29 */
typedef struct {
  v3 start, direction, intersect, normal;
  float color[4];
  float medium;
  int tri;   /* -1 none, -2 sphere, >=0 triangle index */
} Ray;

typedef struct {
  const rt_config* cfg;
  const float* verts;    /* float4[3n] */
  const float* normals;  /* float4[n]  */
  const float* colors;   /* float4[n]  */
  int n;
  v3 light;
  rt_work* work;         /* nullable per-thread counters */
} Scene;

#define AIRF 1.0f
#define GLASSF 1.52f

static inline v3 vert(const Scene* s, int k) { return V(s->verts[4 * k], s->verts[4 * k + 1], s->verts[4 * k + 2]); }

/* kernels.cl:42-47 — xorshift32, component-wise */
static inline void rng_next(uint32_t r[3]) {
  for (int k = 0; k < 3; ++k) { uint32_t s = r[k]; s ^= s << 13; s ^= s >> 17; s ^= s << 5; r[k] = s; }
}
/* kernels.cl:49-52 — (float)UINT_MAX rounds to 2^32 */
static inline v3 crush(const uint32_t r[3], float range) {
  const float d = 4294967296.0f, h = range / 2.f;
  return V(range * (float)r[0] / d - h, range * (float)r[1] / d - h, range * (float)r[2] / d - h);
}

/* kernels.cl:168-241 (single_ray_intersections) == :92-166 per ray (batch_ray_intersections) */
static void closest_hit(const Scene* s, Ray* ray) {
  float current_t = 3.402823466e+38f; /* MAXFLOAT */
  const v3 nd = neg(ray->direction);
  for (int i = 0; i < s->n; ++i) {
    const v3 v0 = vert(s, 3 * i);
    const v3 e1 = sub(vert(s, 3 * i + 1), v0);
    const v3 e2 = sub(vert(s, 3 * i + 2), v0);
    const v3 b = sub(ray->start, v0);
    const float detA_recip = 1.0f / det3(nd, e1, e2);        /* native_recip, :186 */
    const float t = det3(b, e1, e2) * detA_recip;
    const float u = det3(nd, b, e2) * detA_recip;
    const float v = det3(nd, e1, b) * detA_recip;
    if (s->work) s->work->closest_tri_tests++;
    if (t < current_t && u >= 0 && v >= 0 && (u + v) <= 1 && t >= 0) {   /* :194 */
      ray->tri = i;
      ray->intersect = add(add(v0, scale(u, e1)), scale(v, e2));
      ray->normal = V(s->normals[4 * i], s->normals[4 * i + 1], s->normals[4 * i + 2]);
      memcpy(ray->color, s->colors + 4 * i, 16);
      current_t = t;
    }
  }
  for (int i = 0; i < s->cfg->num_spheres; ++i) {            /* :208-239 */
    const rt_sphere* sp = &s->cfg->spheres[i];
    const v3 c = V(sp->center[0], sp->center[1], sp->center[2]);
    const v3 L = sub(ray->start, c);
    const float a = dot(ray->direction, ray->direction);
    const float b = 2 * dot(ray->direction, L);
    const float cc = dot(L, L) - sp->radius_sq;
    const float disc = b * b - 4.0f * a * cc;
    if (s->work) s->work->closest_sphere_tests++;
    if (disc < 0.0f) continue;
    /* the -0.5 literal is double in the reference; x0.5 is exact so the float product is identical */
    const float q = (b > 0) ? -0.5f * (b + sqrtf(disc)) : -0.5f * (b - sqrtf(disc));
    const float x0 = q / a, x1 = cc / q;
    const float x_min = fminf(x0, x1), x_max = fmaxf(x0, x1);
    float x;
    if (x_min >= 0.0f && x_min < current_t) x = x_min;
    else if (x_max >= 0.0f && x_max < current_t) x = x_max;
    else continue;
    ray->tri = -2;
    ray->intersect = add(ray->start, scale(x, ray->direction));   /* direction*x : commutative */
    ray->normal = normalize(sub(ray->intersect, c));
    memcpy(ray->color, sp->color, 16);
    current_t = x;
  }
}

/* kernels.cl:243-311 */
static int in_shadow(const Scene* s, v3 start, v3 dir, float radius_sq) {
  const v3 nd = neg(dir);
  for (int i = 0; i < s->n; ++i) {
    if (s->colors[4 * i + 3] == -1.0f) continue;            /* :247 no glass shadows */
    const v3 v0 = vert(s, 3 * i);
    const v3 e1 = sub(vert(s, 3 * i + 1), v0);
    const v3 e2 = sub(vert(s, 3 * i + 2), v0);
    const v3 b = sub(start, v0);
    const float detA_recip = 1.0f / det3(nd, e1, e2);
    const float t = det3(b, e1, e2) * detA_recip;
    const v3 dv = scale(t, dir);
    const float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
    if (s->work) s->work->shadow_tri_tests++;
    if (t >= 0 && dist < radius_sq) {                        /* :266 */
      const float u = det3(nd, b, e2) * detA_recip;
      const float v = det3(nd, e1, b) * detA_recip;
      if (u >= 0 && v >= 0 && (u + v) <= 1) return 1;        /* :272 */
    }
  }
  for (int i = 0; i < s->cfg->num_spheres; ++i) {            /* :278-307 */
    const rt_sphere* sp = &s->cfg->spheres[i];
    if (sp->color[3] == -1.0f) continue;
    const v3 L = sub(start, V(sp->center[0], sp->center[1], sp->center[2]));
    const float a = dot(dir, dir);
    const float b = 2 * dot(dir, L);
    const float cc = dot(L, L) - sp->radius_sq;
    const float disc = b * b - 4.0f * a * cc;
    if (s->work) s->work->shadow_sphere_tests++;
    if (disc < 0.0f) continue;
    const float q = (b > 0) ? -0.5f * (b + sqrtf(disc)) : -0.5f * (b - sqrtf(disc));
    const float x0 = q / a, x1 = cc / q;
    const float x_min = fminf(x0, x1), x_max = fmaxf(x0, x1);
    const v3 dmin = scale(x_min, dir), dmax = scale(x_max, dir);
    const float min_dist = dot(dmin, dmin), max_dist = dot(dmax, dmax);
    if (x_min >= 0.0f && min_dist < radius_sq) return 1;
    else if (x_max >= 0.0f && max_dist < radius_sq) return 1;
  }
  return 0;
}

/* kernels.cl:313-340.  All three channels of light_color are equal, so one float carries the sum. */
static float direct_light(const Scene* s, const Ray* ray, int global_id) {
  const rt_config* c = s->cfg;
  float total = 0.0f;
  uint32_t r[3] = {(uint32_t)global_id, (uint32_t)((float)global_id * 91.0f), (uint32_t)((float)global_id * 19.0f)};
  rng_next(r);                                               /* :319 */
  const v3 dir = sub(s->light, ray->intersect);
  const v3 start = add(ray->intersect, scale(0.0001f, dir)); /* bias*dir : commutative */
  const float radius_sq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
  if (s->work) s->work->lit_hits++;
  for (int i = 0; i < c->shadow_samples; ++i) {
    rng_next(r);
    const int mask = !in_shadow(s, start, add(dir, crush(r, c->light_spread)), radius_sq);
    if (s->work) s->work->shadow_rays++;
    /* mask*(light_color*max(dot,0)) / (4*pi*r^2), :335 */
    total += ((float)mask * (16.0f * fmaxf(dot(dir, ray->normal), 0.0f))) / (4.0f * 3.14159274f * radius_sq);
  }
  return total / (float)c->shadow_samples;
}

/* kernels.cl:54-65 */
static Ray reflect_ray(const Ray* ray) {
  Ray o; memset(&o, 0, sizeof o);
  o.tri = -1; o.color[3] = 1.0f;
  const float dn = dot(ray->direction, ray->normal);
  o.direction = sub(ray->direction, scale(2.0f, scale(dn, ray->normal)));
  o.start = add(ray->intersect, scale(0.0001f, o.direction));
  o.medium = AIRF;
  o.direction = normalize(o.direction);
  return o;
}

/* kernels.cl:67-88.  The TIR branch (c2 < 0) is unreachable: sqrt of a negative is NaN (:77-80). */
static Ray refract_ray(const Ray* ray) {
  Ray o; memset(&o, 0, sizeof o);
  v3 normal = ray->normal;
  const int air = (ray->medium == AIRF);
  const float n1 = air ? AIRF : GLASSF, n2 = air ? GLASSF : AIRF;
  float c1 = dot(normal, ray->direction);
  if (c1 < 0.0f) normal = scale(-1.0f, normal);
  c1 = fabsf(c1);
  const float n = n1 / n2;
  const float c2 = sqrtf(1 - (n * n) * (1 - (c1 * c1)));
  if (c2 < 0.0f) return reflect_ray(ray);
  o.tri = -1;
  o.color[0] = 1.0f; o.color[3] = 1.0f;
  o.direction = add(scale(n, ray->direction), scale(n * c1 - c2, neg(normal)));
  o.start = add(ray->intersect, scale(0.0001f, o.direction));
  o.medium = n2;
  o.direction = normalize(o.direction);
  return o;
}

/* kernels.cl:342-365 */
static v3 secondary_light(const Scene* s, const Ray* ray, int global_id) {
  Ray p = *ray;
  for (int b = 0; b < s->cfg->max_bounces && p.color[3] <= 0.0f; ++b) {
    p = (p.color[3] == 0.0f) ? reflect_ray(&p) : refract_ray(&p);
    if (s->work) s->work->bounce_rays++;
    closest_hit(s, &p);
    if (p.tri != -1 && p.color[3] > 0.0f) {
      const float l = 0.5f + direct_light(s, &p, global_id);   /* indirect_light + direct, :354 */
      return V((0.9f * l) * p.color[0], (0.9f * l) * p.color[1], (0.9f * l) * p.color[2]);
    }
  }
  return V(0.0f, 0.0f, 0.0f);
}

/* kernels.cl:368-428 for one pixel.  rgb = final_color_total / aa_rays (before color_pixel). */
static void draw_pixel(const Scene* s, const float* rot, v3 cam, float focal, int x, int y, float rgb[3], uint32_t* argb) {
  const rt_config* c = s->cfg;
  const int rx = c->aa_x, ry = c->aa_y, aa = rx * ry;
  const float W = (float)c->width, H = (float)c->height;
  const int global_id = (int)((float)y * W + (float)x);        /* :380, float arithmetic then ->int */
  /* :384.  Units are AA sub-pixels along x; for non-square grids (which the reference cannot express)
   * the y sub-pixel pitch is rescaled by rx/ry — for rx==ry the factor is 1 and the expression is the
   * reference's own. */
  const float sy = (float)rx / (float)ry;
  const float bx = (float)(x * rx) - (W * (float)rx) / 2.0f;
  const float by = (float)(y * ry) - (H * (float)ry) / 2.0f;
  const v3 r0 = V(rot[0], rot[1], rot[2]), r1 = V(rot[4], rot[5], rot[6]), r2 = V(rot[8], rot[9], rot[10]);
  v3 total = V(0.0f, 0.0f, 0.0f);
  for (int dy = 0; dy < ry; ++dy) {
    for (int dx = 0; dx < rx; ++dx) {                          /* index dy*rx+dx : row-major, :393-407 */
      Ray ray; memset(&ray, 0, sizeof ray);
      ray.start = cam;
      const v3 d = V(bx + (float)dx, (by + (float)dy) * sy, focal + 0.0f);
      ray.direction = normalize(V(dot(r0, d), dot(r1, d), dot(r2, d)));
      ray.tri = -1; ray.medium = AIRF; ray.color[3] = 1.0f;
      if (s->work) s->work->primary_rays++;
      closest_hit(s, &ray);                                    /* :411 */
      if (ray.tri != -1) {                                     /* :416 */
        if (ray.color[3] <= 0.0f) {
          total = add(total, secondary_light(s, &ray, global_id));
        } else {
          const float l = 0.5f + direct_light(s, &ray, global_id);
          total = add(total, V(ray.color[0] * l, ray.color[1] * l, ray.color[2] * l));
        }
      }
    }
  }
  const float inv = (float)aa;
  rgb[0] = total.x / inv; rgb[1] = total.y / inv; rgb[2] = total.z / inv;
  /* color_pixel, :37-40 */
  const uint32_t R = (uint32_t)fminf(fmaxf(255 * rgb[0], 0.f), 255.f);
  const uint32_t G = (uint32_t)fminf(fmaxf(255 * rgb[1], 0.f), 255.f);
  const uint32_t B = (uint32_t)fminf(fmaxf(255 * rgb[2], 0.f), 255.f);
  *argb = (255u << 24) + (R << 16) + (G << 8) + B;
}

static void work_add(rt_work* a, const rt_work* b) {
  uint64_t* pa = (uint64_t*)a; const uint64_t* pb = (const uint64_t*)b;
  for (size_t k = 0; k < sizeof(rt_work) / 8; ++k) pa[k] += pb[k];
}

/* Render `npix` pixels (global ids y*W+x in `pix`), or — when pix==NULL — every pixel owned by the
 * band selection of cfg, in packed order.  out_rgb (nullable): 3 floats per pixel, final/aa_rays.
 * work (nullable): exact reference-semantics counters summed over the rendered pixels. */
int rto_render(const rt_config* cfg, const float* verts4, const float* normals4, const float* colors4, int n,
               const float* rot12, const float* cam3, const float* light3, float focal,
               uint32_t* out_argb, float* out_rgb, const int* pix, long npix, int nthreads, rt_work* work) {
  if (!cfg || cfg->width <= 0 || cfg->height <= 0 || cfg->aa_x <= 0 || cfg->aa_y <= 0) return RT_E_INVALID;
  const int W = cfg->width, H = cfg->height;
  const int bc = cfg->band_count > 0 ? cfg->band_count : 1;
  const int br = cfg->band_rows > 0 ? cfg->band_rows : H;
  int* own = NULL;
  if (!pix) {   /* enumerate the rows of this band selection */
    own = (int*)malloc((size_t)H * sizeof(int));
    int k = 0;
    for (int y = 0; y < H; ++y) if ((y / br) % bc == cfg->band_index) own[k++] = y;
    npix = (long)k * W;
  }
  if (nthreads < 1) nthreads = 1;
  rt_work* tw = work ? (rt_work*)calloc((size_t)nthreads, sizeof(rt_work)) : NULL;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
  for (long k = 0; k < npix; ++k) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    Scene s = {cfg, verts4, normals4, colors4, n, {light3[0], light3[1], light3[2]}, tw ? &tw[tid] : NULL};
    int x, y;
    if (pix) { x = pix[k] % W; y = pix[k] / W; } else { y = own[k / W]; x = (int)(k % W); }
    float rgb[3];
    draw_pixel(&s, rot12, V(cam3[0], cam3[1], cam3[2]), focal, x, y, rgb, &out_argb[k]);
    if (out_rgb) { out_rgb[3 * k] = rgb[0]; out_rgb[3 * k + 1] = rgb[1]; out_rgb[3 * k + 2] = rgb[2]; }
  }
  if (work) { memset(work, 0, sizeof *work); for (int t = 0; t < nthreads; ++t) work_add(work, &tw[t]); free(tw); }
  free(own);
  return RT_OK;
}

/* Function-level taps mirroring oracle/ref_shim.cpp's ref_in_shadow / ref_closest_hit. */
void rto_in_shadow(const rt_config* cfg, const float* verts4, const float* colors4, int n,
                   const float* rays, const float* radius_sq, long nray, uint8_t* out) {
  Scene s = {cfg, verts4, NULL, colors4, n, {0, 0, 0}, NULL};
  for (long k = 0; k < nray; ++k)
    out[k] = (uint8_t)in_shadow(&s, V(rays[6 * k], rays[6 * k + 1], rays[6 * k + 2]),
                                V(rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]), radius_sq[k]);
}
void rto_closest_hit(const rt_config* cfg, const float* verts4, const float* normals4, const float* colors4, int n,
                     const float* rays, long nray, int* out_tri, float* out10) {
  Scene s = {cfg, verts4, normals4, colors4, n, {0, 0, 0}, NULL};
  for (long k = 0; k < nray; ++k) {
    Ray r; memset(&r, 0, sizeof r);
    r.start = V(rays[6 * k], rays[6 * k + 1], rays[6 * k + 2]);
    r.direction = V(rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]);
    r.tri = -1; r.medium = AIRF; r.color[3] = 1.0f;
    closest_hit(&s, &r);
    out_tri[k] = r.tri;
    float* o = out10 + 10 * k;
    o[0] = r.intersect.x; o[1] = r.intersect.y; o[2] = r.intersect.z;
    o[3] = r.normal.x; o[4] = r.normal.y; o[5] = r.normal.z;
    memcpy(o + 6, r.color, 16);
  }
}
