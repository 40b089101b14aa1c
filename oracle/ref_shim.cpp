// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// Host shim that lets the reference's own OpenCL kernel file be executed on x86-64:
// `oracle/build_ref.py` compiles /root/reference/Source/kernels.cl *where it lies* with
// `clang -x cl -target x86_64` into an object exporting `draw` (kernels.cl:368) and its helper
// functions, and links it with this file.  This file supplies
//   * the 17 OpenCL builtins the object leaves undefined, and
//   * a driver (`ref_render`, `ref_in_shadow`, `ref_closest_hit`) that calls the reference
//     functions once per pixel / per ray from host threads.
// Builtin semantics (the "strict" oracle): native_recip(x)=1.0f/x, native_divide=a/b,
// native_sqrt=sqrt=sqrtf, normalize(v)=v/sqrtf(dot(v,v)), dot = x*x+y*y+z*z left to right,
// min/max = fminf/fmaxf, convert_uint3 = C truncation, convert_float3 = C (float).
// Outputs of this build go to oracle/_ref/ only (git-ignored).  No reference source is copied.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef float    float3 __attribute__((ext_vector_type(3)));
typedef float    float4 __attribute__((ext_vector_type(4)));
typedef uint32_t uint3  __attribute__((ext_vector_type(3)));
typedef void*    event_t;

static thread_local int   tl_gid[2];
static thread_local float tl_rgb[3];

// ---- OpenCL builtins (Itanium names as left undefined by the kernel object) ------------------
float  cl_native_sqrt(float x)            asm("_Z11native_sqrtf");
float  cl_native_recip(float x)           asm("_Z12native_recipf");
float  cl_native_divide(float a, float b) asm("_Z13native_divideff");
float  cl_sqrt(float x)                   asm("_Z4sqrtf");
float  cl_fabs(float x)                   asm("_Z4fabsf");
float  cl_min(float a, float b)           asm("_Z3minff");
float  cl_max(float a, float b)           asm("_Z3maxff");
float3 cl_min3(float3 a, float b)         asm("_Z3minDv3_ff");
float3 cl_max3(float3 a, float b)         asm("_Z3maxDv3_ff");
float  cl_dot(float3 a, float3 b)         asm("_Z3dotDv3_fS_");
float3 cl_normalize(float3 v)             asm("_Z9normalizeDv3_f");
uint3  cl_convert_uint3(float3 v)         asm("_Z13convert_uint3Dv3_f");
float3 cl_convert_float3(uint3 v)         asm("_Z14convert_float3Dv3_j");
size_t cl_get_global_id(unsigned d)       asm("_Z13get_global_idj");
void   cl_wait_group_events(int n, event_t* e) asm("_Z17wait_group_eventsiPU9CLprivate9ocl_event");
event_t cl_awgc3(float3* dst, const float3* src, size_t n, event_t e)
    asm("_Z21async_work_group_copyPU7CLlocalDv3_fPU8CLglobalKS_m9ocl_event");
event_t cl_awgc4(float4* dst, const float4* src, size_t n, event_t e)
    asm("_Z21async_work_group_copyPU7CLlocalDv4_fPU8CLglobalKS_m9ocl_event");

float  cl_native_sqrt(float x)            { return sqrtf(x); }
float  cl_native_recip(float x)           { return 1.0f / x; }
float  cl_native_divide(float a, float b) { return a / b; }
float  cl_sqrt(float x)                   { return sqrtf(x); }
float  cl_fabs(float x)                   { return fabsf(x); }
float  cl_min(float a, float b)           { return fminf(a, b); }
float  cl_max(float a, float b)           { return fmaxf(a, b); }
float3 cl_min3(float3 a, float b) { float3 r; r.x = fminf(a.x, b); r.y = fminf(a.y, b); r.z = fminf(a.z, b); return r; }
float3 cl_max3(float3 a, float b) { float3 r; r.x = fmaxf(a.x, b); r.y = fmaxf(a.y, b); r.z = fmaxf(a.z, b); return r; }
float  cl_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3 cl_normalize(float3 v) {
  const float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  float3 r; r.x = v.x / len; r.y = v.y / len; r.z = v.z / len; return r;
}
uint3 cl_convert_uint3(float3 v) {
  // the only call site is color_pixel (kernels.cl:38): tap the pre-quantisation colour here
  tl_rgb[0] = v.x; tl_rgb[1] = v.y; tl_rgb[2] = v.z;
  uint3 r; r.x = (uint32_t)v.x; r.y = (uint32_t)v.y; r.z = (uint32_t)v.z; return r;
}
float3 cl_convert_float3(uint3 v) { float3 r; r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; return r; }
size_t cl_get_global_id(unsigned d) { return (size_t)tl_gid[d & 1]; }
void   cl_wait_group_events(int, event_t*) {}
event_t cl_awgc3(float3* dst, const float3* src, size_t n, event_t) { if (dst != src) memcpy(dst, src, n * sizeof(float3)); return nullptr; }
event_t cl_awgc4(float4* dst, const float4* src, size_t n, event_t) { if (dst != src) memcpy(dst, src, n * sizeof(float4)); return nullptr; }

// ---- the reference's own functions (defined by the compiled kernels.cl object) ---------------
struct RefRay {            // mirrors `Ray`, kernels.cl:21-29 (float3 is 16 B / 16-aligned)
  float3 start, direction, intersect, intersect_normal;
  float4 intersect_color;
  float  medium;
  int    intersect_triangle;
};
extern "C" void draw(uint32_t* screen, float3* verts, float3* normals, float4* colors, float3* rot,
                     float3 cam, float3 light, int n, float focal,
                     float3* lverts, float3* lnormals, float4* lcolors);
extern "C" bool in_shadow(float3 start, float3 dir, float3* verts, float4* colors, float radius_sq, int n);
extern "C" void single_ray_intersections(RefRay* ray, float3* verts, float3* normals, float4* colors, int n);

#ifndef REF_W
#error "build with -DREF_W=<SCREEN_WIDTH the kernel object was compiled with> -DREF_H=..."
#endif

extern "C" int ref_width()  { return REF_W; }
extern "C" int ref_height() { return REF_H; }

// Render `npix` pixels (ids y*W+x in `pix`, or the whole frame when pix==NULL) with the reference
// kernel.  Scene arrays use the packed layout of skeleton.cpp:474-484 (float4 per vertex/normal/colour).
// out_argb / out_rgb are indexed by position in `pix` (or by pixel id for a whole frame); out_rgb
// (nullable) receives min(max(255*colour,0),255) as floats, 3 per pixel.
extern "C" int ref_render(const float* verts4, const float* normals4, const float* colors4, int n,
                          const float* rot12, const float* cam3, const float* light3, float focal,
                          uint32_t* out_argb, float* out_rgb, const int* pix, long npix, int nthreads) {
  const int W = REF_W, H = REF_H;
  if (!pix) npix = (long)W * H;
  std::vector<uint32_t> screen((size_t)W * H, 0u);
  if (nthreads < 1) nthreads = 1;
  auto worker = [&](int tid) {
    float3 cam, light; cam.x = cam3[0]; cam.y = cam3[1]; cam.z = cam3[2];
    light.x = light3[0]; light.y = light3[1]; light.z = light3[2];
    for (long k = tid; k < npix; k += nthreads) {
      const int id = pix ? pix[k] : (int)k;
      tl_gid[0] = id % W; tl_gid[1] = id / W;
      draw(screen.data(), (float3*)verts4, (float3*)normals4, (float4*)colors4, (float3*)rot12,
           cam, light, n, focal, (float3*)verts4, (float3*)normals4, (float4*)colors4);
      out_argb[k] = screen[(size_t)id];
      if (out_rgb) { out_rgb[3 * k] = tl_rgb[0]; out_rgb[3 * k + 1] = tl_rgb[1]; out_rgb[3 * k + 2] = tl_rgb[2]; }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
  worker(0);
  for (auto& t : th) t.join();
  return 0;
}

// Function-level taps for the restatement's unit tests -----------------------------------------
// rays: nray x 6 floats (start xyz, dir xyz); radius_sq per ray; out: 0/1 per ray (kernels.cl:243)
extern "C" void ref_in_shadow(const float* verts4, const float* colors4, int n,
                              const float* rays, const float* radius_sq, long nray, uint8_t* out) {
  for (long k = 0; k < nray; ++k) {
    float3 s, d; s.x = rays[6 * k]; s.y = rays[6 * k + 1]; s.z = rays[6 * k + 2];
    d.x = rays[6 * k + 3]; d.y = rays[6 * k + 4]; d.z = rays[6 * k + 5];
    out[k] = in_shadow(s, d, (float3*)verts4, (float4*)colors4, radius_sq[k], n) ? 1 : 0;
  }
}
// closest hit of one ray (kernels.cl:168): out_tri = intersect_triangle, out10 = intersect xyz,
// normal xyz, colour xyzw (only meaningful when out_tri != -1)
extern "C" void ref_closest_hit(const float* verts4, const float* normals4, const float* colors4, int n,
                                const float* rays, long nray, int* out_tri, float* out10) {
  for (long k = 0; k < nray; ++k) {
    RefRay r; memset(&r, 0, sizeof r);
    r.start.x = rays[6 * k]; r.start.y = rays[6 * k + 1]; r.start.z = rays[6 * k + 2];
    r.direction.x = rays[6 * k + 3]; r.direction.y = rays[6 * k + 4]; r.direction.z = rays[6 * k + 5];
    r.intersect_triangle = -1; r.medium = 1.0f; r.intersect_color.w = 1.0f;
    single_ray_intersections(&r, (float3*)verts4, (float3*)normals4, (float4*)colors4, n);
    out_tri[k] = r.intersect_triangle;
    float* o = out10 + 10 * k;
    o[0] = r.intersect.x; o[1] = r.intersect.y; o[2] = r.intersect.z;
    o[3] = r.intersect_normal.x; o[4] = r.intersect_normal.y; o[5] = r.intersect_normal.z;
    o[6] = r.intersect_color.x; o[7] = r.intersect_color.y; o[8] = r.intersect_color.z; o[9] = r.intersect_color.w;
  }
}
