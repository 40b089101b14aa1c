// rt_kernel_generic.hip — one-thread-per-pixel gfx950 ray tracer for ANY rt_config.
//
// This is the general path (any AA grid, any sample count, spheres, bounces, band partitions) and the
// instrumented work counter.  It keeps the reference's loop structure (kernel `draw`,
// Source/kernels.cl:368-428) so the early-exit semantics of in_shadow (:243-311) are reproduced
// literally, which the exact work counters need.  The headline configuration (64 shadow samples) runs
// on the wave-per-hit-point kernel in rt_kernel_wave.hip instead.
//
// Layout: a workgroup is 64x4 pixels (4 waves, one image-row segment of 64 pixels per wave, so every
// framebuffer store is a fully coalesced 256-B / 1-KiB wave access).  The triangle list is staged ONCE
// per workgroup into LDS as float4 SoA records (rt_trace.h): all lanes of a wave read the same record in
// the intersection loops, which LDS serves as a broadcast.
// Compiled with -ffp-contract=off: see rt_math.h for the numerics contract.
#include "rt_trace.h"

namespace uobrt {

namespace {

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
    // mask*(light_color*max(dot,0)) / (4 pi r^2): 1.0f*x == x exactly; a shadowed sample adds 0*x
    total += sh ? 0.0f * term : term;
  }
  return div_count(total, P.S, P.inv_S);
}

}  // namespace

// Stage the records of a mesh that does not fit one LDS stage into HBM instead (P.records), once per frame.
__global__ __launch_bounds__(256) void rt_stage_records(const FrameParams P) {
  stage_triangles(P, P.records, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// kernels.cl:368-428.  Grid: x = ceil(W/64), y = ceil(owned_rows/4); block 64x4.
// BIG = the triangle records are read from HBM/L2 (every lane the same address) instead of LDS: the path
// for meshes of more than kLdsMaxTriangles triangles until the LDS-tiled traversal exists.
template <bool COUNT, bool BIG>
__global__ __launch_bounds__(256) void rt_draw_generic(const FrameParams P) {
  extern __shared__ float4 lds_dyn[];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const float4* lds = BIG ? P.records : lds_dyn;
  if (!BIG) {
    stage_triangles(P, lds_dyn, tid, 256);
    __syncthreads();
  }

  const int x = blockIdx.x * 64 + threadIdx.x;
  const int lr = blockIdx.y * 4 + threadIdx.y;
  Work wk;
  if (COUNT) for (int k = 0; k < 8; ++k) wk.v[k] = 0;

  if (x < P.W && lr < P.owned_rows) {
    const LdsScene S = lds_scene(lds, P.n);
    const int y = band_global_row(P, lr);
    const int global_id = pixel_global_id(P, x, y);
    f3 total = mk(0.f, 0.f, 0.f);
    for (int dy = 0; dy < P.aa_y; ++dy) {
      for (int dx = 0; dx < P.aa_x; ++dx) {
        Ray ray = primary_ray(P, x, y, dx, dy);
        if (COUNT) wk.v[W_PRIMARY]++;
        closest_hit_primary<COUNT>(S, P, ray, wk);
        if (ray.tri != -1) {
          if (ray.col.w <= 0.0f) {          // mirror or glass: secondary_light, kernels.cl:342-365
            if (bounce_to_diffuse<COUNT>(S, P, ray, wk)) {
              const float k = 0.9f * (0.5f + direct_light<COUNT>(S, P, ray, global_id, wk));
              total = total + mk(k * ray.col.x, k * ray.col.y, k * ray.col.z);
            } else {
              total = total + mk(0.f, 0.f, 0.f);
            }
          } else {
            const float l = 0.5f + direct_light<COUNT>(S, P, ray, global_id, wk);
            total = total + mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
          }
        }
      }
    }
    const int aa = P.aa_x * P.aa_y;
    const f3 c = mk(div_count(total.x, aa, P.inv_aa), div_count(total.y, aa, P.inv_aa), div_count(total.z, aa, P.inv_aa));
    const size_t o = (size_t)(P.out_global ? y : lr) * P.W + x;
    if (!COUNT) {
      P.out_argb[o] = pack_argb(c);
      if (P.out_rgb) P.out_rgb[o] = make_float4(c.x, c.y, c.z, 1.0f);
    }
  }
  if (COUNT) {   // one atomic per counter per wave
    for (int k = 0; k < 8; ++k) {
      unsigned long long v = wk.v[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0 && v) atomicAdd(&P.counters[k], v);
    }
  }
}

template __global__ void rt_draw_generic<false, false>(const FrameParams);
template __global__ void rt_draw_generic<true, false>(const FrameParams);
template __global__ void rt_draw_generic<false, true>(const FrameParams);
template __global__ void rt_draw_generic<true, true>(const FrameParams);

// Diagnostic (rt_debug_trace_rays): the device functions on caller rays, one lane per ray.
//   what 0: in_shadow (kernels.cl:243-311)   what 1: single_ray_intersections (:168-241)
template <bool BIG>
__global__ __launch_bounds__(256) void rt_trace_rays(const FrameParams P, int what, const float* rays, const float* r2, long nray,
                                                     int* out_tri, float* out10) {
  extern __shared__ float4 lds_dyn[];
  const float4* lds = BIG ? P.records : lds_dyn;
  if (!BIG) {
    stage_triangles(P, lds_dyn, threadIdx.x, 256);
    __syncthreads();
  }
  const LdsScene S = lds_scene(lds, P.n);
  Work wk;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nray; k += (long)gridDim.x * 256) {
    const f3 start = mk(rays[6 * k], rays[6 * k + 1], rays[6 * k + 2]), dir = mk(rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]);
    if (what == 0) {
      out_tri[k] = in_shadow<false>(S, P, start, dir, r2[k], wk) ? 1 : 0;
    } else {
      Ray ray;
      ray.start = start; ray.dir = dir; ray.tri = -1; ray.medium = RT_AIR;
      ray.col = make_float4(0.f, 0.f, 0.f, 1.0f);
      ray.P = mk(0.f, 0.f, 0.f); ray.N = mk(0.f, 0.f, 0.f);
      closest_hit<false>(S, P, ray, wk);
      out_tri[k] = ray.tri;
      if (ray.tri != -1) {
        float* o = out10 + 10 * k;
        o[0] = ray.P.x; o[1] = ray.P.y; o[2] = ray.P.z; o[3] = ray.N.x; o[4] = ray.N.y; o[5] = ray.N.z;
        o[6] = ray.col.x; o[7] = ray.col.y; o[8] = ray.col.z; o[9] = ray.col.w;
      }
    }
  }
}

bool generic_needs_records(int n) { return n > kLdsMaxTriangles; }

// P.records must already hold the staged records when the mesh exceeds one LDS stage
void launch_trace_rays(const FrameParams& P, int what, const float* d_rays, const float* d_r2, long nray, int* d_tri,
                       float* d_out10, hipStream_t stream) {
  const long blocks = (nray + 255) / 256;
  const dim3 grid((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096));
  if (generic_needs_records(P.n)) hipLaunchKernelGGL((rt_trace_rays<true>), grid, dim3(256), 0, stream, P, what, d_rays, d_r2, nray, d_tri, d_out10);
  else hipLaunchKernelGGL((rt_trace_rays<false>), grid, dim3(256), (size_t)P.n * kLdsRecords * sizeof(float4), stream, P, what, d_rays, d_r2, nray, d_tri, d_out10);
}

void launch_stage_records(const FrameParams& P, hipStream_t stream) {
  hipLaunchKernelGGL(rt_stage_records, dim3((P.n + 255) / 256), dim3(256), 0, stream, P);
}

void launch_generic(const FrameParams& P, bool count, hipStream_t stream) {
  const dim3 block(64, 4);
  const dim3 grid((P.W + 63) / 64, (P.owned_rows + 3) / 4);
  if (generic_needs_records(P.n)) {
    launch_stage_records(P, stream);
    if (count) hipLaunchKernelGGL((rt_draw_generic<true, true>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((rt_draw_generic<false, true>), grid, block, 0, stream, P);
    return;
  }
  const size_t lds_bytes = (size_t)P.n * kLdsRecords * sizeof(float4);
  if (count) hipLaunchKernelGGL((rt_draw_generic<true, false>), grid, block, lds_bytes, stream, P);
  else hipLaunchKernelGGL((rt_draw_generic<false, false>), grid, block, lds_bytes, stream, P);
}

}  // namespace uobrt
