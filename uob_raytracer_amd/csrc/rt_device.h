// rt_device.h — structures shared by the host API (rt_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uob_rt.h"

namespace uobrt {

// The wave kernel's job queue has several heads (a returning device-scope atomic on ONE word saturates at ~88
// hand-outs per microsecond on this part): head h hands out the jobs h, h + kJobHeads, h + 2 kJobHeads, ...
// An ODD number: a row has a power-of-two number of jobs (64 at 4096 pixels), and with 32 heads every head served the same
// two columns of the frame for ever — the heads over the penumbra columns ran dry long after the others, whose waves then
// walked from head to head.  Headline frame: 32 heads 3.08 ms, 8 / 16: 3.04, 12 / 24: 3.00, any odd count 21..31: 2.99; 64: 3.24.
constexpr int kJobHeads = 31;
constexpr int kJobHeadStride = 32;      // in 32-bit words: one 128-byte line per head
// HeavyState, in 32-bit words: [0] entries in the list, [2..3] sum of all job costs (64-bit, s_memtime ticks),
// [4] jobs counted
constexpr int kHeavyStateWords = 8;
// Largest |coordinate| rt_init / rt_render accept (vertices, sphere centres, camera, light, light_spread): the range
// over which the exact culls are verified against the unculled paths (tools/fuzz_paths.py --wide)
constexpr float kMaxCoordinate = 65536.0f;
constexpr int kRecordsPerTriangle = 8;   // float4 records per triangle staged by stage_triangles (rt_trace.h)

struct DevSphere {
  float cx, cy, cz, r2;
  float col[4];
};

// Everything one frame needs, passed by value as the kernel argument (lives in SGPRs / the scalar cache).
struct FrameParams {
  float rot[12];          // 3 rows x (x,y,z,pad), skeleton.cpp:149-151
  float cam[3];
  float focal;
  float light[3];
  float spread;           // light_spread, kernels.cl:317
  int32_t W, H, aa_x, aa_y;
  int32_t S;              // shadow samples
  int32_t bounces;
  int32_t nsph;
  int32_t n;              // triangles
  int32_t band_rows, band_index, band_count, owned_rows;
  float sy;               // aa_x / aa_y: y sub-pixel pitch in x sub-pixel units (1 for square grids)
  // Frame invariants of the reference's arithmetic, evaluated once on the host with the same FP32 operations (rt_api.hip
  // is built with -ffp-contract=off too).  gfx950 has no scalar float ALU: inside the kernel the compiler computes such
  // uniform values with VECTOR instructions, hoists them out of the job loop and keeps them in vector registers (17 of
  // the wave kernel's 96, all spilled to scratch and re-loaded for every job); as kernel arguments they are scalar operands.
  float half_wx, half_hy; // ((float)W * (float)aa_x) / 2.0f, ((float)H * (float)aa_y) / 2.0f   (kernels.cl:384)
  float w_f;              // (float)W                                                           (kernels.cl:380)
  float focal0;           // focal + 0.0f: z of the un-rotated ray direction
  float rzf[3];           // rot[2] * focal0, rot[6] * focal0, rot[10] * focal0: the last product of each row of R.d
  float hbox;             // light_spread / 2.f                                                 (kernels.cl:51)
  float light_inf;        // max |light_k|
  float job_hx, job_hy;   // wave kernel: half extent of a job's sub-pixel rectangle
  float job_eu[3];        // wave kernel: 1.0001 * (|r_k.x| * job_hx + |r_k.y| * job_hy): half width of the job's direction box
  uint32_t nseg_magic, band_rows_magic;   // ceil(2^32 / d) for q = n / d by one multiply-high (0: d == 1)
  float inv_S, inv_aa;    // 1/S, 1/(aa_x*aa_y) when that count is a power of two (x / 2^k == x * 2^-k bit for bit,
                          // one multiplication instead of an IEEE division), else 0
  int32_t n_shadow;       // triangles that can cast a shadow (glass removed), for the wave kernel
  DevSphere sph[RT_MAX_SPHERES];
  const DevSphere* sph_dev;   // the same table in device memory: the wave-mapped kernels stage it into LDS (as kernel
                              // arguments the spheres sit in 16-32 scalar registers for the whole kernel)
  const float4* verts;    // float4[3n]  (HBM, read once per workgroup while staging into LDS)
  const float4* normals;  // float4[n]
  const float4* colors;   // float4[n], w = material flag
  uint32_t* out_argb;     // owned_rows * W ARGB8888 words
  float4* out_rgb;        // nullable: owned_rows * W pre-quantisation colours
  int32_t out_global;     // 1: the output buffers are whole frames addressed by the GLOBAL row (a device of a
                          // multi-device context writing its bands straight into the assembled frame); 0: packed rows
  int32_t wave_blocks;    // wave kernel: workgroups the device holds at once (its persistent grid)
  unsigned long long* counters;  // nullable: rt_work, 8 x u64
  unsigned int* job_counter;   // wave kernel: kJobHeads queue heads, one per 128-B line (zeroed before each launch)
  int32_t njobs, nseg;    // wave kernel: jobs in total / per row
  int32_t aa_magic;       // wave kernel: ceil(65536 / aa): lane / aa == (lane * aa_magic) >> 16 for lane < 64
  int32_t aax_magic;      // ceil(65536 / aa_x): a / aa_x == (a * aax_magic) >> 16 for an AA sample index a < 4096
  int32_t job_tasks;      // wave kernel: 64-ray tasks per job (a job = job_tasks * 64 / aa consecutive pixels of a row)
  int32_t split_listed;   // wave kernel: 1 = last frame's expensive jobs are handed out one task at a time (short frames)
  int32_t no_specialise;  // wave kernel: 1 = always the generic instantiation (UOB_RT_NO_SPECIALISE: A/B and equivalence tests)
  // wave kernel: jobs that were expensive in the previous frame of this context are handed out first (the kernel
  // ends when its last job does, so the long ones should start early); nullptr = plain order
  const unsigned int* heavy_prev;     // their job ids
  const unsigned int* heavy_prev_state;   // HeavyState of the previous frame
  unsigned int* heavy_new;            // this frame's expensive jobs, appended as they finish
  unsigned int* heavy_new_state;
  const unsigned int* heavy_flags;    // per job: == heavy_gen when the job is in heavy_prev (phase A renders it; the plain sequence skips it)
  unsigned int* heavy_flags_new;      // per job: == heavy_gen + 1 once the job is on the list this frame builds (the two arrays swap
                                      // roles every frame; stale generations compare unequal, so neither is ever cleared)
  uint32_t heavy_gen;
  int32_t heavy_factor4;              // a job is expensive above heavy_factor4 / 4 times the average job cost
  int32_t heavy_cap;
  float l1_inflate;                   // wave kernel: level 1 bounds a point set this many times as wide as the task's own, and the job's
                                      // next tasks reuse it while they stay inside (1 = the task's own set, nothing to reuse)
  int32_t heavy_dilate;               // 1: an expensive job is listed together with its two neighbours in the row
  float4* records;        // staged triangle records in HBM (8 x n float4), used when n exceeds one LDS stage
  // mesh kernel: persistent workgroups pull 16x16-pixel blocks; last frame's expensive blocks first
  const unsigned int* mesh_order;   // job order of this frame (nullptr: plain order)
  unsigned int* mesh_cost;          // per block: s_memtime ticks it took this frame (nullptr: not recorded)
  unsigned int* mesh_order_out;     // next frame's order, written by rt_mesh_order after the frame
  unsigned int* mesh_queue_len;     // its length (a very expensive block is entered as four cooperative sub-block jobs)
  int32_t mesh_blocks;              // workgroups the device holds at once
  const int* orig;        // mesh kernel: original index of every (reordered) triangle; nullptr = the order is the original
  const float4* tile_box; // mesh kernel: per 64-triangle tile, 3 float4: box lo.xyz | eta, box hi.xyz | sigma, normal-cone axis | chi (rt_api.hip)
  // mesh kernel: per-frame candidate-tile masks (rt_kernel_mesh.hip), nullptr = visit every tile
  unsigned long long* screen_masks;   // [scy][scx][nwords]: tiles a primary ray through that 64x64-pixel cell may hit
  unsigned long long* world_masks;    // [G][G][G][nwords]: tiles that may shadow a surface point inside that world cell
  unsigned int* world_occ;            // bitmaps of the world cells that can hold a surface point: G^3, (G/2)^3, (G/4)^3 bits
  int32_t mask_debug;                 // diagnostic (UOB_RT_MASK_DEBUG): 1 = ignore screen masks, 2 = ignore world masks, 4 = every world cell occupied,
                                      // 8 = every block of the next frame cooperative (tests), 16 = a context's first frame in row order (no cost guess),
                                      // 64 = bounce rays visit every tile (no tile pre-test), 256 = shadow-ray masks without the tile-level certificate
  int32_t nwords, scx, scy, grid_g;   // 64-bit words per mask; screen cells per row / column; world cells per axis
  float grid_lo[3], grid_cell, grid_inv;   // world grid: origin, cell edge, 1 / cell edge
};

// n / d for n < 2^16 * ... (rows, jobs) with the host's magic = ceil(2^32 / d): exact while n * (magic * d - 2^32) < 2^32,
// which holds for every row index (< 2^15) and job index (< 2^20, d <= 2^11) of a frame; magic 0 stands for d == 1
__device__ inline uint32_t div_magic(uint32_t n, uint32_t magic) { return magic ? __umulhi(n, magic) : n; }

// Map a packed local row index to the global image row (band partition, include/uob_rt.h rt_config).
__host__ __device__ inline int band_global_row(int lr, int band_rows, int band_index, int band_count) {
  return ((lr / band_rows) * band_count + band_index) * band_rows + (lr % band_rows);
}
// the same with the division done by the frame's magic number
__device__ inline int band_global_row(const struct FrameParams& P, int lr);

__device__ inline int band_global_row(const FrameParams& P, int lr) {
  const int q = (int)div_magic((uint32_t)lr, P.band_rows_magic);
  return (q * P.band_count + P.band_index) * P.band_rows + (lr - q * P.band_rows);
}

}  // namespace uobrt
