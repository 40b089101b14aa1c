/* uob_rt.h — C ABI of the MI355X-native Cornell-Box ray tracer (libuob_rt.so).
 *
 * This library replaces ONE path of harrywaugh/UOB_Raytracer: the OpenCL device boundary of
 * Source/skeleton.cpp — `opencl_initialise` (:366-497, build kernel + upload scene once) and
 * `offload_rendering` (:146-182, per-frame args + clEnqueueNDRangeKernel(draw) + blocking readback) —
 * and the kernel behind it, `draw` (Source/kernels.cl:368-428).  Plain pointers and sizes only; no C++
 * or torch types cross this boundary.  A maintainer's binding is shown in INTEGRATION.md.
 *
 * Conventions: every function returns RT_OK (0) or a negative RT_E_* code; the message for the last
 * failure on the calling thread is available from rt_last_error().  A context is not thread-safe
 * (the reference has one host thread and one in-order queue, skeleton.cpp:388), and its frames run one at a
 * time: a frame enqueued on another stream first waits for the context's previous frame.
 * State a context carries from frame to frame: the wave kernel hands out the row segments that were expensive
 * in the context's PREVIOUS frame first (scheduling only — no pixel depends on it; RT_FLAG_PLAIN_ORDER
 * switches it off).  Tuning knobs are read ONCE, in rt_init, from the environment (UOB_RT_JOB_TASKS,
 * UOB_RT_HEAVY_FACTOR4, UOB_RT_FULL_GRID, UOB_RT_SPLIT_LISTED, UOB_RT_TIMELINE: see DESIGN.md 4.1); nothing reads the
 * environment afterwards.
 */
#ifndef UOB_RT_H
#define UOB_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2
#define RT_MAX_SPHERES 4
#define RT_MAX_DEVICES 8

enum {
  RT_OK = 0,
  RT_E_INVALID = -1,  /* bad argument / configuration                                            */
  RT_E_DEVICE = -2,   /* HIP runtime error (message carries hipGetErrorString)                   */
  RT_E_NOMEM = -3,
  RT_E_IO = -4,       /* file open/parse failure (OBJ loader, image writer)                      */
  RT_E_UNSUPPORTED = -5
};

/* One analytic sphere: kernels.cl:7-10 (sphere_centers / sphere_colors / sphere_radius_sqs).
 * color[3] is the material flag exactly like Triangle::color.w: >0 diffuse, 0 mirror, <0 glass.  */
typedef struct rt_sphere {
  float center[3];
  float radius_sq;
  float color[4];
} rt_sphere;

/* Every knob that the reference hard-codes as a #define/const (SURVEY.md §5 "Config / flags").
 * rt_config_default() fills in the reference's shipped values.                                   */
typedef struct rt_config {
  int32_t width, height;        /* SCREEN_WIDTH / SCREEN_HEIGHT, kernels.cl:16-17, skeleton.cpp:32-33 */
  int32_t aa_x, aa_y;           /* rays_x / rays_y, kernels.cl:12-13 (aa_rays = aa_x*aa_y); each 1..16        */
  int32_t shadow_samples;       /* light_sources, kernels.cl:316                                       */
  float   light_spread;         /* kernels.cl:317                                                      */
  int32_t max_bounces;          /* bounces, kernels.cl:343                                             */
  int32_t num_spheres;          /* SPHERES, kernels.cl:7 (0..RT_MAX_SPHERES)                           */
  rt_sphere spheres[RT_MAX_SPHERES];
  /* Row-band partition of the frame for multi-GPU rendering: this context owns the rows y with
   * (y / band_rows) % band_count == band_index, packed top to bottom in its output buffer.
   * band_count = 1 renders the whole frame.  Ray directions and the per-pixel RNG seed always use the
   * GLOBAL pixel coordinates (kernels.cl:378-380).                                                   */
  int32_t band_rows, band_index, band_count;
  int32_t device;               /* HIP device ordinal; -1 = current device                            */
  int32_t flags;                /* RT_FLAG_*                                                           */
  /* Several GPUs inside ONE context (one host thread, one stream per device — SURVEY.md 8(b) "Threading"):
   * with num_devices > 1 the rows this context owns are split into interleaved bands of device_band_rows
   * rows over devices[0..num_devices-1] (the same ordinal may be listed more than once), every device renders
   * its bands on its own stream, and rt_render / rt_render_device deliver the assembled frame exactly as a
   * single device would (bit-identical).  num_devices <= 1: `device` alone.  Needs band_count == 1.        */
  int32_t num_devices;
  int32_t devices[RT_MAX_DEVICES];
  int32_t device_band_rows;     /* 0 = 32                                                              */
} rt_config;

#define RT_FLAG_GENERIC_KERNEL 2u   /* always use the one-thread-per-pixel kernel (A/B and parity tests)  */
#define RT_FLAG_NO_CULL 4u          /* wave kernel: test every triangle for every surface point (no interval
                                       culling); output is bit-identical either way                        */
#define RT_FLAG_NO_TILE_BINS 8u     /* mesh kernel (n > 64): visit every 64-triangle tile instead of the per-frame
                                       candidate-tile masks; output is bit-identical either way                */
#define RT_FLAG_PLAIN_ORDER 16u     /* wave kernel: hand the jobs out in plain order every frame (no "last frame's
                                       expensive jobs first"): the context then carries no state from frame to frame */
#define RT_FLAG_STAGED_GATHER 32u   /* several devices: every device renders into its own stripe and the bands are
                                       copied to the destination, also for devices that could write it directly   */

typedef struct rt_ctx rt_ctx;

/* Per-frame work counters filled by rt_count_work (exact, reference early-exit semantics). */
typedef struct rt_work {
  uint64_t primary_rays, bounce_rays, shadow_rays;
  uint64_t closest_tri_tests, closest_sphere_tests;   /* batch_/single_ray_intersections loops    */
  uint64_t shadow_tri_tests, shadow_sphere_tests;     /* in_shadow loops incl. its early return   */
  uint64_t lit_hits;                                  /* direct_light invocations                 */
} rt_work;

/* ---- configuration -------------------------------------------------------------------------- */
/* Reference constants: 1024x1024, 2x2 AA, 10 shadow samples, spread 0.05, 10 bounces, the two live
 * spheres of kernels.cl:8-10, whole frame on the current device.                                     */
void rt_config_default(rt_config* cfg);
/* Number of rows / pixels of the frame owned by cfg's band selection.                                */
int32_t rt_config_owned_rows(const rt_config* cfg);

/* ---- the device boundary (replaces skeleton.cpp:366-497 and :146-182) ----------------------- */
/* Upload the scene once.  Arrays use the reference's packed layout (skeleton.cpp:474-484):
 * vertices4 = float4[3n] (w ignored), normals4 = float4[n] (w ignored), colors4 = float4[n] with
 * w = material flag.  The caller keeps ownership; data is copied before the call returns (as the
 * CL_TRUE writes at skeleton.cpp:486-496 do).                                                        */
int rt_init(const rt_config* cfg, const float* vertices4, const float* normals4, const float* colors4,
            int32_t n_triangles, rt_ctx** out_ctx);

/* Render one frame and read it back: rot = 3 rows x (x,y,z,pad) exactly as rot_matrix[12] at
 * skeleton.cpp:149-151; cam/light = first 12 bytes of camera_position / light_position (:162,:164);
 * focal = focal_length (:166), in units of AA sub-pixels along x.  out_argb receives
 * owned_rows*width ARGB8888 words (A=255, kernels.cl:39); out_rgb_f32 (nullable) receives the
 * pre-quantisation colour final/aa_rays as float4 (w=1) per pixel — the parity tap.  Synchronous,
 * like the CL_TRUE read at skeleton.cpp:179.                                                         */
int rt_render(rt_ctx* ctx, const float rot[12], const float cam[3], const float light[3], float focal,
              uint32_t* out_argb, float* out_rgb_f32);

/* Same frame, but the ARGB (and optional float4) output stays in device memory the caller owns
 * (e.g. a torch tensor handed to an RCCL gather).  Enqueued on `hip_stream` (a hipStream_t, may be
 * NULL for the default stream); returns without synchronising.  With several devices in the context
 * (rt_config.devices) the buffers must live on devices[0] and hip_stream must be a stream of that device: the other
 * devices' bands arrive by peer copy, ordered after the caller's earlier work on the stream and before its later work. */
int rt_render_device(rt_ctx* ctx, const float rot[12], const float cam[3], const float light[3],
                     float focal, void* d_out_argb, void* d_out_rgb_f32, void* hip_stream);

/* Exact work counters of the frame (un-timed instrumented pass; reference semantics).               */
int rt_count_work(rt_ctx* ctx, const float rot[12], const float cam[3], const float light[3],
                  float focal, rt_work* out);

/* Work the wave kernel actually EXECUTES for the frame (un-timed instrumented pass; fails with
 * RT_E_UNSUPPORTED for configurations that run on the generic kernel).  out[0] = surface points whose
 * samples were tested (level 3), out[1] = first-stage (t) sample-test passes = 64 sample tests each,
 * out[2] = second-stage (u,v) passes, out[3] = wave-wide sphere evaluations, out[4] = lit surface points
 * decided fully lit by the interval bounds, out[5] = 64-ray tasks that needed no sampling at all,
 * out[6..7] = 0.
 * Meshes (n > 64, tiled kernel): out[0] = (wave, tile) visits of the primary pass, out[1] = triangles left by
 * the primary bound over those visits, out[2] = (wave, tile) visits of the shadow pass, out[3] = triangles
 * left by level 1, out[4] = level-3 point-pair calls, out[5] = their first-stage passes, out[6] = the longest
 * 16x16-pixel block in s_memtime ticks (shader cycles), out[7] = task rounds per wave summed over waves.                                                                       */
int rt_count_executed(rt_ctx* ctx, const float rot[12], const float cam[3], const float light[3],
                      float focal, uint64_t out[8]);

/* Device time of the most recent rt_render / rt_render_device kernel(s) on this context in ms,
 * measured with hipEvents on the launch stream (synchronises that stream).                          */
int rt_last_kernel_ms(rt_ctx* ctx, float* out_ms);

/* Diagnostic: run the device functions of the path on caller-supplied rays (host arrays), one lane per ray,
 * against the context's scene and sphere table — the function-level golden vectors of the reference are checked
 * through this entry (tests/test_gpu_functions.py).  rays6 = nray x (start.xyz, direction.xyz).
 *   RT_TRACE_IN_SHADOW   : in_shadow (kernels.cl:243-311) with radius_sq[nray]  -> out_tri[k] = 0 / 1
 *   RT_TRACE_CLOSEST_HIT : single_ray_intersections (kernels.cl:168-241) -> out_tri[k] = -1 / -2 / triangle,
 *                          out10[10k..] = intersect.xyz, normal.xyz, colour.xyzw (zero on a miss)          */
enum { RT_TRACE_IN_SHADOW = 0, RT_TRACE_CLOSEST_HIT = 1 };
int rt_debug_trace_rays(rt_ctx* ctx, int32_t what, const float* rays6, const float* radius_sq, int64_t nray,
                        int32_t* out_tri, float* out10);

/* Diagnostic, mesh kernel (n > 64): the cost of every 16x16-pixel block of the most recent frame in s_memtime ticks
 * (shader cycles) — the scheduling state "last frame's expensive blocks first" is built from it.  Row-major over
 * ceil(owned_rows/16) x ceil(width/16) blocks; writes min(count, cap) values, returns the block count, or
 * RT_E_UNSUPPORTED when the context keeps no such state (n <= 64, RT_FLAG_PLAIN_ORDER, generic kernel).          */
int rt_debug_block_costs(rt_ctx* ctx, uint32_t* out, int32_t cap);

/* Diagnostic, mesh kernel (n > 64): the most recent frame's shadow-ray tile masks — for every world cell (x fastest,
 * G x G x G cells) `words` 64-bit words, bit t = "a shadow ray that starts in this cell may hit a triangle of tile t" (tiles in
 * the kernel's own order; a cell no surface point can start from reads 0).  Writes min(count, cap) words, returns the word
 * count G^3 * words and stores G and words, or RT_E_UNSUPPORTED when the context builds no tile masks.                       */
int rt_debug_world_masks(rt_ctx* ctx, uint64_t* out, int64_t cap, int32_t* grid, int32_t* words);

/* Optional: let the device write the frame STRAIGHT into the caller's host framebuffer (screen->buffer,
 * SDLauxiliary.h:105) instead of rendering into device memory and copying 4 bytes per pixel back after the kernel
 * (clEnqueueReadBuffer, skeleton.cpp:179-180): the pixels cross PCIe while the frame is still being rendered.
 * rt_register_output pins and maps `bytes` bytes at `host` until rt_unregister_output / rt_destroy; every later
 * rt_render of this context whose out_argb range lies inside a registered range (and whose out_rgb_f32 is NULL) takes
 * the direct path.  Same pixels, same blocking semantics.  The caller must not free the memory while it is registered.
 * One range per context.  In a multi-device context every device writes its bands into the range over its own PCIe
 * link (RT_FLAG_STAGED_GATHER keeps the copy engines).                                                             */
int rt_register_output(rt_ctx* ctx, void* host, size_t bytes);
int rt_unregister_output(rt_ctx* ctx);

/* Diagnostic, wave kernel (n <= 64), contexts created with UOB_RT_TIMELINE=1 in the environment: how the persistent
 * waves of the most recent frame spent the kernel's duration, from the 100 MHz s_memrealtime clock.
 *   out[0] waves   out[1] first wave start   out[2] last wave end   out[3] sum of starts   out[4] sum of ends
 *   out[5] jobs done by all waves   out[6] most jobs done by one wave   out[7] length of the expensive-job list the frame started from
 * (start = a wave's first request for a job, end = its exit; mean idle tail = out[2] - out[4]/out[0]).
 * RT_E_UNSUPPORTED when the context was not created with the knob set or its last frame ran on another kernel. */
int rt_debug_wave_timeline(rt_ctx* ctx, uint64_t out[8]);

/* Diagnostic, several devices in one context (rt_config.devices): the copies that carry device k's packed bands
 * (bands k, k+N, ... of device_band_rows rows, owned rows packed top to bottom in its stripe) to image order in a
 * destination frame of `width` x `height` elements of elem_bytes — exactly what rt_render / rt_render_device enqueue
 * for a device that does not write the destination itself.  Pure arithmetic, no device needed: the cross-device branches
 * cannot run on a one-GPU machine, so tests check and replay the plan on the CPU (tests/test_band_copy_plan.py).
 *   dev_to_dev   : 0 = into host memory (rt_render), 1 = into the root device's memory (rt_render_device)
 *   peer_ok      : the source device may copy 2-D into the root's memory directly (hipDeviceEnablePeerAccess succeeded)
 *   same_device  : the source device IS the root device
 * Writes min(count, cap) entries, returns the count (or a negative RT_E_* code).  Offsets and pitches are in bytes.     */
enum { RT_COPY_2D = 0,      /* hipMemcpy2DAsync: `rows` rows of width_bytes, source pitch src_pitch, destination pitch dst_pitch */
       RT_COPY_PEER = 1,    /* hipMemcpyPeerAsync of width_bytes bytes (no peer mapping, or the ragged last band across devices) */
       RT_COPY_LINEAR = 2   /* hipMemcpyAsync of width_bytes bytes (ragged last band, same device or to the host)          */ };
typedef struct rt_band_copy {
  int32_t op, reserved;
  uint64_t dst_offset, dst_pitch, src_offset, src_pitch, width_bytes, rows;
} rt_band_copy;
int rt_debug_band_copy_plan(int32_t num_devices, int32_t k, int32_t device_band_rows, int32_t width, int32_t height,
                            int32_t elem_bytes, int32_t dev_to_dev, int32_t peer_ok, int32_t same_device,
                            rt_band_copy* out, int32_t cap);

/* On-device self test of the exact-reciprocal building block (rt_math.h rcp_newton): sweeps all 2^32
 * FP32 patterns and compares v_rcp_f32 + 1/2 Newton steps with the correctly rounded 1.0f/x.
 * out[0],out[1] = mismatches (1-step, 2-step) for 2^-100 <= |x| <= 2^100; out[2],out[3] = mismatches for
 * the remaining finite non-zero x; out[4] = examples recorded; out[8..63] = mismatching bit patterns.   */
int rt_selftest_rcp(uint64_t out[64]);

/* On-device self test of normalize()'s building blocks (rt_math.h normalize3): (a) v_rsq_f32 refined once against the
 * correctly rounded sqrtf for every FP32 pattern in [2^-60, 2^60]; (b) the quotient q = fma(fma(-b, a r, a), r, a r) from the
 * exact reciprocal r of b against the correctly rounded a / b for EVERY significand of a and every b_stride-th significand
 * of b (b_stride = 1: all 2^46 pairs, about half a minute; tools/div_check.hip is the same sweep as a program).
 * out[0] mismatches of (a), out[1] mismatches of (b), out[2] pairs checked by (b), out[3] / out[4] a mismatching pattern each. */
int rt_selftest_normalize(uint64_t out[8], uint32_t b_stride);

void rt_destroy(rt_ctx* ctx);
/* Message of the last failure on the calling thread.  After an RT_OK from rt_init / rt_render_device of a multi-device
 * context it may instead hold a line that starts with "warning:" — a device without peer access to the root device
 * works, through slower copies (band by band), and says so here.                                                      */
const char* rt_last_error(void);
int rt_abi_version(void);

/* ---- scene format (replaces TestModelH.h / Loader.cpp) -------------------------------------- */
/* Triangle AoS exactly as TestModelH.h:14-18: 5 x vec4 = v0, v1, v2, normal, color (80 bytes).      */
typedef struct rt_triangle {
  float v0[4], v1[4], v2[4], normal[4], color[4];
} rt_triangle;

/* LoadTestModel (TestModelH.h:44-219): the 26-triangle Cornell Box.  Writes up to `cap` triangles,
 * returns the triangle count (26) or a negative error.                                               */
int rt_scene_cornell_box(rt_triangle* out, int32_t cap);
/* load_obj (Loader.cpp:11-59): `v x y z` / `f a b c` lines, scale 1.5, negate, translate
 * (-0.4,1.15,-0.7), colour blue (0,0.2,0.4,0.5); normals are those of the un-negated triangle.
 * Returns the triangle count (may exceed cap; only cap are written) or a negative error.             */
int rt_scene_load_obj(const char* path, rt_triangle* out, int32_t cap);
/* load_obj with the constants of Loader.cpp:20,42,48-52 as arguments: color[4] (w = material: >0 diffuse,
 * 0 mirror, <0 glass), scale, and translate[3] applied after the negation (v' = -(scale*v) + translate).
 * NULL color / translate = the reference's (0,0.2,0.4,0.5) / (-0.4,1.15,-0.7); rt_scene_load_obj(path,...) ==
 * rt_scene_load_obj_ex(path, NULL, 1.5f, NULL, ...).                                                          */
int rt_scene_load_obj_ex(const char* path, const float color[4], float scale, const float translate[3],
                         rt_triangle* out, int32_t cap);
/* ComputeNormal (TestModelH.h:26-35): normal = normalize(cross(v2-v0, v1-v0)), w = 1.               */
void rt_triangle_compute_normal(rt_triangle* t);
/* AoS -> the three packed float4 arrays (skeleton.cpp:474-484).                                      */
void rt_scene_pack(const rt_triangle* tris, int32_t n, float* vertices4, float* normals4, float* colors4);
/* Rotation matrix from yaw/pitch exactly as skeleton.cpp:149-151 (float cos/sin).                    */
void rt_rotation_matrix(float yaw, float pitch, float rot[12]);

#ifdef __cplusplus
}
#endif
#endif /* UOB_RT_H */
