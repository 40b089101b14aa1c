// rt_api.hip — the C-ABI device boundary (include/uob_rt.h) over the gfx950 kernels.
//
// rt_init    replaces opencl_initialise  (Source/skeleton.cpp:366-497): device pick, buffer allocation,
//            one-time blocking upload of the packed scene.
// rt_render  replaces offload_rendering  (Source/skeleton.cpp:146-182): per-frame arguments, the kernel
//            launch that stands where clEnqueueNDRangeKernel(draw) stood (:172), blocking readback (:179).
// There is no CPU fallback: without a HIP device every device entry point fails with RT_E_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_device.h"

namespace uobrt {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

void launch_generic(const FrameParams& P, bool count, hipStream_t stream);
bool generic_needs_records(int n);
void launch_stage_records(const FrameParams& P, hipStream_t stream);
void launch_mesh(const FrameParams& P, bool count, hipStream_t stream);
bool mesh_kernel_supports(const FrameParams& P);
int mesh_tiles(int n);
int mesh_occ_words(int grid);
int mesh_screen_cells(int pixels);
void launch_wave(const FrameParams& P, bool cull, bool count, hipStream_t stream);
void launch_wave_prof(const FrameParams& P, hipStream_t stream);
bool wave_kernel_supports(const FrameParams& P);

}  // namespace uobrt

using namespace uobrt;

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RT_E_DEVICE;                                                               \
    }                                                                                   \
  } while (0)

constexpr int kWorldGrid = 32;       // world cells per axis of the mesh kernel's shadow-ray tile masks

struct rt_ctx {
  rt_config cfg;
  int device = 0;
  int n = 0, n_shadow = 0;
  int owned_rows = 0;
  float4 *d_verts = nullptr, *d_normals = nullptr, *d_colors = nullptr;
  uint32_t* d_argb = nullptr;      // internal framebuffer for rt_render
  float4* d_rgb = nullptr;         // lazily allocated float tap
  unsigned long long* d_counters = nullptr;
  unsigned int* d_jobctr = nullptr; // wave kernel's job queue heads
  int cus = 256;                    // compute units of the device
  // wave kernel: last frame's expensive jobs go first (rt_device.h FrameParams::heavy_*); two lists, used in turn
  unsigned int *d_heavy[2] = {nullptr, nullptr}, *d_heavy_flags = nullptr;
  int heavy_cap = 0, heavy_phase = 0;
  uint32_t heavy_gen = 0;
  float4* d_records = nullptr;     // staged records in HBM for meshes beyond one LDS stage
  // mesh kernel: per-frame candidate-tile masks (rt_kernel_mesh.hip) and the scene's bounding box for its world grid
  unsigned long long *d_screen_masks = nullptr, *d_world_masks = nullptr;
  unsigned int* d_world_occ = nullptr;
  int nwords = 0, scx = 0, scy = 0;
  float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
};

static int validate_config(const rt_config* c) {
  if (!c) { set_error("rt_config is NULL"); return RT_E_INVALID; }
  if (c->width < 1 || c->height < 1 || c->width > 32767 || c->height > 32767) {
    // the reference casts x,y to short (kernels.cl:427)
    set_error("width/height must be in [1, 32767] (got %dx%d)", c->width, c->height); return RT_E_INVALID;
  }
  if (c->aa_x < 1 || c->aa_y < 1 || c->aa_x > 16 || c->aa_y > 16) { set_error("aa_x/aa_y must be in [1,16]"); return RT_E_INVALID; }
  if (c->shadow_samples < 1 || c->shadow_samples > 4096) { set_error("shadow_samples must be in [1,4096]"); return RT_E_INVALID; }
  if (c->max_bounces < 0 || c->max_bounces > 64) { set_error("max_bounces must be in [0,64]"); return RT_E_INVALID; }
  if (c->num_spheres < 0 || c->num_spheres > RT_MAX_SPHERES) { set_error("num_spheres must be in [0,%d]", RT_MAX_SPHERES); return RT_E_INVALID; }
  if (c->band_count < 1 || c->band_index < 0 || c->band_index >= c->band_count || c->band_rows < 1) {
    set_error("band partition invalid (rows=%d index=%d count=%d)", c->band_rows, c->band_index, c->band_count); return RT_E_INVALID;
  }
  if (!(c->light_spread >= 0.0f) || !(c->light_spread <= 1048576.0f)) { set_error("light_spread must be in [0, 2^20]"); return RT_E_INVALID; }
  for (int i = 0; i < c->num_spheres; ++i) {
    const rt_sphere& s = c->spheres[i];
    for (int k = 0; k < 3; ++k)
      if (!(fabsf(s.center[k]) <= 1048576.0f)) { set_error("sphere %d: |centre| must be finite and <= 2^20", i); return RT_E_INVALID; }
    if (!(fabsf(s.radius_sq) <= 1099511627776.0f)) { set_error("sphere %d: radius_sq must be finite and <= 2^40", i); return RT_E_INVALID; }
  }
  if ((double)c->width * c->height > 16777216.0) {
    // global_id = y*W+x is formed in FP32 by the reference (kernels.cl:380): exact only up to 2^24
    set_error("width*height must not exceed 2^24 (the reference's FP32 pixel id)"); return RT_E_INVALID;
  }
  return RT_OK;
}

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_last_error.c_str(); }

void rt_config_default(rt_config* cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof *cfg);
  cfg->width = 1024; cfg->height = 1024;           // skeleton.cpp:32-33
  cfg->aa_x = 2; cfg->aa_y = 2;                    // kernels.cl:12-13
  cfg->shadow_samples = 10; cfg->light_spread = 0.05f;   // kernels.cl:316-317
  cfg->max_bounces = 10;                           // kernels.cl:343
  cfg->num_spheres = 2;                            // kernels.cl:7-10 (third initialiser dropped)
  const rt_sphere glass = {{0.3f, 0.1f, -0.5f}, 0.075f, {0.0f, 0.f, 0.f, -1.0f}};
  const rt_sphere mirror = {{-0.4f, 0.8f, -0.5f}, 0.05f, {0.0f, 0.f, 0.f, 0.0f}};
  cfg->spheres[0] = glass; cfg->spheres[1] = mirror;
  cfg->band_rows = cfg->height; cfg->band_index = 0; cfg->band_count = 1;
  cfg->device = -1; cfg->flags = 0;
}

int32_t rt_config_owned_rows(const rt_config* c) {
  if (!c || c->band_rows < 1 || c->band_count < 1) return 0;
  int rows = 0;
  for (int y = 0; y < c->height; ++y) rows += ((y / c->band_rows) % c->band_count) == c->band_index;
  return rows;
}

int rt_init(const rt_config* cfg, const float* vertices4, const float* normals4, const float* colors4,
            int32_t n, rt_ctx** out_ctx) {
  if (!out_ctx) { set_error("out_ctx is NULL"); return RT_E_INVALID; }
  *out_ctx = nullptr;
  int rc = validate_config(cfg);
  if (rc != RT_OK) return rc;
  if (n < 0 || (n > 0 && (!vertices4 || !normals4 || !colors4))) { set_error("scene arrays missing"); return RT_E_INVALID; }
  // Coordinate bound: keeps every determinant of the intersection tests below 2^126, the range in which
  // the v_rcp_f32 + Newton reciprocal equals IEEE division bit for bit (rt_math.h rcp_exact).
  for (size_t k = 0; k < (size_t)n * 12; ++k) {
    if ((k & 3) != 3 && !(fabsf(vertices4[k]) <= 1048576.0f)) {
      set_error("vertex %zu: coordinates must be finite and |x| <= 2^20", k / 4); return RT_E_INVALID;
    }
  }
  if (cfg->flags & RT_FLAG_FAST_MATH) { set_error("RT_FLAG_FAST_MATH is not built into this library"); return RT_E_UNSUPPORTED; }
  if (n > 4000000) { set_error("triangle list of %d exceeds the supported maximum of 4000000", n); return RT_E_UNSUPPORTED; }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) { set_error("no HIP device present"); return RT_E_DEVICE; }
  rt_ctx* c = new (std::nothrow) rt_ctx();
  if (!c) { set_error("out of host memory"); return RT_E_NOMEM; }
  c->cfg = *cfg;
  if (cfg->device >= 0) { c->device = cfg->device; } else { hipGetDevice(&c->device); }
  auto fail = [&](int code) { rt_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", c->device); return fail(RT_E_DEVICE); }
  c->n = n;
  c->owned_rows = rt_config_owned_rows(cfg);
  if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || c->cus < 1) c->cus = 256;
  const size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(float4);
  const size_t px = (size_t)(c->owned_rows > 0 ? c->owned_rows : 1) * cfg->width;
  if (hipMalloc(&c->d_verts, 3 * nb) != hipSuccess || hipMalloc(&c->d_normals, nb) != hipSuccess ||
      hipMalloc(&c->d_colors, nb) != hipSuccess || hipMalloc(&c->d_argb, px * 4) != hipSuccess ||
      hipMalloc(&c->d_counters, sizeof(rt_work)) != hipSuccess || hipMalloc(&c->d_jobctr, (2 * kJobHeads + 2) * kJobHeadStride * sizeof(unsigned int)) != hipSuccess) {
    set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_NOMEM);
  }
  if (n > 64 && hipMalloc(&c->d_records, (size_t)n * 8 * sizeof(float4)) != hipSuccess) {
    set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_NOMEM);
  }
  // candidate-tile masks: from 17 tiles on (with fewer, building and reading them costs more than the visits they save)
  if (n > 16 * 64 && !(cfg->flags & (RT_FLAG_NO_TILE_BINS | RT_FLAG_NO_CULL | RT_FLAG_GENERIC_KERNEL))) {
    c->nwords = (mesh_tiles(n) + 63) / 64;
    c->scx = mesh_screen_cells(cfg->width); c->scy = mesh_screen_cells(cfg->height);
    const size_t g3 = (size_t)kWorldGrid * kWorldGrid * kWorldGrid;
    if (hipMalloc(&c->d_screen_masks, (size_t)c->scx * c->scy * c->nwords * 8) != hipSuccess ||
        hipMalloc(&c->d_world_masks, g3 * c->nwords * 8) != hipSuccess ||
        hipMalloc(&c->d_world_occ, (size_t)mesh_occ_words(kWorldGrid) * sizeof(unsigned int)) != hipSuccess) {
      set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_NOMEM);
    }
    // every surface point lies on a triangle or a sphere: their bounding box (the world grid spans it)
    for (int k = 0; k < 3; ++k) { c->box_lo[k] = 3.0e38f; c->box_hi[k] = -3.0e38f; }
    for (size_t v = 0; v < (size_t)n * 3; ++v)
      for (int k = 0; k < 3; ++k) {
        c->box_lo[k] = fminf(c->box_lo[k], vertices4[4 * v + k]);
        c->box_hi[k] = fmaxf(c->box_hi[k], vertices4[4 * v + k]);
      }
    for (int i = 0; i < cfg->num_spheres; ++i) {
      const float r = sqrtf(fmaxf(cfg->spheres[i].radius_sq, 0.0f)) * 1.0001f + 1e-6f;
      for (int k = 0; k < 3; ++k) {
        c->box_lo[k] = fminf(c->box_lo[k], cfg->spheres[i].center[k] - r);
        c->box_hi[k] = fmaxf(c->box_hi[k], cfg->spheres[i].center[k] + r);
      }
    }
  }
  {   // wave kernel: lists of last frame's expensive jobs (sized for the smallest job, one 64-ray task)
    const int aa = cfg->aa_x * cfg->aa_y;
    const int pt = (aa >= 1 && aa <= 64) ? 64 / aa : 64;
    const size_t jobs_max = (size_t)((cfg->width + pt - 1) / pt) * (size_t)(c->owned_rows > 0 ? c->owned_rows : 1);
    c->heavy_cap = (int)(jobs_max / 8 > 64 ? jobs_max / 8 : 64);
    if (hipMemset(c->d_jobctr, 0, (2 * kJobHeads + 2) * kJobHeadStride * sizeof(unsigned int)) != hipSuccess) {
      set_error("hipMemset failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_DEVICE);
    }
    if (n >= 1 && n <= 64) {
      if (hipMalloc(&c->d_heavy[0], (size_t)c->heavy_cap * 4) != hipSuccess || hipMalloc(&c->d_heavy[1], (size_t)c->heavy_cap * 4) != hipSuccess ||
          hipMalloc(&c->d_heavy_flags, jobs_max * 4) != hipSuccess || hipMemset(c->d_heavy_flags, 0, jobs_max * 4) != hipSuccess) {
        set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_NOMEM);
      }
    }
  }
  if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    set_error("stream/event creation failed"); return fail(RT_E_DEVICE);
  }
  if (n > 0) {   // blocking uploads, as the CL_TRUE writes at skeleton.cpp:486-496
    if (hipMemcpy(c->d_verts, vertices4, 3 * nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_normals, normals4, nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_colors, colors4, nb, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("scene upload failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_DEVICE);
    }
  }
  c->n_shadow = 0;
  for (int i = 0; i < n; ++i) c->n_shadow += (colors4[4 * i + 3] != -1.0f);
  *out_ctx = c;
  return RT_OK;
}

static void fill_params(const rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
                        FrameParams* P) {
  memset(P, 0, sizeof *P);
  memcpy(P->rot, rot, 12 * sizeof(float));
  memcpy(P->cam, cam, 3 * sizeof(float));
  memcpy(P->light, light, 3 * sizeof(float));
  P->focal = focal;
  const rt_config& g = c->cfg;
  P->spread = g.light_spread;
  P->W = g.width; P->H = g.height; P->aa_x = g.aa_x; P->aa_y = g.aa_y;
  P->S = g.shadow_samples; P->bounces = g.max_bounces; P->nsph = g.num_spheres; P->n = c->n;
  P->band_rows = g.band_rows; P->band_index = g.band_index; P->band_count = g.band_count;
  P->owned_rows = c->owned_rows;
  P->sy = (float)g.aa_x / (float)g.aa_y;
  P->n_shadow = c->n_shadow;
  for (int i = 0; i < g.num_spheres; ++i) {
    P->sph[i].cx = g.spheres[i].center[0]; P->sph[i].cy = g.spheres[i].center[1]; P->sph[i].cz = g.spheres[i].center[2];
    P->sph[i].r2 = g.spheres[i].radius_sq;
    memcpy(P->sph[i].col, g.spheres[i].color, 16);
  }
  P->verts = c->d_verts; P->normals = c->d_normals; P->colors = c->d_colors;
  P->records = c->d_records;
  P->job_counter = c->d_jobctr + kJobHeadStride;      // [HeavyState 0 | queue heads | HeavyState 1], one line each
  {   // wave kernel: a job is a run of job_tasks 64-ray tasks (job_tasks * 64/aa pixels) of one row
    const int aa = g.aa_x * g.aa_y;
    const bool wave_aa = aa >= 1 && aa <= 64;
    // Job size: up to 64 pixels, halved while the queue would hold fewer than ~16 jobs per resident wave (jobs differ
    // 10x in cost, but every hand-out stalls its wave for microseconds; measured with last frame's expensive jobs
    // going first: 4096 and 2048 rows -> 64 px, 1024 and 512 rows -> 32 px), but not below 16 pixels.
    const int pt = wave_aa ? 64 / aa : 64;              // pixels per 64-ray task
    P->aa_magic = wave_aa ? (65536 + aa - 1) / aa : 65536;
    const long waves = (long)c->cus * (g.band_count > 1 ? 4 : 5) * 4;
    int jt = wave_aa ? 64 / pt : 1;                      // tasks of a 64-pixel job (aa for the power-of-two grids)
    while (jt > 1 && ((jt + 1) / 2) * pt >= 16 && (long)((g.width + jt * pt - 1) / (jt * pt)) * c->owned_rows < 16 * waves) jt = (jt + 1) / 2;
    if (const char* e = getenv("UOB_RT_JOB_TASKS")) { const int v = atoi(e); if (wave_aa && v >= 1 && v * pt <= 64) jt = v; }
    P->job_tasks = jt;
    const int job_pixels = jt * pt;
    P->nseg = (g.width + job_pixels - 1) / job_pixels;
    P->njobs = P->nseg * c->owned_rows;
  }
  if (c->d_screen_masks) {
    P->screen_masks = c->d_screen_masks; P->world_masks = c->d_world_masks; P->world_occ = c->d_world_occ;
    P->nwords = c->nwords; P->scx = c->scx; P->scy = c->scy; P->grid_g = kWorldGrid;
    // World grid: a cube over the scene box, grown so that every shadow-ray start point X + 1e-4 (light - X)
    // of a surface point X in the box (kernels.cl:324) stays inside, rounding included.
    float ext = 0.0f, dmax = 0.0f, amax = 0.0f;
    for (int k = 0; k < 3; ++k) {
      ext = fmaxf(ext, c->box_hi[k] - c->box_lo[k]);
      dmax = fmaxf(dmax, fmaxf(fabsf(light[k] - c->box_lo[k]), fabsf(light[k] - c->box_hi[k])));
      amax = fmaxf(amax, fmaxf(fabsf(c->box_lo[k]), fabsf(c->box_hi[k])));
    }
    const float grow = 2e-4f * dmax + 1e-4f * (1.0f + amax) + 1e-3f * ext;
    for (int k = 0; k < 3; ++k) P->grid_lo[k] = c->box_lo[k] - grow;
    P->grid_cell = (ext + 2.0f * grow) / (float)kWorldGrid;
    P->grid_inv = 1.0f / P->grid_cell;
  }
}

static int launch_frame(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
                        uint32_t* d_argb, float4* d_rgb, hipStream_t stream) {
  if (!c || !rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
  for (int k = 0; k < 3; ++k)
    if (!(fabsf(cam[k]) <= 1048576.0f) || !(fabsf(light[k]) <= 1048576.0f)) {
      set_error("camera / light coordinates must be finite and <= 2^20"); return RT_E_INVALID;
    }
  if (c->owned_rows == 0) return RT_OK;
  FrameParams P;
  fill_params(c, rot, cam, light, focal, &P);
  P.out_argb = d_argb; P.out_rgb = d_rgb; P.counters = nullptr;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventRecord(c->ev0, stream));
  const bool wave_paths = !(c->cfg.flags & RT_FLAG_GENERIC_KERNEL);
  if (wave_paths && wave_kernel_supports(P)) {
    // last frame's expensive jobs first — where jobs are long enough (4+ tasks) for the extra look-up per
    // hand-out not to matter (measured: 1024^2 frames with 16-pixel jobs lose 12-18 % to it, larger ones gain 2-8 %)
    if (c->d_heavy_flags && P.job_tasks >= 4 && !getenv("UOB_RT_PLAIN_ORDER")) {
      const int prev = c->heavy_phase, cur = prev ^ 1;
      unsigned int* const st[2] = {c->d_jobctr, c->d_jobctr + (2 * kJobHeads + 1) * kJobHeadStride};
      P.heavy_prev = c->d_heavy[prev]; P.heavy_prev_state = st[prev];
      P.heavy_new = c->d_heavy[cur]; P.heavy_new_state = st[cur];
      P.heavy_flags = c->d_heavy_flags; P.heavy_gen = ++c->heavy_gen;
      P.heavy_factor4 = 8;                                          // expensive = more than twice the average job
      if (const char* e = getenv("UOB_RT_HEAVY_FACTOR4")) { const int v = atoi(e); if (v >= 1 && v <= 4096) P.heavy_factor4 = v; }
      P.heavy_cap = P.njobs / 8 < c->heavy_cap ? P.njobs / 8 : c->heavy_cap;
      c->heavy_phase = cur;
    }
    launch_wave(P, !(c->cfg.flags & RT_FLAG_NO_CULL), false, stream);
  } else if (wave_paths && !(c->cfg.flags & RT_FLAG_NO_CULL) && mesh_kernel_supports(P)) {
    launch_stage_records(P, stream);        // per frame: the records hold camera-dependent terms
    launch_mesh(P, false, stream);
  } else {
    launch_generic(P, false, stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, stream));
  c->timed = true;
  return RT_OK;
}

int rt_render_device(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
                     void* d_out_argb, void* d_out_rgb_f32, void* hip_stream) {
  if (!c || !d_out_argb) { set_error("NULL argument"); return RT_E_INVALID; }
  return launch_frame(c, rot, cam, light, focal, (uint32_t*)d_out_argb, (float4*)d_out_rgb_f32, (hipStream_t)hip_stream);
}

int rt_render(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
              uint32_t* out_argb, float* out_rgb_f32) {
  if (!c || !out_argb) { set_error("NULL argument"); return RT_E_INVALID; }
  const size_t px = (size_t)c->owned_rows * c->cfg.width;
  if (out_rgb_f32 && !c->d_rgb) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&c->d_rgb, (px ? px : 1) * sizeof(float4)));
  }
  int rc = launch_frame(c, rot, cam, light, focal, c->d_argb, out_rgb_f32 ? c->d_rgb : nullptr, c->stream);
  if (rc != RT_OK) return rc;
  if (px == 0) return RT_OK;
  // blocking readback, as clEnqueueReadBuffer(CL_TRUE) at skeleton.cpp:179
  HIP_TRY(hipMemcpyAsync(out_argb, c->d_argb, px * 4, hipMemcpyDeviceToHost, c->stream));
  if (out_rgb_f32) HIP_TRY(hipMemcpyAsync(out_rgb_f32, c->d_rgb, px * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_count_work(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal, rt_work* out) {
  if (!c || !out || !rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
  memset(out, 0, sizeof *out);
  if (c->owned_rows == 0) return RT_OK;
  FrameParams P;
  fill_params(c, rot, cam, light, focal, &P);
  P.counters = c->d_counters;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(rt_work), c->stream));
  launch_generic(P, true, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->d_counters, sizeof(rt_work), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_count_executed(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal, uint64_t out[8]) {
  if (!c || !out || !rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
  memset(out, 0, 8 * sizeof(uint64_t));
  if (c->owned_rows == 0) return RT_OK;
  FrameParams P;
  fill_params(c, rot, cam, light, focal, &P);
  const bool generic = (c->cfg.flags & RT_FLAG_GENERIC_KERNEL) != 0;
  const bool mesh = !generic && !wave_kernel_supports(P) && !(c->cfg.flags & RT_FLAG_NO_CULL) && mesh_kernel_supports(P);
  if (generic || (!wave_kernel_supports(P) && !mesh) || (!mesh && P.S > 64)) {
    set_error("rt_count_executed: this configuration runs on the generic kernel, whose executed work is rt_count_work");
    return RT_E_UNSUPPORTED;
  }
  P.counters = c->d_counters;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(rt_work), c->stream));
  if (mesh) { launch_stage_records(P, c->stream); launch_mesh(P, true, c->stream); }
  else if (getenv("UOB_RT_PHASE_PROFILE")) launch_wave_prof(P, c->stream);   // diagnostic: s_memtime per phase
  else launch_wave(P, !(c->cfg.flags & RT_FLAG_NO_CULL), true, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->d_counters, sizeof(rt_work), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_last_kernel_ms(rt_ctx* c, float* out_ms) {
  if (!c || !out_ms) { set_error("NULL argument"); return RT_E_INVALID; }
  if (!c->timed) { set_error("no frame has been rendered on this context"); return RT_E_INVALID; }
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(out_ms, c->ev0, c->ev1));
  return RT_OK;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  hipFree(c->d_verts); hipFree(c->d_normals); hipFree(c->d_colors);
  hipFree(c->d_argb); hipFree(c->d_rgb); hipFree(c->d_counters); hipFree(c->d_records); hipFree(c->d_jobctr);
  hipFree(c->d_screen_masks); hipFree(c->d_world_masks); hipFree(c->d_world_occ);
  hipFree(c->d_heavy[0]); hipFree(c->d_heavy[1]); hipFree(c->d_heavy_flags);
  delete c;
}

}  // extern "C"
