// rt_api.hip — the C-ABI device boundary (include/uob_rt.h) over the gfx950 kernels.
//
// rt_init    replaces opencl_initialise  (Source/skeleton.cpp:366-497): device pick, buffer allocation,
//            one-time blocking upload of the packed scene.
// rt_render  replaces offload_rendering  (Source/skeleton.cpp:146-182): per-frame arguments, the kernel
//            launch that stands where clEnqueueNDRangeKernel(draw) stood (:172), blocking readback (:179).
// There is no CPU fallback: without a HIP device every device entry point fails with RT_E_DEVICE.
//
// Several GPUs in one context (rt_config.num_devices > 1; SURVEY.md 8(b) "Threading", section 5): the context
// owns one child context per listed device, each with its own stream, scene copy and stripe; a frame is launched
// on all of them from the one host thread, and the bands are delivered by the copy engines — straight into the
// caller's host framebuffer over each device's own PCIe link (rt_render), or into the caller's device buffer
// over the direct xGMI link to that device (rt_render_device).  A gather to one root over point-to-point links
// IS N-1 independent peer copies; they occupy no compute unit, so they run beside the next frame's persistent
// grid.  (The one-process-per-GPU flow of bench.py gathers with RCCL through torch.distributed instead.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
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
void launch_trace_rays(const FrameParams& P, int what, const float* d_rays, const float* d_r2, long nray, int* d_tri,
                       float* d_out10, hipStream_t stream);
void launch_mesh(const FrameParams& P, bool count, bool prof, hipStream_t stream, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join);
bool mesh_kernel_supports(const FrameParams& P);
int mesh_tiles(int n);
int mesh_occ_words(int grid);
int mesh_screen_cells(int pixels);
void launch_wave(const FrameParams& P, bool cull, bool count, hipStream_t stream);
void launch_wave_prof(const FrameParams& P, hipStream_t stream);
bool wave_kernel_supports(const FrameParams& P);
int wave_blocks_per_cu(bool leave_room);
int mesh_blocks_per_cu();

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

// Tuning knobs, read from the environment ONCE per context (rt_init); 0 / false = the built-in choice
struct Tuning {
  int job_tasks = 0;          // UOB_RT_JOB_TASKS: 64-ray tasks per job of the wave kernel
  int heavy_factor4 = 8;      // UOB_RT_HEAVY_FACTOR4: a job is expensive above this / 4 times the average cost
  bool plain_order = false;   // RT_FLAG_PLAIN_ORDER or UOB_RT_PLAIN_ORDER
  bool full_grid = false;     // UOB_RT_FULL_GRID: a rank of a multi-GPU job fills every wave slot too
  float l1_inflate = 3.5f;    // UOB_RT_L1_INFLATE: width of the point set level 1 bounds, in units of the task's own spread (1 .. 64)
  bool heavy_dilate = true;   // UOB_RT_HEAVY_DILATE=0: expensive jobs are listed without their row neighbours
  bool no_specialise = false; // UOB_RT_NO_SPECIALISE: the generic wave-kernel instantiation also where a specialised one exists
  int grid_per_cu = 0;        // UOB_RT_GRID_PER_CU: workgroups per CU of the wave kernel's persistent grid (experiments)
  bool phase_profile = false; // UOB_RT_PHASE_PROFILE: rt_count_executed returns s_memtime shares per phase
  int split_listed = 0;       // UOB_RT_SPLIT_LISTED=1: last frame's expensive jobs are handed out one task at a time
  bool timeline = false;      // UOB_RT_TIMELINE: the wave kernel records when its waves start and end (rt_debug_wave_timeline)
  int mask_debug = 0;         // UOB_RT_MASK_DEBUG: mesh kernel, switch single tile-mask stages off (fault isolation)
};

struct rt_ctx {
  rt_config cfg;
  Tuning tune;
  int device = 0;
  int n = 0, n_shadow = 0;
  int owned_rows = 0;
  float4 *d_verts = nullptr, *d_normals = nullptr, *d_colors = nullptr;
  uint32_t* d_argb = nullptr;      // internal framebuffer (stripe) for rt_render
  float4* d_rgb = nullptr;         // lazily allocated float tap
  unsigned long long* d_counters = nullptr;
  unsigned int* d_jobctr = nullptr; // wave kernel's job queue heads
  int cus = 256;                    // compute units of the device
  // wave kernel: last frame's expensive jobs go first (rt_device.h FrameParams::heavy_*); two lists, used in turn
  unsigned int *d_heavy[2] = {nullptr, nullptr}, *d_heavy_flags = nullptr;
  int heavy_cap = 0, heavy_phase = 0;
  size_t heavy_jobs_max = 0;       // entries of each of the two per-job flag arrays in d_heavy_flags
  // rt_register_output: a host range the device writes frames into directly
  char* reg_host = nullptr; char* reg_dev = nullptr; size_t reg_bytes = 0;
  bool reg_owner = false;        // this context called hipHostRegister (a child of a multi-device context only holds its device's alias)
  bool timeline_valid = false;   // the last frame left one (start, end, jobs) record per wave in d_timeline
  uint64_t* d_timeline = nullptr;
  size_t timeline_waves = 0;
  uint32_t heavy_gen = 0;
  float4* d_records = nullptr;     // staged records in HBM for meshes beyond one LDS stage
  // mesh kernel (n > 64): the scene once more, reordered so that every 64-triangle tile is spatially compact (large
  // triangles first, then Morton order of the centroids), the original index of each triangle, and the tiles' boxes
  float4 *d_verts_m = nullptr, *d_normals_m = nullptr, *d_colors_m = nullptr, *d_tile_box = nullptr;
  int* d_orig = nullptr;
  DevSphere* d_spheres = nullptr;  // the sphere table in device memory (the wave-mapped kernels stage it into LDS)
  unsigned int *d_mesh_cost = nullptr, *d_mesh_order = nullptr;   // per 16x16-pixel block: last frame's cost, this frame's order
  bool mesh_order_valid = false;
  // mesh kernel: per-frame candidate-tile masks (rt_kernel_mesh.hip) and the scene's bounding box for its world grid
  unsigned long long *d_screen_masks = nullptr, *d_world_masks = nullptr;
  unsigned int* d_world_occ = nullptr;
  int nwords = 0, scx = 0, scy = 0;
  float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;             // mesh kernel: the primary-ray masks are built beside the shadow-ray masks
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  hipStream_t last_stream = nullptr;
  // several devices: one child context per entry of cfg.devices (then this context owns no device memory)
  std::vector<rt_ctx*> kids;
  hipEvent_t ev_go = nullptr;       // parent: "the caller's stream has reached this frame"
  hipEvent_t ev_done = nullptr;     // child: "this device's bands have been delivered"
  bool peer_ok = true;              // child: its device can copy 2-D into the destination device directly
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
  if (!(c->light_spread >= 0.0f) || !(c->light_spread <= kMaxCoordinate)) { set_error("light_spread must be in [0, 2^16]"); return RT_E_INVALID; }
  for (int i = 0; i < c->num_spheres; ++i) {
    const rt_sphere& s = c->spheres[i];
    for (int k = 0; k < 3; ++k)
      if (!(fabsf(s.center[k]) <= kMaxCoordinate)) { set_error("sphere %d: |centre| must be finite and <= 2^16", i); return RT_E_INVALID; }
    if (!(fabsf(s.radius_sq) <= kMaxCoordinate * kMaxCoordinate)) { set_error("sphere %d: radius_sq must be finite and <= 2^32", i); return RT_E_INVALID; }
  }
  if ((double)c->width * c->height > 16777216.0) {
    // global_id = y*W+x is formed in FP32 by the reference (kernels.cl:380): exact only up to 2^24
    set_error("width*height must not exceed 2^24 (the reference's FP32 pixel id)"); return RT_E_INVALID;
  }
  if (c->flags & 1) { set_error("flag bit 0 (the former RT_FLAG_FAST_MATH) is not defined in ABI %d", RT_ABI_VERSION); return RT_E_UNSUPPORTED; }
  if (c->num_devices < 0 || c->num_devices > RT_MAX_DEVICES || c->device_band_rows < 0) {
    set_error("num_devices must be in [0,%d] and device_band_rows >= 0", RT_MAX_DEVICES); return RT_E_INVALID;
  }
  if (c->num_devices > 1 && c->band_count != 1) {
    set_error("several devices in one context need the whole frame (band_count == 1)"); return RT_E_UNSUPPORTED;
  }
  return RT_OK;
}

static Tuning read_tuning(const rt_config& cfg) {
  Tuning t;
  if (const char* e = getenv("UOB_RT_JOB_TASKS")) t.job_tasks = atoi(e);
  if (const char* e = getenv("UOB_RT_HEAVY_FACTOR4")) { const int v = atoi(e); if (v >= 1 && v <= 4096) t.heavy_factor4 = v; }
  t.plain_order = (cfg.flags & RT_FLAG_PLAIN_ORDER) != 0 || getenv("UOB_RT_PLAIN_ORDER") != nullptr;
  t.full_grid = getenv("UOB_RT_FULL_GRID") != nullptr;
  t.no_specialise = getenv("UOB_RT_NO_SPECIALISE") != nullptr;
  if (const char* e = getenv("UOB_RT_HEAVY_DILATE")) t.heavy_dilate = atoi(e) != 0;
  if (const char* e = getenv("UOB_RT_L1_INFLATE")) { const float v = (float)atof(e); if (v >= 1.0f && v <= 64.0f) t.l1_inflate = v; }
  if (const char* e = getenv("UOB_RT_GRID_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) t.grid_per_cu = v; }
  t.phase_profile = getenv("UOB_RT_PHASE_PROFILE") != nullptr;
  t.timeline = getenv("UOB_RT_TIMELINE") != nullptr;
  if (const char* e = getenv("UOB_RT_SPLIT_LISTED")) { const int v = atoi(e); if (v == 0 || v == 1) t.split_listed = v; }
  if (const char* e = getenv("UOB_RT_MASK_DEBUG")) t.mask_debug = atoi(e);
  return t;
}

// Keeps the calling thread's current device unchanged across an API call (the caller may be a torch process)
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

// 10 bits -> every third bit
static uint32_t spread3(uint32_t v) {
  v &= 1023u;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// The mesh kernel's copy of the scene (rt_kernel_mesh.hip): the triangle ORDER is a free choice there — shadow tests are
// any-hit, and the closest-hit search resolves equal t by the ORIGINAL index (the reference's loop order, kernels.cl:120)
// — so the triangles are sorted into spatially compact tiles of 64: a task's rays then meet few tiles.  Triangles whose
// extent exceeds a quarter of the scene's (walls) come first, the rest in Morton order of their centroids.
static int upload_tiled_scene(rt_ctx* c, const float* v4, const float* n4, const float* c4, int n) {
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (size_t v = 0; v < (size_t)n * 3; ++v)
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], v4[4 * v + k]); hi[k] = fmaxf(hi[k], v4[4 * v + k]); }
  float ext = 0.0f;
  for (int k = 0; k < 3; ++k) ext = fmaxf(ext, hi[k] - lo[k]);
  const float inv = ext > 0.0f ? 1023.0f / ext : 0.0f;
  std::vector<std::pair<uint32_t, int>> key((size_t)n);
  for (int i = 0; i < n; ++i) {
    const float* a = v4 + (size_t)12 * i;
    float tl[3], th[3];
    for (int k = 0; k < 3; ++k) { tl[k] = fminf(fminf(a[k], a[4 + k]), a[8 + k]); th[k] = fmaxf(fmaxf(a[k], a[4 + k]), a[8 + k]); }
    const float te = fmaxf(fmaxf(th[0] - tl[0], th[1] - tl[1]), th[2] - tl[2]);
    uint32_t code = 0u;
    if (!(te > 0.25f * ext)) {
      uint32_t q[3];
      for (int k = 0; k < 3; ++k) {
        const float f = (0.5f * (tl[k] + th[k]) - lo[k]) * inv;
        q[k] = f >= 0.0f ? (f < 1023.0f ? (uint32_t)f : 1023u) : 0u;
      }
      code = 0x40000000u | spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    }
    key[(size_t)i] = std::make_pair(code, i);
  }
  std::stable_sort(key.begin(), key.end(), [](const std::pair<uint32_t, int>& x, const std::pair<uint32_t, int>& y) { return x.first < y.first; });
  // UOB_RT_TILE_ORDER=kd (default): the small triangles are not left in Morton order (runs of 64 along a space-filling curve
  // jump between octants: a quarter of this round's test mesh's tiles had a normal-cone chord above 0.8) but split top-down at
  // the median of the longest axis of their centroids' box, every cut on a tile boundary, until a range is one tile: compact
  // boxes, compact normal cones.  =morton keeps round 2's order (A/B).  The order is a free choice (see above).
  {
    const char* mode = getenv("UOB_RT_TILE_ORDER");
    int nb = 0;
    while (nb < n && key[(size_t)nb].first == 0u) ++nb;                  // the large triangles, in original order
    if (!(mode && !strcmp(mode, "morton")) && n - nb > 64) {
      std::vector<float> cen((size_t)n * 3);
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cen[(size_t)3 * i + k] = (v4[(size_t)12 * i + k] + v4[(size_t)12 * i + 4 + k] + v4[(size_t)12 * i + 8 + k]) * (1.0f / 3.0f);
      std::vector<int> idx((size_t)(n - nb));
      for (int j = nb; j < n; ++j) idx[(size_t)(j - nb)] = key[(size_t)j].second;
      // ranges [b, e) of idx; position p of idx is position nb + p of the tiled order: cuts where (nb + p) % 64 == 0
      std::vector<std::pair<int, int>> stack;
      stack.push_back(std::make_pair(0, n - nb));
      while (!stack.empty()) {
        const int b = stack.back().first, e = stack.back().second;
        stack.pop_back();
        if ((nb + b) / 64 == (nb + e - 1) / 64) continue;               // one tile
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int p = b; p < e; ++p)
          for (int k = 0; k < 3; ++k) { clo[k] = fminf(clo[k], cen[(size_t)3 * idx[(size_t)p] + k]); chi[k] = fmaxf(chi[k], cen[(size_t)3 * idx[(size_t)p] + k]); }
        int ax = 0;
        if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
        if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        // the tile boundary nearest to the middle of the range
        const int first_cut = ((nb + b) / 64 + 1) * 64 - nb, last_cut = ((nb + e - 1) / 64) * 64 - nb;
        int m = ((nb + (b + e) / 2 + 32) / 64) * 64 - nb;
        m = m < first_cut ? first_cut : (m > last_cut ? last_cut : m);
        std::nth_element(idx.begin() + b, idx.begin() + m, idx.begin() + e,
                         [&](int x, int y) { return cen[(size_t)3 * x + ax] < cen[(size_t)3 * y + ax] || (cen[(size_t)3 * x + ax] == cen[(size_t)3 * y + ax] && x < y); });
        stack.push_back(std::make_pair(b, m));
        stack.push_back(std::make_pair(m, e));
      }
      for (int j = nb; j < n; ++j) key[(size_t)j].second = idx[(size_t)(j - nb)];
    }
  }
  const int ntiles = mesh_tiles(n);
  std::vector<float> pv((size_t)n * 12), pn((size_t)n * 4), pc((size_t)n * 4), box((size_t)ntiles * 12);
  std::vector<int> orig((size_t)n);
  for (int t = 0; t < ntiles; ++t) { for (int k = 0; k < 3; ++k) { box[(size_t)12 * t + k] = 3.0e38f; box[(size_t)12 * t + 4 + k] = -3.0e38f; } box[(size_t)12 * t + 3] = box[(size_t)12 * t + 7] = 0.0f; }
  for (int j = 0; j < n; ++j) {
    const int i = key[(size_t)j].second;
    orig[(size_t)j] = i;
    memcpy(&pv[(size_t)12 * j], v4 + (size_t)12 * i, 48);
    memcpy(&pn[(size_t)4 * j], n4 + (size_t)4 * i, 16);
    memcpy(&pc[(size_t)4 * j], c4 + (size_t)4 * i, 16);
    float* b = &box[(size_t)12 * (j / 64)];
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) { b[k] = fminf(b[k], v4[(size_t)12 * i + 4 * v + k]); b[4 + k] = fmaxf(b[4 + k], v4[(size_t)12 * i + 4 * v + k]); }
  }
  // Per tile, for the bounce rays' tile pre-test (rt_kernel_mesh.hip tile_clear_for_bundle), in double from the float vertices:
  //   lo.w  eta   = max over the tile's triangles of max(|e1|, |e2|, |e2 - e1|) / |e1 x e2|   (inverse altitudes)
  //   hi.w  emax  = max edge length
  //   third float4: unit axis of the triangles' normals (signs aligned) | chi = max |n_T - axis|_2 (chord of the normal cone)
  // A tile with a degenerate triangle gets chi = 4: never certified clear, always visited.
  for (int t = 0; t < ntiles; ++t) {
    const int j0 = t * 64, j1 = (j0 + 64 < n) ? j0 + 64 : n;
    double ax[3] = {0, 0, 0}, eta = 0.0, emax = 0.0;
    bool degenerate = false;
    std::vector<double> nn((size_t)(j1 - j0) * 3);
    for (int j = j0; j < j1; ++j) {
      const float* a = &pv[(size_t)12 * j];
      const double e1[3] = {(double)a[4] - a[0], (double)a[5] - a[1], (double)a[6] - a[2]};
      const double e2[3] = {(double)a[8] - a[0], (double)a[9] - a[1], (double)a[10] - a[2]};
      double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
      const double lc = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
      if (!(lc > 1e-30) || !(l1 > 0) || !(l2 > 0) || !(lc >= 1e-9 * l1 * l2)) { degenerate = true; break; }
      const double l3 = sqrt((e2[0] - e1[0]) * (e2[0] - e1[0]) + (e2[1] - e1[1]) * (e2[1] - e1[1]) + (e2[2] - e1[2]) * (e2[2] - e1[2]));
      const double le = fmax(fmax(l1, l2), l3);
      eta = fmax(eta, le / lc);
      emax = fmax(emax, le);
      double* q = &nn[(size_t)(j - j0) * 3];
      for (int k = 0; k < 3; ++k) q[k] = cr[k] / lc;
      if (j > j0 && q[0] * nn[0] + q[1] * nn[1] + q[2] * nn[2] < 0) for (int k = 0; k < 3; ++k) q[k] = -q[k];   // align with the first
      for (int k = 0; k < 3; ++k) ax[k] += q[k];
    }
    float* b = &box[(size_t)12 * t];
    const double la = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double chi = 4.0;
    if (!degenerate && la > 1e-12) {
      for (int k = 0; k < 3; ++k) ax[k] /= la;
      chi = 0.0;
      for (int j = j0; j < j1; ++j) {
        const double* q = &nn[(size_t)(j - j0) * 3];
        const double dx = q[0] - ax[0], dy = q[1] - ax[1], dz = q[2] - ax[2];
        chi = fmax(chi, sqrt(dx * dx + dy * dy + dz * dz));
      }
    } else {
      ax[0] = 1.0; ax[1] = ax[2] = 0.0; eta = 1e30; emax = 1e30;
    }
    b[3] = (float)(eta * 1.0001); b[7] = (float)(emax * 1.0001);
    b[8] = (float)ax[0]; b[9] = (float)ax[1]; b[10] = (float)ax[2]; b[11] = (float)(chi * 1.0001 + 1e-6);
  }
  const size_t nb = (size_t)n * sizeof(float4);
  if (hipMalloc(&c->d_verts_m, 3 * nb) != hipSuccess || hipMalloc(&c->d_normals_m, nb) != hipSuccess ||
      hipMalloc(&c->d_colors_m, nb) != hipSuccess || hipMalloc(&c->d_orig, (size_t)n * sizeof(int)) != hipSuccess ||
      hipMalloc(&c->d_tile_box, (size_t)ntiles * 3 * sizeof(float4)) != hipSuccess) {
    set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return RT_E_NOMEM;
  }
  if (hipMemcpy(c->d_verts_m, pv.data(), 3 * nb, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_normals_m, pn.data(), nb, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_colors_m, pc.data(), nb, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_orig, orig.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_tile_box, box.data(), (size_t)ntiles * 3 * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("scene upload failed: %s", hipGetErrorString(hipGetLastError())); return RT_E_DEVICE;
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
  cfg->num_devices = 0; cfg->device_band_rows = 0;
}

int32_t rt_config_owned_rows(const rt_config* c) {
  if (!c || c->band_rows < 1 || c->band_count < 1) return 0;
  int rows = 0;
  for (int y = 0; y < c->height; ++y) rows += ((y / c->band_rows) % c->band_count) == c->band_index;
  return rows;
}

static int init_parent(const rt_config* cfg, const float* vertices4, const float* normals4, const float* colors4,
                       int32_t n, rt_ctx** out_ctx) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) { set_error("no HIP device present"); return RT_E_DEVICE; }
  for (int d = 0; d < cfg->num_devices; ++d)
    if (cfg->devices[d] < 0 || cfg->devices[d] >= ndev) {
      set_error("devices[%d] = %d, but %d HIP device(s) are present", d, cfg->devices[d], ndev); return RT_E_INVALID;
    }
  rt_ctx* p = new (std::nothrow) rt_ctx();
  if (!p) { set_error("out of host memory"); return RT_E_NOMEM; }
  std::string downgrade;                     // devices that will copy band by band (reported through rt_last_error)
  p->cfg = *cfg;
  p->device = cfg->devices[0];
  p->n = n;
  p->owned_rows = rt_config_owned_rows(cfg);
  const int dbr = cfg->device_band_rows > 0 ? cfg->device_band_rows : 32;
  p->cfg.device_band_rows = dbr;
  for (int d = 0; d < cfg->num_devices; ++d) {
    rt_config kc = *cfg;
    kc.num_devices = 0; kc.device = cfg->devices[d];
    kc.band_rows = dbr; kc.band_index = d; kc.band_count = cfg->num_devices;
    rt_ctx* k = nullptr;
    const int rc = rt_init(&kc, vertices4, normals4, colors4, n, &k);
    if (rc != RT_OK) { rt_destroy(p); return rc; }
    p->kids.push_back(k);
    if (hipSetDevice(k->device) != hipSuccess || hipEventCreateWithFlags(&k->ev_done, hipEventDisableTiming) != hipSuccess) {
      set_error("event creation failed on device %d", k->device); rt_destroy(p); return RT_E_DEVICE;
    }
    if (k->device != p->device) {           // let the copy engines of this device write the root's memory directly
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, k->device, p->device);
      if (e == hipSuccess && can) {
        e = hipDeviceEnablePeerAccess(p->device, 0);
        k->peer_ok = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
      } else {
        k->peer_ok = false;
      }
      (void)hipGetLastError();
      if (!k->peer_ok) {                     // not an error: the bands travel band by band through hipMemcpyPeerAsync — but say so
        char line[160];
        snprintf(line, sizeof line, "%sdevice %d has no peer access to device %d (%s)", downgrade.empty() ? "warning: " : "; ",
                 k->device, p->device, e == hipSuccess ? "hipDeviceCanAccessPeer: no" : hipGetErrorString(e));
        downgrade += line;
      }
    }
  }
  if (hipSetDevice(p->device) != hipSuccess || hipEventCreateWithFlags(&p->ev_go, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreate(&p->stream) != hipSuccess || hipEventCreate(&p->ev0) != hipSuccess || hipEventCreate(&p->ev1) != hipSuccess) {
    set_error("stream/event creation failed on device %d", p->device); rt_destroy(p); return RT_E_DEVICE;
  }
  // RT_OK with a "warning: ..." line in rt_last_error(): the context works, through the slower copies
  if (!downgrade.empty()) set_error("%s: their bands are copied band by band (hipMemcpyPeerAsync)", downgrade.c_str());
  *out_ctx = p;
  return RT_OK;
}

int rt_init(const rt_config* cfg, const float* vertices4, const float* normals4, const float* colors4,
            int32_t n, rt_ctx** out_ctx) {
  if (!out_ctx) { set_error("out_ctx is NULL"); return RT_E_INVALID; }
  *out_ctx = nullptr;
  int rc = validate_config(cfg);
  if (rc != RT_OK) return rc;
  if (n < 0 || (n > 0 && (!vertices4 || !normals4 || !colors4))) { set_error("scene arrays missing"); return RT_E_INVALID; }
  // Coordinate bound: the range over which the exact culls are verified and which keeps every
  // determinant of the intersection tests below 2^126, where the v_rcp_f32 + Newton reciprocal equals IEEE
  // division bit for bit (rt_math.h rcp_exact).
  for (size_t k = 0; k < (size_t)n * 12; ++k) {
    if ((k & 3) != 3 && !(fabsf(vertices4[k]) <= kMaxCoordinate)) {
      set_error("vertex %zu: coordinates must be finite and |x| <= 2^16", k / 4); return RT_E_INVALID;
    }
  }
  if (n > 4000000) { set_error("triangle list of %d exceeds the supported maximum of 4000000", n); return RT_E_UNSUPPORTED; }
  DeviceGuard guard;
  if (cfg->num_devices > 1) return init_parent(cfg, vertices4, normals4, colors4, n, out_ctx);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) { set_error("no HIP device present"); return RT_E_DEVICE; }
  rt_ctx* c = new (std::nothrow) rt_ctx();
  if (!c) { set_error("out of host memory"); return RT_E_NOMEM; }
  c->cfg = *cfg;
  c->tune = read_tuning(*cfg);
  if (cfg->num_devices == 1) c->device = cfg->devices[0];
  else if (cfg->device >= 0) c->device = cfg->device;
  else hipGetDevice(&c->device);
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
  if (n > 64 && hipMalloc(&c->d_records, (size_t)n * kRecordsPerTriangle * sizeof(float4)) != hipSuccess) {
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
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
      set_error("stream/event creation failed"); return fail(RT_E_DEVICE);
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
    const int pt = (aa >= 1 && aa <= 64) ? 64 / aa : 16;        // smallest job: one task; more than 64 AA samples: 16 pixels
    const size_t jobs_max = (size_t)((cfg->width + pt - 1) / pt) * (size_t)(c->owned_rows > 0 ? c->owned_rows : 1);
    c->heavy_cap = (int)(jobs_max / 3 > 64 ? jobs_max / 3 : 64);
    c->heavy_jobs_max = jobs_max;
    if (hipMemset(c->d_jobctr, 0, (2 * kJobHeads + 2) * kJobHeadStride * sizeof(unsigned int)) != hipSuccess) {
      set_error("hipMemset failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_DEVICE);
    }
    if (n >= 1 && n <= 64 && !c->tune.plain_order) {
      if (hipMalloc(&c->d_heavy[0], (size_t)c->heavy_cap * 4) != hipSuccess || hipMalloc(&c->d_heavy[1], (size_t)c->heavy_cap * 4) != hipSuccess ||
          hipMalloc(&c->d_heavy_flags, 2 * jobs_max * 4) != hipSuccess || hipMemset(c->d_heavy_flags, 0, 2 * jobs_max * 4) != hipSuccess) {
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
  {
    DevSphere tab[RT_MAX_SPHERES];
    memset(tab, 0, sizeof tab);
    for (int i = 0; i < cfg->num_spheres; ++i) {
      tab[i].cx = cfg->spheres[i].center[0]; tab[i].cy = cfg->spheres[i].center[1]; tab[i].cz = cfg->spheres[i].center[2];
      tab[i].r2 = cfg->spheres[i].radius_sq;
      memcpy(tab[i].col, cfg->spheres[i].color, 16);
    }
    if (hipMalloc(&c->d_spheres, sizeof tab) != hipSuccess || hipMemcpy(c->d_spheres, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("sphere table upload failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_DEVICE);
    }
  }
  c->n_shadow = 0;
  for (int i = 0; i < n; ++i) c->n_shadow += (colors4[4 * i + 3] != -1.0f);
  if (n > 64 && !(cfg->flags & RT_FLAG_GENERIC_KERNEL)) {
    rc = upload_tiled_scene(c, vertices4, normals4, colors4, n);
    if (rc != RT_OK) return fail(rc);
    if (!c->tune.plain_order) {
      const size_t jobs = (size_t)((cfg->width + 15) / 16) * (size_t)((c->owned_rows + 15) / 16);
      // order list: up to four entries per block, + its length in the word behind it
      if (hipMalloc(&c->d_mesh_cost, (jobs ? jobs : 1) * 4) != hipSuccess || hipMalloc(&c->d_mesh_order, (4 * (jobs ? jobs : 1) + 1) * 4) != hipSuccess) {
        set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); return fail(RT_E_NOMEM);
      }
    }
  }
  *out_ctx = c;
  return RT_OK;
}

}  // extern "C"

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
  {   // frame invariants of the reference's arithmetic (rt_device.h), same FP32 operations as the kernels would perform
    P->half_wx = ((float)g.width * (float)g.aa_x) / 2.0f;
    P->half_hy = ((float)g.height * (float)g.aa_y) / 2.0f;
    P->w_f = (float)g.width;
    P->focal0 = focal + 0.0f;
    P->rzf[0] = rot[2] * P->focal0; P->rzf[1] = rot[6] * P->focal0; P->rzf[2] = rot[10] * P->focal0;
    P->hbox = g.light_spread / 2.f;
    P->light_inf = fmaxf(fmaxf(fabsf(light[0]), fabsf(light[1])), fabsf(light[2]));
    P->band_rows_magic = g.band_rows > 1 ? (uint32_t)((0x100000000ull + (uint64_t)g.band_rows - 1) / (uint64_t)g.band_rows) : 0u;
  }
  {
    const int aa = g.aa_x * g.aa_y;
    P->inv_S = (g.shadow_samples & (g.shadow_samples - 1)) == 0 ? 1.0f / (float)g.shadow_samples : 0.0f;
    P->inv_aa = (aa & (aa - 1)) == 0 ? 1.0f / (float)aa : 0.0f;
  }
  P->n_shadow = c->n_shadow;
  for (int i = 0; i < g.num_spheres; ++i) {
    P->sph[i].cx = g.spheres[i].center[0]; P->sph[i].cy = g.spheres[i].center[1]; P->sph[i].cz = g.spheres[i].center[2];
    P->sph[i].r2 = g.spheres[i].radius_sq;
    memcpy(P->sph[i].col, g.spheres[i].color, 16);
  }
  P->verts = c->d_verts; P->normals = c->d_normals; P->colors = c->d_colors;
  P->sph_dev = c->d_spheres;
  P->records = c->d_records;
  P->mask_debug = c->tune.mask_debug;
  P->job_counter = c->d_jobctr + kJobHeadStride;      // [HeavyState 0 | queue heads | HeavyState 1], one line each
  {   // wave kernel: a job is a run of job_tasks 64-ray tasks (job_tasks * 64/aa pixels) of one row
    const int aa = g.aa_x * g.aa_y;
    const bool wave_aa = aa >= 1 && aa <= 64;
    const int big_chunks = aa > 64 ? (aa + 63) / 64 : 0;      // 65..256 AA samples: a pixel is big_chunks tasks (rt_kernel_wave.hip BIGAA)
    // Job size: up to 64 pixels, halved while the queue would hold fewer than ~16 jobs per resident wave (jobs differ
    // 10x in cost, but every hand-out stalls its wave for microseconds; measured with last frame's expensive jobs
    // going first: 4096 and 2048 rows -> 64 px, 1024 and 512 rows -> 32 px), but not below 16 pixels.
    const int pt = wave_aa ? 64 / aa : 64;              // pixels per 64-ray task
    P->aa_magic = wave_aa ? (65536 + aa - 1) / aa : 65536;
    P->aax_magic = (65536 + g.aa_x - 1) / g.aa_x;
    // workgroups the chip holds at once; a rank of a multi-GPU job leaves one slot per CU free (registers and LDS
    // for a workgroup of a collective's kernels), so that the gather of the previous frame can run beside it
    const int per_cu = c->tune.grid_per_cu ? c->tune.grid_per_cu : wave_blocks_per_cu(g.band_count > 1 && !c->tune.full_grid);
    P->wave_blocks = c->cus * per_cu;
    const long waves = (long)P->wave_blocks * 4;
    int jt = wave_aa ? 64 / pt : 1;                      // tasks of a 64-pixel job (aa for the power-of-two grids)
    while (jt > 1 && ((jt + 1) / 2) * pt >= 16 && (long)((g.width + jt * pt - 1) / (jt * pt)) * c->owned_rows < 16 * waves) jt = (jt + 1) / 2;
    // (the knob may not make a job smaller than 16 pixels: div_magic's exactness bound, rt_device.h, is stated for >= 16)
    if (wave_aa && c->tune.job_tasks >= 1 && c->tune.job_tasks * pt <= 64 && (c->tune.job_tasks * pt >= 16 || c->tune.job_tasks >= jt))
      jt = c->tune.job_tasks;
    if (big_chunks) jt = 16 * big_chunks;                      // jobs of 16 pixels
    // job / nseg by one multiply-high (rt_device.h div_magic) is exact while (njobs - 1) * (magic * nseg - 2^32) < 2^32.
    // Every accepted frame with jobs of 16+ pixels satisfies it; a knob that asks for smaller jobs is honoured only as far
    // as the bound still holds (checked here, not assumed): the job is doubled until it does.
    int job_pixels = 0;
    for (;;) {
      job_pixels = big_chunks ? jt / big_chunks : jt * pt;
      if (big_chunks) {
        P->nseg = (g.width + job_pixels - 1) / job_pixels;
        P->njobs = P->nseg * c->owned_rows;
        P->nseg_magic = P->nseg > 1 ? (uint32_t)((0x100000000ull + (uint64_t)P->nseg - 1) / (uint64_t)P->nseg) : 0u;
        break;                                                   // 16-pixel jobs: within div_magic's bound for every accepted frame
      }
      P->nseg = (g.width + job_pixels - 1) / job_pixels;
      P->njobs = P->nseg * c->owned_rows;
      P->nseg_magic = P->nseg > 1 ? (uint32_t)((0x100000000ull + (uint64_t)P->nseg - 1) / (uint64_t)P->nseg) : 0u;
      const uint64_t err = P->nseg_magic ? (uint64_t)P->nseg_magic * (uint64_t)P->nseg - 0x100000000ull : 0ull;
      if ((uint64_t)(P->njobs > 0 ? P->njobs - 1 : 0) * err < 0x100000000ull || 2 * jt * pt > 64) break;
      jt *= 2;
    }
    P->job_tasks = jt;
    // (measured on one rank's 512 rows of the headline frame, whose longest jobs last 0.5 of its 0.57 ms: 0.570 ms with,
    // 0.566 without — the span is set by the work per wave and the ~60 us tail, not by the longest job; off unless asked for)
    P->split_listed = c->tune.split_listed == 1 && jt > 1 && !big_chunks ? 1 : 0;
    P->no_specialise = c->tune.no_specialise ? 1 : 0;
    P->l1_inflate = c->tune.l1_inflate;
    P->job_hx = 0.5f * (float)(job_pixels * g.aa_x - 1);
    P->job_hy = 0.5f * (float)(g.aa_y - 1) * P->sy;
    for (int k = 0; k < 3; ++k)
      P->job_eu[k] = 1.0001f * (fabsf(rot[4 * k]) * P->job_hx + fabsf(rot[4 * k + 1]) * P->job_hy);
  }
  if (c->d_screen_masks) {
    P->screen_masks = c->d_screen_masks; P->world_masks = c->d_world_masks; P->world_occ = c->d_world_occ;
    P->nwords = c->nwords; P->scx = c->scx; P->scy = c->scy; P->grid_g = kWorldGrid;
    // World grid: a cube over the scene box, grown so that every shadow-ray start point X + 1e-4 (light - X)
    // of a surface point X in the box (kernels.cl:324) stays inside, rounding included; X itself is computed
    // from the camera (X = cam + t dir, or v0 + u e1 + v e2), so its rounding scales with the camera's and the
    // scene's coordinates.
    float ext = 0.0f, dmax = 0.0f, amax = 0.0f, cmax = 0.0f;
    for (int k = 0; k < 3; ++k) {
      ext = fmaxf(ext, c->box_hi[k] - c->box_lo[k]);
      dmax = fmaxf(dmax, fmaxf(fabsf(light[k] - c->box_lo[k]), fabsf(light[k] - c->box_hi[k])));
      amax = fmaxf(amax, fmaxf(fabsf(c->box_lo[k]), fabsf(c->box_hi[k])));
      cmax = fmaxf(cmax, fabsf(cam[k]));
    }
    float grow = 2e-4f * dmax + 1e-4f * (amax + cmax) + 1e-3f * ext + 1e-30f;
    // hit points on spheres can lie 2e-3 (|ray origin - centre| + R) off the sphere (rt_bin_occupancy): the grid holds them
    for (int i = 0; i < g.num_spheres; ++i) {
      float l2 = 0.0f;
      for (int k = 0; k < 3; ++k) l2 += (cam[k] - g.spheres[i].center[k]) * (cam[k] - g.spheres[i].center[k]);
      const float far = fmaxf(sqrtf(l2), 1.7321f * (ext + 2.0f * grow));
      grow = fmaxf(grow, 2.5e-3f * (far + sqrtf(fmaxf(g.spheres[i].radius_sq, 0.0f))));
    }
    for (int k = 0; k < 3; ++k) P->grid_lo[k] = c->box_lo[k] - grow;
    P->grid_cell = (ext + 2.0f * grow) / (float)kWorldGrid;
    P->grid_inv = 1.0f / P->grid_cell;
  }
}

// The mesh kernel works on the reordered copy of the scene (upload_tiled_scene)
static void use_tiled_scene(const rt_ctx* c, FrameParams* P) {
  P->verts = c->d_verts_m; P->normals = c->d_normals_m; P->colors = c->d_colors_m;
  P->orig = c->d_orig; P->tile_box = c->d_tile_box;
  P->mesh_blocks = c->cus * mesh_blocks_per_cu();
}

// One frame of a single-device context into d_argb (packed rows, or global rows when out_global) on `stream`
static int launch_frame(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
                        uint32_t* d_argb, float4* d_rgb, hipStream_t stream, bool out_global = false) {
  if (!c || !rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
  for (int k = 0; k < 3; ++k)
    if (!(fabsf(cam[k]) <= kMaxCoordinate) || !(fabsf(light[k]) <= kMaxCoordinate)) {
      set_error("camera / light coordinates must be finite and <= 2^16"); return RT_E_INVALID;
    }
  if (!(fabsf(focal) <= 1.0e9f)) { set_error("focal length must be finite and <= 1e9"); return RT_E_INVALID; }
  for (int k = 0; k < 12; ++k)
    if (!(fabsf(rot[k]) <= 4.0f)) { set_error("rotation matrix entries must be finite and <= 4"); return RT_E_INVALID; }
  if (c->owned_rows == 0) return RT_OK;
  FrameParams P;
  fill_params(c, rot, cam, light, focal, &P);
  P.out_argb = d_argb; P.out_rgb = d_rgb; P.counters = nullptr;
  P.out_global = out_global ? 1 : 0;
  HIP_TRY(hipSetDevice(c->device));
  // one frame of a context at a time: the queue heads, the expensive-job lists and the tile masks are shared
  if (c->timed && stream != c->last_stream) HIP_TRY(hipStreamWaitEvent(stream, c->ev1, 0));
  HIP_TRY(hipEventRecord(c->ev0, stream));
  const bool wave_paths = !(c->cfg.flags & RT_FLAG_GENERIC_KERNEL);
  if (wave_paths && wave_kernel_supports(P) && (P.aa_x * P.aa_y <= 64 || !(c->cfg.flags & RT_FLAG_NO_CULL))) {
    // last frame's expensive jobs first — where jobs are long enough (4+ tasks) for the extra look-up per
    // hand-out not to matter (measured: 1024^2 frames with 16-pixel jobs lose 12-18 % to it, larger ones gain 2-8 %)
    if (c->d_heavy_flags && P.job_tasks >= 4) {
      const int prev = c->heavy_phase, cur = prev ^ 1;
      unsigned int* const st[2] = {c->d_jobctr, c->d_jobctr + (2 * kJobHeads + 1) * kJobHeadStride};
      P.heavy_prev = c->d_heavy[prev]; P.heavy_prev_state = st[prev];
      P.heavy_new = c->d_heavy[cur]; P.heavy_new_state = st[cur];
      P.heavy_flags = c->d_heavy_flags + (size_t)prev * c->heavy_jobs_max; P.heavy_flags_new = c->d_heavy_flags + (size_t)cur * c->heavy_jobs_max;
      P.heavy_gen = ++c->heavy_gen;
      P.heavy_factor4 = c->tune.heavy_factor4;                      // expensive = more than twice the average job
      P.heavy_cap = P.njobs / 3 < c->heavy_cap ? P.njobs / 3 : c->heavy_cap;
      P.heavy_dilate = c->tune.heavy_dilate ? 1 : 0;
      c->heavy_phase = cur;
    }
    c->timeline_valid = false;
    if (c->tune.timeline) {                 // diagnostic: the shipped kernel, with its per-wave start / end stamps
      const size_t waves = (size_t)P.wave_blocks * 4;
      if (!c->d_timeline) {
        if (hipMalloc(&c->d_timeline, waves * 3 * sizeof(uint64_t)) != hipSuccess) { set_error("hipMalloc failed (timeline)"); return RT_E_NOMEM; }
        c->timeline_waves = waves;
      }
      if (c->timeline_waves >= waves) {
        HIP_TRY(hipMemsetAsync(c->d_timeline, 0, c->timeline_waves * 3 * sizeof(uint64_t), stream));
        P.counters = reinterpret_cast<unsigned long long*>(c->d_timeline);
        c->timeline_valid = true;
      }
    }
    launch_wave(P, !(c->cfg.flags & RT_FLAG_NO_CULL), false, stream);
  } else if (wave_paths && !(c->cfg.flags & RT_FLAG_NO_CULL) && mesh_kernel_supports(P)) {
    use_tiled_scene(c, &P);
    if (c->d_mesh_cost) {                   // last frame's expensive blocks first; this frame's costs make the next order
      P.mesh_order = c->mesh_order_valid ? c->d_mesh_order : nullptr;
      P.mesh_cost = c->d_mesh_cost; P.mesh_order_out = c->d_mesh_order;
      P.mesh_queue_len = c->d_mesh_order + 4 * (size_t)((c->cfg.width + 15) / 16) * (size_t)((c->owned_rows + 15) / 16);
      c->mesh_order_valid = true;
    }
    launch_stage_records(P, stream);        // per frame: the records hold camera-dependent terms
    launch_mesh(P, false, false, stream, c->aux_stream, c->ev_fork, c->ev_join);
  } else {
    launch_generic(P, false, stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, stream));
  c->timed = true;
  c->last_stream = stream;
  return RT_OK;
}

// ---- several devices --------------------------------------------------------------------------------------
// Child k of a parent with N children owns the bands k, k+N, ... of `dbr` rows; its stripe holds them packed.
// band_copy_plan lists the copies that put them into image order at the destination (row pitch W elements of `elem`
// bytes): one 2-D copy whose "rows" are whole bands, plus the ragged last band if this child owns it.  The plan is
// pure arithmetic (no HIP call): rt_debug_band_copy_plan exposes it so that a CPU test can check every offset and pitch
// and replay it with memcpy — the cross-device branches cannot run on a one-GPU box.
static int band_copy_plan(int N, int k, int dbr, int W, int owned_rows, size_t elem, bool dev_to_dev, bool peer_ok,
                          bool same_device, rt_band_copy* out, int cap) {
  const size_t band_bytes = (size_t)dbr * W * elem;
  const int full = owned_rows / dbr, rem = owned_rows % dbr;
  const size_t d0 = (size_t)k * band_bytes;
  int cnt = 0;
  auto emit = [&](int op, size_t doff, size_t dpitch, size_t soff, size_t spitch, size_t width, size_t rows) {
    if (cnt < cap && out) {
      rt_band_copy& c = out[cnt];
      c.op = op; c.reserved = 0; c.dst_offset = doff; c.dst_pitch = dpitch; c.src_offset = soff; c.src_pitch = spitch;
      c.width_bytes = width; c.rows = rows;
    }
    ++cnt;
  };
  if (full > 0) {
    if (!dev_to_dev || peer_ok) {
      emit(RT_COPY_2D, d0, (size_t)N * band_bytes, 0, band_bytes, band_bytes, (size_t)full);
    } else {                                   // no peer mapping: band by band through the runtime
      for (int b = 0; b < full; ++b)
        emit(RT_COPY_PEER, d0 + (size_t)b * N * band_bytes, 0, (size_t)b * band_bytes, 0, band_bytes, 1);
    }
  }
  if (rem > 0) {
    const size_t bytes = (size_t)rem * W * elem;
    emit(dev_to_dev && !same_device ? RT_COPY_PEER : RT_COPY_LINEAR, d0 + (size_t)full * N * band_bytes, 0,
         (size_t)full * band_bytes, 0, bytes, 1);
  }
  return cnt;
}

static int deliver_bands(rt_ctx* p, int k, const void* stripe, void* dst, size_t elem, hipMemcpyKind kind, hipStream_t stream) {
  rt_ctx* c = p->kids[k];
  const int N = (int)p->kids.size(), dbr = p->cfg.device_band_rows, W = p->cfg.width;
  const bool d2d = kind == hipMemcpyDeviceToDevice;
  const int full = c->owned_rows / dbr;
  std::vector<rt_band_copy> plan((size_t)full + 2);
  int cnt = band_copy_plan(N, k, dbr, W, c->owned_rows, elem, d2d, c->peer_ok, c->device == p->device, plan.data(), (int)plan.size());
  for (int i = 0; i < cnt; ++i) {
    const rt_band_copy& q = plan[(size_t)i];
    char* const d = (char*)dst + q.dst_offset;
    const char* const sp = (const char*)stripe + q.src_offset;
    if (q.op == RT_COPY_2D) {
      const hipError_t e = hipMemcpy2DAsync(d, q.dst_pitch, sp, q.src_pitch, q.width_bytes, q.rows, kind, stream);
      if (e != hipSuccess && d2d && c->device != p->device) {
        // the direct 2-D copy was refused after all: from now on this device copies band by band through the runtime
        (void)hipGetLastError();
        c->peer_ok = false;
        set_error("warning: device %d cannot copy 2-D into device %d (%s): its bands go band by band through hipMemcpyPeerAsync",
                  c->device, p->device, hipGetErrorString(e));
        cnt = band_copy_plan(N, k, dbr, W, c->owned_rows, elem, d2d, false, false, plan.data(), (int)plan.size());
        i = -1;
        continue;
      }
      if (e != hipSuccess) { set_error("hipMemcpy2DAsync failed: %s", hipGetErrorString(e)); return RT_E_DEVICE; }
    } else if (q.op == RT_COPY_PEER) {
      HIP_TRY(hipMemcpyPeerAsync(d, p->device, sp, c->device, q.width_bytes, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(d, sp, q.width_bytes, kind, stream));
    }
  }
  return RT_OK;
}

static int parent_render(rt_ctx* p, const float rot[12], const float cam[3], const float light[3], float focal,
                         uint32_t* host_argb, float* host_rgb, uint32_t* d_argb, float4* d_rgb, hipStream_t caller) {
  const bool to_host = host_argb != nullptr;
  const size_t W = (size_t)p->cfg.width;
  const size_t nk = p->kids.size();
  if (!to_host) {                                     // the caller's earlier work on the destination comes first
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev0, caller));
    HIP_TRY(hipEventRecord(p->ev_go, caller));
  }
  const bool want_rgb = to_host ? host_rgb != nullptr : d_rgb != nullptr;
  std::vector<char> launched(nk, 0), direct(nk, 0);
  // On an error the devices already launched may still be writing into the caller's buffers: wait for them before returning
  auto fail = [&](int rc) {
    const std::string msg = g_last_error;
    for (size_t k = 0; k < nk; ++k)
      if (launched[k]) { hipSetDevice(p->kids[k]->device); hipStreamSynchronize(p->kids[k]->stream); }
    (void)hipGetLastError();
    g_last_error = msg;
    return rc;
  };
  // pass 1: every device's frame is launched before any band is delivered.  (Delivering inside this loop made the host
  // thread wait for device k's kernel and copy — a device-to-pageable-host copy blocks — before it launched device k+1:
  // the devices then ran one after another.)
  for (size_t k = 0; k < nk; ++k) {
    rt_ctx* c = p->kids[k];
    if (c->owned_rows == 0) continue;
    if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", c->device); return fail(RT_E_DEVICE); }
    if (want_rgb && !c->d_rgb && hipMalloc(&c->d_rgb, (size_t)c->owned_rows * W * sizeof(float4)) != hipSuccess) {
      set_error("hipMalloc failed (float tap, device %d)", c->device); return fail(RT_E_NOMEM);
    }
    if (!to_host && hipStreamWaitEvent(c->stream, p->ev_go, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return fail(RT_E_DEVICE); }
    // a device that holds the destination writes its rows there itself; the others render into their stripe.  A registered
    // host framebuffer (rt_register_output) is held by every device: each writes its bands into it over its own PCIe link.
    const bool mapped = to_host && !host_rgb && c->reg_host && (char*)host_argb >= c->reg_host &&
                        (char*)host_argb + (size_t)p->cfg.height * W * 4 <= c->reg_host + c->reg_bytes &&
                        !(p->cfg.flags & RT_FLAG_STAGED_GATHER);
    direct[k] = mapped || (!to_host && c->device == p->device && !(p->cfg.flags & RT_FLAG_STAGED_GATHER));
    uint32_t* const dst = mapped ? reinterpret_cast<uint32_t*>(c->reg_dev + ((char*)host_argb - c->reg_host)) : d_argb;
    const int rc = direct[k] ? launch_frame(c, rot, cam, light, focal, dst, mapped ? nullptr : d_rgb, c->stream, true)
                             : launch_frame(c, rot, cam, light, focal, c->d_argb, want_rgb ? c->d_rgb : nullptr, c->stream);
    if (rc != RT_OK) return fail(rc);
    launched[k] = 1;
  }
  // pass 2: the copy engines deliver the bands, every device on its own stream behind its own kernel
  for (size_t k = 0; k < nk; ++k) {
    rt_ctx* c = p->kids[k];
    if (!launched[k]) continue;
    if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", c->device); return fail(RT_E_DEVICE); }
    if (!direct[k]) {
      const hipMemcpyKind kind = to_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
      int rc = deliver_bands(p, (int)k, c->d_argb, to_host ? (void*)host_argb : (void*)d_argb, 4, kind, c->stream);
      if (rc == RT_OK && want_rgb)
        rc = deliver_bands(p, (int)k, c->d_rgb, to_host ? (void*)host_rgb : (void*)d_rgb, sizeof(float4), kind, c->stream);
      if (rc != RT_OK) return fail(rc);
    }
    if (!to_host && hipEventRecord(c->ev_done, c->stream) != hipSuccess) { set_error("hipEventRecord failed"); return fail(RT_E_DEVICE); }
  }
  if (to_host) {
    for (rt_ctx* c : p->kids) if (c->owned_rows) { HIP_TRY(hipSetDevice(c->device)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    p->timed = false;             // rt_last_kernel_ms: this frame's time is the children's, not an older device-path interval
  } else {
    HIP_TRY(hipSetDevice(p->device));
    for (rt_ctx* c : p->kids) if (c->owned_rows) HIP_TRY(hipStreamWaitEvent(caller, c->ev_done, 0));
    HIP_TRY(hipEventRecord(p->ev1, caller));
    p->timed = true;
  }
  return RT_OK;
}

extern "C" {

int rt_render_device(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
                     void* d_out_argb, void* d_out_rgb_f32, void* hip_stream) {
  if (!c || !d_out_argb) { set_error("NULL argument"); return RT_E_INVALID; }
  DeviceGuard guard;
  if (!c->kids.empty()) {
    if (!rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
    return parent_render(c, rot, cam, light, focal, nullptr, nullptr, (uint32_t*)d_out_argb, (float4*)d_out_rgb_f32, (hipStream_t)hip_stream);
  }
  return launch_frame(c, rot, cam, light, focal, (uint32_t*)d_out_argb, (float4*)d_out_rgb_f32, (hipStream_t)hip_stream);
}

int rt_render(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal,
              uint32_t* out_argb, float* out_rgb_f32) {
  if (!c || !out_argb) { set_error("NULL argument"); return RT_E_INVALID; }
  DeviceGuard guard;
  if (!c->kids.empty()) {
    if (!rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
    return parent_render(c, rot, cam, light, focal, out_argb, out_rgb_f32, nullptr, nullptr, nullptr);
  }
  const size_t px = (size_t)c->owned_rows * c->cfg.width;
  if (out_rgb_f32 && !c->d_rgb) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&c->d_rgb, (px ? px : 1) * sizeof(float4)));
  }
  if (c->reg_host && !out_rgb_f32 && px != 0 && (char*)out_argb >= c->reg_host &&
      (char*)out_argb + px * 4 <= c->reg_host + c->reg_bytes) {
    // the caller's framebuffer is mapped: the kernel's stores ARE the read-back (rt_register_output)
    uint32_t* const d_out = reinterpret_cast<uint32_t*>(c->reg_dev + ((char*)out_argb - c->reg_host));
    const int rc0 = launch_frame(c, rot, cam, light, focal, d_out, nullptr, c->stream);
    if (rc0 != RT_OK) return rc0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
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

int rt_unregister_output(rt_ctx* c) {
  if (!c) { set_error("NULL argument"); return RT_E_INVALID; }
  if (!c->reg_host || !c->reg_owner) return RT_OK;
  DeviceGuard guard;
  for (rt_ctx* k : c->kids) {                        // no frame of any device may still be writing
    HIP_TRY(hipSetDevice(k->device));
    HIP_TRY(hipStreamSynchronize(k->stream));
    k->reg_host = k->reg_dev = nullptr; k->reg_bytes = 0;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (c->timed) HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipHostUnregister(c->reg_host));
  c->reg_host = c->reg_dev = nullptr; c->reg_bytes = 0; c->reg_owner = false;
  return RT_OK;
}

int rt_register_output(rt_ctx* c, void* host, size_t bytes) {
  if (!c || !host || bytes == 0) { set_error("NULL argument"); return RT_E_INVALID; }
  const int rc = rt_unregister_output(c);
  if (rc != RT_OK) return rc;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  auto alias = [&](rt_ctx* x) -> bool {              // the range as device x->device addresses it
    void* dev = nullptr;
    if (hipSetDevice(x->device) != hipSuccess || hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || !dev) return false;
    x->reg_host = static_cast<char*>(host); x->reg_dev = static_cast<char*>(dev); x->reg_bytes = bytes;
    return true;
  };
  bool ok = alias(c);
  for (rt_ctx* k : c->kids) ok = ok && alias(k);
  if (!ok) {
    for (rt_ctx* k : c->kids) { k->reg_host = k->reg_dev = nullptr; k->reg_bytes = 0; }
    c->reg_host = c->reg_dev = nullptr; c->reg_bytes = 0;
    hipHostUnregister(host);
    set_error("hipHostGetDevicePointer failed: %s", hipGetErrorString(hipGetLastError()));
    return RT_E_DEVICE;
  }
  c->reg_owner = true;
  return RT_OK;
}

int rt_count_work(rt_ctx* c, const float rot[12], const float cam[3], const float light[3], float focal, rt_work* out) {
  if (!c || !out || !rot || !cam || !light) { set_error("NULL argument"); return RT_E_INVALID; }
  memset(out, 0, sizeof *out);
  DeviceGuard guard;
  if (!c->kids.empty()) {
    for (rt_ctx* k : c->kids) {
      rt_work w;
      const int rc = rt_count_work(k, rot, cam, light, focal, &w);
      if (rc != RT_OK) return rc;
      for (size_t q = 0; q < sizeof(rt_work) / 8; ++q) ((uint64_t*)out)[q] += ((const uint64_t*)&w)[q];
    }
    return RT_OK;
  }
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
  DeviceGuard guard;
  if (!c->kids.empty()) {
    for (rt_ctx* k : c->kids) {
      uint64_t w[8];
      const int rc = rt_count_executed(k, rot, cam, light, focal, w);
      if (rc != RT_OK) return rc;
      for (int q = 0; q < 8; ++q) out[q] += w[q];
    }
    return RT_OK;
  }
  if (c->owned_rows == 0) return RT_OK;
  FrameParams P;
  fill_params(c, rot, cam, light, focal, &P);
  const bool generic = (c->cfg.flags & RT_FLAG_GENERIC_KERNEL) != 0;
  const bool mesh = !generic && !wave_kernel_supports(P) && !(c->cfg.flags & RT_FLAG_NO_CULL) && mesh_kernel_supports(P);
  if (generic || (!wave_kernel_supports(P) && !mesh) || (!mesh && (P.S > 64 || P.aa_x * P.aa_y > 64))) {
    set_error("rt_count_executed: this configuration runs on the generic kernel, whose executed work is rt_count_work");
    return RT_E_UNSUPPORTED;
  }
  P.counters = c->d_counters;
  HIP_TRY(hipSetDevice(c->device));
  if (c->timed) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev1, 0));
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(rt_work), c->stream));
  if (mesh) { use_tiled_scene(c, &P); launch_stage_records(P, c->stream); launch_mesh(P, true, c->tune.phase_profile, c->stream, nullptr, nullptr, nullptr); }
  else if (c->tune.phase_profile) launch_wave_prof(P, c->stream);   // diagnostic: s_memtime per phase
  else launch_wave(P, !(c->cfg.flags & RT_FLAG_NO_CULL), true, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->d_counters, sizeof(rt_work), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_debug_wave_timeline(rt_ctx* c, uint64_t out[8]) {
  if (!c || !out) { set_error("NULL argument"); return RT_E_INVALID; }
  if (!c->kids.empty()) c = c->kids[0];
  if (!c->tune.timeline || !c->timeline_valid) {
    set_error("rt_debug_wave_timeline: needs UOB_RT_TIMELINE=1 at rt_init and a frame rendered by the wave kernel");
    return RT_E_UNSUPPORTED;
  }
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(c->device));
  if (c->timed) HIP_TRY(hipEventSynchronize(c->ev1));
  std::vector<uint64_t> rec(c->timeline_waves * 3);
  HIP_TRY(hipMemcpy(rec.data(), c->d_timeline, rec.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  for (int q = 0; q < 8; ++q) out[q] = 0;
  out[1] = ~0ull;
  for (size_t w = 0; w < c->timeline_waves; ++w) {
    const uint64_t t0 = rec[3 * w], t1 = rec[3 * w + 1], jobs = rec[3 * w + 2] & 0xffffffffull;
    if (t1 == 0) continue;                  // a slot no wave of the grid wrote
    out[7] = rec[3 * w + 2] >> 32;
    out[0] += 1; out[3] += t0; out[4] += t1; out[5] += jobs;
    if (t0 < out[1]) out[1] = t0;
    if (t1 > out[2]) out[2] = t1;
    if (jobs > out[6]) out[6] = jobs;
  }
  return RT_OK;
}

int rt_debug_trace_rays(rt_ctx* c, int32_t what, const float* rays6, const float* radius_sq, int64_t nray,
                        int32_t* out_tri, float* out10) {
  if (!c || !rays6 || !out_tri || nray < 0) { set_error("NULL argument"); return RT_E_INVALID; }
  if (what != RT_TRACE_IN_SHADOW && what != RT_TRACE_CLOSEST_HIT) { set_error("rt_debug_trace_rays: unknown mode %d", what); return RT_E_INVALID; }
  if (what == RT_TRACE_IN_SHADOW && !radius_sq) { set_error("rt_debug_trace_rays: radius_sq missing"); return RT_E_INVALID; }
  if (what == RT_TRACE_CLOSEST_HIT && !out10) { set_error("rt_debug_trace_rays: out10 missing"); return RT_E_INVALID; }
  if (!c->kids.empty()) c = c->kids[0];
  if (nray == 0) return RT_OK;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(c->device));
  const float zero3[3] = {0.f, 0.f, 0.f}, ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  FrameParams P;
  fill_params(c, ident, zero3, zero3, 1.0f, &P);
  float *d_rays = nullptr, *d_r2 = nullptr, *d_out = nullptr;
  int* d_tri = nullptr;
  int rc = RT_OK;
  if (hipMalloc(&d_rays, (size_t)nray * 24) != hipSuccess || hipMalloc(&d_tri, (size_t)nray * 4) != hipSuccess ||
      (radius_sq && hipMalloc(&d_r2, (size_t)nray * 4) != hipSuccess) || (out10 && hipMalloc(&d_out, (size_t)nray * 40) != hipSuccess)) {
    set_error("hipMalloc failed: %s", hipGetErrorString(hipGetLastError())); rc = RT_E_NOMEM;
  }
  auto ok = [&](hipError_t e, const char* opn) {
    if (rc == RT_OK && e != hipSuccess) { set_error("%s failed: %s", opn, hipGetErrorString(e)); rc = RT_E_DEVICE; }
  };
  if (rc == RT_OK) {
    if (c->timed) ok(hipStreamWaitEvent(c->stream, c->ev1, 0), "hipStreamWaitEvent");
    ok(hipMemcpyAsync(d_rays, rays6, (size_t)nray * 24, hipMemcpyHostToDevice, c->stream), "ray upload");
    if (radius_sq) ok(hipMemcpyAsync(d_r2, radius_sq, (size_t)nray * 4, hipMemcpyHostToDevice, c->stream), "ray upload");
    if (d_out) ok(hipMemsetAsync(d_out, 0, (size_t)nray * 40, c->stream), "hipMemsetAsync");
    if (rc == RT_OK) {
      if (generic_needs_records(c->n)) launch_stage_records(P, c->stream);
      launch_trace_rays(P, what, d_rays, d_r2, (long)nray, d_tri, d_out, c->stream);
      ok(hipGetLastError(), "trace kernel launch");
    }
    ok(hipMemcpyAsync(out_tri, d_tri, (size_t)nray * 4, hipMemcpyDeviceToHost, c->stream), "read-back");
    if (out10 && what == RT_TRACE_CLOSEST_HIT) ok(hipMemcpyAsync(out10, d_out, (size_t)nray * 40, hipMemcpyDeviceToHost, c->stream), "read-back");
    ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  }
  hipFree(d_rays); hipFree(d_r2); hipFree(d_out); hipFree(d_tri);
  return rc;
}

int rt_debug_band_copy_plan(int32_t num_devices, int32_t k, int32_t device_band_rows, int32_t width, int32_t height,
                            int32_t elem_bytes, int32_t dev_to_dev, int32_t peer_ok, int32_t same_device,
                            rt_band_copy* out, int32_t cap) {
  if (num_devices < 1 || num_devices > RT_MAX_DEVICES || k < 0 || k >= num_devices || device_band_rows < 0 || width < 1 ||
      height < 1 || elem_bytes < 1 || cap < 0 || (!out && cap > 0)) {
    set_error("rt_debug_band_copy_plan: invalid argument"); return RT_E_INVALID;
  }
  const int dbr = device_band_rows > 0 ? device_band_rows : 32;
  rt_config kc;
  memset(&kc, 0, sizeof kc);
  kc.height = height; kc.band_rows = dbr; kc.band_index = k; kc.band_count = num_devices;
  return band_copy_plan(num_devices, k, dbr, width, rt_config_owned_rows(&kc), (size_t)elem_bytes, dev_to_dev != 0, peer_ok != 0,
                        same_device != 0, out, cap);
}

int rt_debug_block_costs(rt_ctx* c, uint32_t* out, int32_t cap) {
  if (!c || (!out && cap > 0) || cap < 0) { set_error("NULL argument"); return RT_E_INVALID; }
  if (!c->kids.empty()) c = c->kids[0];
  if (!c->d_mesh_cost || !c->mesh_order_valid) { set_error("rt_debug_block_costs: this context records no block costs"); return RT_E_UNSUPPORTED; }
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(c->device));
  const int jobs = ((c->cfg.width + 15) / 16) * ((c->owned_rows + 15) / 16);
  if (c->timed) HIP_TRY(hipEventSynchronize(c->ev1));
  // job id -> block: rows are numbered from the middle outwards (rt_kernel_mesh.hip), undo that here
  std::vector<uint32_t> raw((size_t)jobs);
  HIP_TRY(hipMemcpy(raw.data(), c->d_mesh_cost, (size_t)jobs * 4, hipMemcpyDeviceToHost));
  const int wx = (c->cfg.width + 15) / 16, wy = (c->owned_rows + 15) / 16, mid = (wy + 1) >> 1;
  for (int j = 0; j < jobs; ++j) {
    const int jy = j / wx, jx = j - jy * wx;
    const int row = (jy & 1) ? mid + (jy >> 1) : mid - 1 - (jy >> 1);
    const int at = row * wx + jx;
    if (at >= 0 && at < cap) out[at] = raw[(size_t)j];
  }
  return jobs;
}

int rt_debug_world_masks(rt_ctx* c, uint64_t* out, int64_t cap, int32_t* grid, int32_t* words) {
  if (!c || (!out && cap > 0) || cap < 0 || !grid || !words) { set_error("NULL argument"); return RT_E_INVALID; }
  if (!c->kids.empty()) c = c->kids[0];
  if (!c->d_world_masks || c->nwords <= 0) { set_error("rt_debug_world_masks: this context builds no tile masks"); return RT_E_UNSUPPORTED; }
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(c->device));
  if (c->timed) HIP_TRY(hipEventSynchronize(c->ev1));
  const size_t total = (size_t)kWorldGrid * kWorldGrid * kWorldGrid * (size_t)c->nwords;
  *grid = kWorldGrid; *words = c->nwords;
  const size_t take = total < (size_t)cap ? total : (size_t)cap;
  if (take > 0) HIP_TRY(hipMemcpy(out, c->d_world_masks, take * 8, hipMemcpyDeviceToHost));
  return (int)total;
}

int rt_last_kernel_ms(rt_ctx* c, float* out_ms) {
  if (!c || !out_ms) { set_error("NULL argument"); return RT_E_INVALID; }
  DeviceGuard guard;
  if (!c->kids.empty() && !c->timed) {      // after rt_render into host memory: the slowest device's kernel
    float mx = -1.0f;
    for (rt_ctx* k : c->kids) {
      float ms = 0.0f;
      if (k->timed && rt_last_kernel_ms(k, &ms) == RT_OK && ms > mx) mx = ms;
    }
    if (mx < 0.0f) { set_error("no frame has been rendered on this context"); return RT_E_INVALID; }
    *out_ms = mx;
    return RT_OK;
  }
  if (!c->timed) { set_error("no frame has been rendered on this context"); return RT_E_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(out_ms, c->ev0, c->ev1));
  return RT_OK;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  DeviceGuard guard;
  for (rt_ctx* k : c->kids) rt_destroy(k);
  hipSetDevice(c->device);
  if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->aux_stream) { hipStreamSynchronize(c->aux_stream); hipStreamDestroy(c->aux_stream); }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_go) hipEventDestroy(c->ev_go);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  hipFree(c->d_verts); hipFree(c->d_normals); hipFree(c->d_colors);
  hipFree(c->d_argb); hipFree(c->d_rgb); hipFree(c->d_counters); hipFree(c->d_records); hipFree(c->d_jobctr);
  hipFree(c->d_screen_masks); hipFree(c->d_world_masks); hipFree(c->d_world_occ);
  if (c->reg_host && c->reg_owner) hipHostUnregister(c->reg_host);
  hipFree(c->d_heavy[0]); hipFree(c->d_heavy[1]); hipFree(c->d_heavy_flags); hipFree(c->d_timeline);
  hipFree(c->d_mesh_cost); hipFree(c->d_mesh_order); hipFree(c->d_spheres);
  hipFree(c->d_verts_m); hipFree(c->d_normals_m); hipFree(c->d_colors_m); hipFree(c->d_orig); hipFree(c->d_tile_box);
  delete c;
}

}  // extern "C"
