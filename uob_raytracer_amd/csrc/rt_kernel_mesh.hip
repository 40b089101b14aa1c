// rt_kernel_mesh.hip — the wave-mapped ray tracer for meshes of ANY size (n > 64 triangles), e.g. the
// Cornell Box plus a Loader.cpp OBJ mesh of ~100 k triangles (BASELINE.json configs[4]).
//
// Same mapping and the same three exact levels as rt_kernel_wave.hip (see there and DESIGN.md 4.1); what
// changes is where the triangles live.  The staged records (rt_trace.h, 128 B per triangle) are written
// once per frame to HBM by rt_stage_records and STREAM THROUGH LDS IN TILES OF 64 TRIANGLES shared by the
// four waves of a workgroup (each wave = one 64-ray task, all four walk the tiles in lock step):
//   primary rays : per tile, lane = triangle bounds the task's ray bundle (primary_clear) and the closest-hit
//                  loop visits only the tile's surviving triangles, carrying (t, index, u, v) across tiles in
//                  index order — ties resolve exactly as in the reference's single loop (kernels.cl:120);
//   shadows      : per tile, level 1 (lane = triangle) bounds all lit surface points of the task, level 2
//                  (lane = surface point) bounds each point against the tile's survivors, level 3 (lane =
//                  shadow sample) runs the reference's test for what is left; each surface point's blocked-
//                  sample mask lives in its lane across tiles, so the any-hit OR over the whole mesh and its
//                  early-out (mask == all samples) are exact.
// A wave's 64 pixels form an 8x8 block (not a row segment as in rt_kernel_wave.hip), and each 64-ray task a
// compact PTx x PTy sub-block of it: the bounds are taken over a task's rays / surface points, so their
// footprint should be as small as possible in both directions.
// A brute-force pass over 100 k triangles costs ~100 k tests per ray; here it costs one bound per
// (64-ray task, triangle) plus the few real tests.  Mirror / glass bounce rays run in rounds, the workgroup's four
// tasks together, every round streaming all tiles through LDS with the same lane = triangle bound in front.
#define RT_SPHERES_IN_LDS
#define RT_RNG_JUMP_IN_LDS
#include "rt_wave_common.h"

namespace uobrt {

namespace {

#ifndef RT_MESH_MIN_BLOCKS
#define RT_MESH_MIN_BLOCKS 4
#endif
constexpr int kTile = 64;                   // triangles per LDS tile
constexpr int kBatch = 4;                   // candidate tiles staged per barrier round (4 records of each: 16 KB)
constexpr int kSlot = 4 * kTile;            // float4 per staged tile
constexpr int kMeshWaves = 4;               // waves (= tasks) per workgroup sharing a tile
constexpr int kDirectSamples = 2;           // up to this many, level 2 is skipped as well (the test is cheaper than its bound)
// Up to this many shadow samples level 3 runs lane = surface point (every lane tests its own point's samples one after the other)
// instead of lane = sample (two points per pass, NS of 64 lanes busy).  Measured (1024^2, reference constants otherwise): box +
// 226-triangle mesh, 10 samples: 0.70 ms with the limit at 8, 0.585 at 16; box + 4 680 triangles, 16 samples: 7.98 / 5.94;
// 10 samples at 2048^2: 10.2 / 5.07; at 24 and 32 samples the two forms are level (0.73 / 0.72, 8.11 / 8.45), at 64 lane =
// point is 1.7 x slower.
#ifndef RT_MESH_POINT_SAMPLES
#define RT_MESH_POINT_SAMPLES 16
#endif
constexpr int kPointSamples = RT_MESH_POINT_SAMPLES;
constexpr int kScreenCell = 32;             // pixels per side of a screen cell of the primary-ray tile masks
constexpr int kScreenCellLog = 5;

// World cell of a shadow ray's start point (rt_bin_shadow bounds exactly the points that map to a cell)
__device__ __forceinline__ int world_cell(const FrameParams& P, f3 s) {
  const int g1 = P.grid_g - 1;
  int ix = (int)floorf((s.x - P.grid_lo[0]) * P.grid_inv), iy = (int)floorf((s.y - P.grid_lo[1]) * P.grid_inv),
      iz = (int)floorf((s.z - P.grid_lo[2]) * P.grid_inv);
  ix = ix < 0 ? 0 : (ix > g1 ? g1 : ix); iy = iy < 0 ? 0 : (iy > g1 ? g1 : iy); iz = iz < 0 ? 0 : (iz > g1 ? g1 : iz);
  return (iz * P.grid_g + iy) * P.grid_g + ix;
}

struct MeshWaveLds {
  float4* h0;   // start.xyz | radius_sq
  float4* h1;   // dir.xyz
  uint32_t* rng;
  float4* grp;  // 4 float4 per group: s0|es, D0|ed, (dlen_min, dlen_max, hh, M), (lane mask lo, hi, -, -)
};
constexpr int kMaxGroups = 6;               // coherent groups of a task's surface points bounded separately by level 1
__host__ __device__ constexpr int mesh_wave_lds_bytes(bool rng) {          // rng: scratch of the lane = sample level 3
  return 64 * 32 + kMaxGroups * 64 + (rng ? kRngPixels * kRngStride * 4 : 0);
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)v, lane);
}

// Level 3 for two surface points of one pixel against the tile triangles K & need (rt_kernel_wave.hip
// wave_unshadowed_pair), with the points' blocked-sample masks passed in and out.
struct Mask2 { unsigned long long a, b; };
__device__ __forceinline__ Mask2 tile_test_pair(const float4* tv0, const float4* te1, const float4* te2, const float4* tc,
                                                const MeshWaveLds& L, int ja, int jb, unsigned long long K,
                                                unsigned long long need, f3 jit, unsigned long long active,
                                                unsigned long long sha, unsigned long long shb, unsigned long long& iters) {
  const float4 ha0 = L.h0[ja], ha1 = L.h1[ja], hb0 = L.h0[jb], hb1 = L.h1[jb];     // LDS broadcasts
  const f3 sa = mk(ha0.x, ha0.y, ha0.z), sb = mk(hb0.x, hb0.y, hb0.z);
  const float ra = ha0.w, rb = hb0.w;
  const f3 da = mk(ha1.x, ha1.y, ha1.z) + jit, db = mk(hb1.x, hb1.y, hb1.z) + jit;   // dir + crush(...), :333
  const f3 nda = -da, ndb = -db;
  for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull) {
    const int k = __builtin_ctzll(kk);
    if (((need >> k) & 1ull) == 0ull) continue;
    ++iters;
    const f3 v0 = xyz(tv0[k]), e1 = xyz(te1[k]), e2 = xyz(te2[k]), c = xyz(tc[k]);
    const f3 ba = sa - v0, bb = sb - v0;
    const float nA0a = detc(ba, c), nA0b = detc(bb, c);
    const float detAa = detc(nda, c), detAb = detc(ndb, c);
    float rra = rcp_newton(detAa, 1), rrb = rcp_newton(detAb, 1);
    float ta = nA0a * rra, tb = nA0b * rrb;
    f3 dva = ta * da, dvb = tb * db;
    float dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
    float distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
    unsigned long long passa = ballot(!(ta < 0.0f)) & ballot(!(dista >= ra));
    unsigned long long passb = ballot(!(tb < 0.0f)) & ballot(!(distb >= rb));
    if (((passa & active & ~sha) | (passb & active & ~shb)) == 0ull) continue;
    if ((ballot(rra != rra) | ballot(rrb != rrb)) != 0ull) {         // rare: reciprocal outside v_rcp's range
      rra = 1.0f / detAa; rrb = 1.0f / detAb;
      ta = nA0a * rra; tb = nA0b * rrb;
      dva = ta * da; dvb = tb * db;
      dista = dva.x * dva.x + dva.y * dva.y + dva.z * dva.z;
      distb = dvb.x * dvb.x + dvb.y * dvb.y + dvb.z * dvb.z;
      passa = ballot(ta >= 0) & ballot(dista < ra);
      passb = ballot(tb >= 0) & ballot(distb < rb);
    }
    const float ua = detc(nda, cof(ba, e2)) * rra, va = detc(nda, cof(e1, ba)) * rra;
    const float ub = detc(ndb, cof(bb, e2)) * rrb, vb = detc(ndb, cof(e1, bb)) * rrb;
    sha |= active & passa & ballot(ua >= 0) & ballot(va >= 0) & ballot((ua + va) <= 1);
    shb |= active & passb & ballot(ub >= 0) & ballot(vb >= 0) & ballot((ub + vb) <= 1);
    if (sha == active && shb == active) break;
  }
  Mask2 r;
  r.a = sha; r.b = shb;
  return r;
}

// ---- bounce rays: may ANY ray of a bundle hit ANY triangle of a tile? -----------------------------------------------------
// The per-triangle bound (task_bound, lane = triangle) needs the tile in LDS; this one needs 48 bytes per tile and is asked
// lane = tile, 64 tiles per pass, before anything is loaded.  A plain ray-box test would NOT do: for a ray that lies in a
// triangle's plane all the determinants of the reference's test vanish, its t, u, v are rounding noise, and the reference may
// "hit" a triangle the ray passes at any distance — noise this library must reproduce.  The certificate:
//   With W1 = det(A1), W2 = det(A2), W0 = det(A) - W1 - W2 (the three edge functions; u = W1 / det(A), v = W2 / det(A),
//   1 - u - v = W0 / det(A)) and any n perpendicular to the ray's direction,   sum_i W_i n.(v_i - o) = 0   holds identically
//   (sum_i W_i (v_i - o) = det(A0) d).  Take n = d x e_k (k = x, y, z: the separating axes of a line and a box): if the
//   tile's box lies on one side of that plane through the ray, every a_i = n.(v_i - o) is in [gap, gap + 2 rad], gap > 0.
//   The reference accepts only if u >= 0, v >= 0, fl(u + v) <= 1, i.e. if the COMPUTED W_i all have det(A)'s sign (up to
//   4 eps of their magnitudes); by the identity that is possible only if all three computed W_i are within
//   Omega = 3 E (1 + 2 rad / gap) of zero, E <= 21 eps |d| (|b| + |e|) |e| bounding their rounding errors (eps = 2^-24).
//   And they are NOT all that small unless the ray lies in the triangle's plane:   max_i |W_i| >= 0.28 theta |c| |d|   where
//   theta <= max(|n_T . d| / |d|, |n_T . (o - v)| / bmax) (n_T the triangle's unit normal, bmax >= |o - v|): from
//   det(A) = sum W_i and, for in-plane g, sum W_i g.(v_i - o) = det(A0) g.d.  Per tile the normals lie in a cone (unit axis a,
//   chord chi = max |n_T -+ a|), so theta >= max(|a . d| / |d|, |a . (o - v)| / bmax) - chi, bounded over the bundle and the box.
//   Certified clear iff  gap > 0  and  theta >= 150 eps (4 + 6 rad / gap) (bmax + emax) eta   (factor of safety 2 included),
//   eta = max |e| / |e1 x e2| and emax = max |e| over the tile's triangles (rt_api.hip upload_tiled_scene).
// Bundle: origins s0 +- es, directions D0 +- ed per component, |d|_2 <= dmax2.  Conservative in every term; `false` = visit.
__device__ __forceinline__ bool tile_clear_for_bundle(const float4* __restrict__ tb, f3 s0, f3 D0, float es, float ed, float dmax2) {
  const float4 lo4 = tb[0], hi4 = tb[1], ax4 = tb[2];
  const float eta = lo4.w, emax = hi4.w, chi = ax4.w;
  const f3 axis = xyz(ax4);
  const f3 cB = 0.5f * (xyz(lo4) + xyz(hi4));
  // (half extents with the rounding of the staged e1 = v1 - v0, e2 = v2 - v0 and of cB itself)
  const float cs = 4e-7f * (norm_inf(xyz(lo4)) + norm_inf(xyz(hi4)));
  const f3 hB = mk(0.5f * (hi4.x - lo4.x) * 1.00001f + cs, 0.5f * (hi4.y - lo4.y) * 1.00001f + cs, 0.5f * (hi4.z - lo4.z) * 1.00001f + cs);
  const f3 r0 = cB - s0;
  const float esb = es * 1.0001f + 1e-6f * norm_inf(s0);
  const float bmax = 1.7321f * (norm_inf(r0) + fmaxf(fmaxf(hB.x, hB.y), hB.z) + esb);
  const float rr[3] = {r0.x, r0.y, r0.z}, dd[3] = {D0.x, D0.y, D0.z}, hh[3] = {hB.x, hB.y, hB.z};
  float ratio = 3.0e38f;                                               // smallest rad / gap over the axes that separate
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int j = (k + 1) % 3, l = (k + 2) % 3;
    const float t1 = dd[l] * rr[j], t2 = dd[j] * rr[l];
    const float f0 = t1 - t2;                                          // (d x e_k) . r up to sign
    const float dev = ed * (fabsf(rr[j]) + fabsf(rr[l])) + esb * (fabsf(dd[l]) + fabsf(dd[j])) + 2.0f * ed * esb;
    const float rad = hh[j] * (fabsf(dd[l]) + ed) + hh[l] * (fabsf(dd[j]) + ed);
    const float gap = fabsf(f0) - dev - rad - 1e-5f * (fabsf(t1) + fabsf(t2) + dev + rad);   // (minus this evaluation's own rounding)
    if (gap > 0.0f) ratio = fminf(ratio, rad / gap);
  }
  if (!(ratio < 1.0e30f)) return false;
  const float a1 = norm1(axis);
  const float k_line = (fabsf(bdot3(axis, D0)) - ed * a1) / dmax2;
  const float k_orig = (fabsf(bdot3(axis, r0)) - (fabsf(axis.x) * hB.x + fabsf(axis.y) * hB.y + fabsf(axis.z) * hB.z) - esb * a1) / bmax;
  const float theta = fmaxf(k_line, k_orig) * 0.9999f - chi;
  return theta >= 8.95e-6f * (4.0f + 6.0f * ratio) * (bmax + emax) * eta;     // 150 * 2^-24 = 8.94e-6
}

// Geometry of a wave's 8x8 pixel block: its 64 pixels are numbered along a Z-order curve (x bits 0,2,4 / y bits
// 1,3,5 of the number), task k covers the PT consecutive numbers from k*PT — for a power of two PT a compact
// rectangle (4x2, 4x4, 8x4 ...), PTx >= PTy; for any other PT (AA grids such as 3x3: PT = 7) a compact run of the curve.
__device__ __forceinline__ int zorder_x(int q) { return (q & 1) | ((q >> 1) & 2) | ((q >> 2) & 4); }
__device__ __forceinline__ int zorder_y(int q) { return ((q >> 1) & 1) | ((q >> 2) & 2) | ((q >> 3) & 4); }
__device__ __forceinline__ int zorder_of(int x, int y) {
  return (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2) | ((x & 4) << 2) | ((y & 4) << 3);
}
struct BlockGeom {
  int x0, lr0;        // first pixel column / first packed local row of the 8x8 block
  int PT;             // pixels per task
  __device__ __forceinline__ int q(int k, int p) const { return k * PT + p; }
  __device__ __forceinline__ int bx(int k, int p) const { return zorder_x(q(k, p)); }
  __device__ __forceinline__ int by(int k, int p) const { return zorder_y(q(k, p)); }
};

// xorshift streams of the GP pixels of RNG group g of the current task into the wave's scratch (:319,:331): the
// `cnt` samples that follow the first `skip` ones (passes of 64 samples when there are more than 64)
__device__ __forceinline__ void generate_streams(const FrameParams& P, const MeshWaveLds& L, int lane, int GP, int skip, int cnt,
                                                 const BlockGeom& B, int k, int first_p) {
  // kRngSegs lanes per stream, lane `seg` writing samples [13 seg, 13 seg + 13) from the jumped state (rt_wave_common.h)
  const int q3 = lane / 3, comp = lane - 3 * q3;
  const int seg = GP == 4 ? q3 >> 2 : (GP == 2 ? q3 >> 1 : (GP == 1 ? q3 : q3 / GP)), pp = q3 - seg * GP;
  if (seg < kRngSegs) {
    const int qq = B.q(k, first_p + pp) < 64 ? B.q(k, first_p + pp) : 63;      // a pixel past the block is never lit
    const int px = B.x0 + zorder_x(qq);
    const int py = band_global_row(P, B.lr0 + zorder_y(qq));
    const int gid = pixel_global_id(P, px, py);
    const uint32_t seed = comp == 0 ? (uint32_t)gid : (uint32_t)((float)gid * (comp == 1 ? 91.0f : 19.0f));
    uint32_t s = xorshift(seed);
    uint32_t* dst = L.rng + pp * kRngStride + comp;
    for (int it = 0; it < skip; ++it) s = xorshift(s);
    if (seg > 0) s = rng_jump(s, seg);
    const int it0 = seg * kRngSegLen;
#pragma unroll 1
    for (int j = 0; j < kRngSegLen; ++j) { s = xorshift(s); if (it0 + j < cnt) dst[(it0 + j) * 4] = s; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// ---- per-frame candidate-tile masks --------------------------------------------------------------------
// The exact bounds of rt_wave_common.h are certificates about a SET of rays ("no ray of the set can make the
// reference's test accept this triangle"); what holds for a set holds for every subset.  So the same bounds,
// evaluated once per frame for coarse sets, tell every task which 64-triangle tiles it may skip WITHOUT
// loading them:
//   rt_bin_primary : set = all primary rays through a 64x64-pixel screen cell  -> screen_masks[cell] bit t
//   rt_bin_shadow  : set = all shadow rays whose start point lies in a world-grid cell (any lit surface point
//                    there, any jitter sample)                                  -> world_masks[cell] bit t
// bit t = "some triangle of tile t is not certified clear".  A workgroup ORs the masks of the cells its rays
// belong to and walks only the set bits; inside a tile the per-task levels 1-3 run as before.
// One wave per tile (lane = triangle, registers hold its record), looping over the cells.
__global__ __launch_bounds__(64 * kMeshWaves) void rt_bin_primary(const FrameParams P) {
  const int lane = threadIdx.x & 63;
  const int n = P.n, ntiles = (n + kTile - 1) / kTile;
  const int t = blockIdx.x * kMeshWaves + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const int gi = t * kTile + lane;
  const bool ok = gi < n;
  const size_t g = ok ? gi : t * kTile;
  const float4 c4 = P.records[(size_t)3 * n + g];
  const f3 c = xyz(c4), pc = xyz(P.records[(size_t)6 * n + g]), qc = xyz(P.records[(size_t)7 * n + g]);
  const f3 r0 = mk(P.rot[0], P.rot[1], P.rot[2]), r1 = mk(P.rot[4], P.rot[5], P.rot[6]), r2 = mk(P.rot[8], P.rot[9], P.rot[10]);
  // does any triangle of the tile stay uncertified for the primary rays through the pixel rectangle [xlo, xhi] x [ylo, yhi]?
  auto rect_open = [&](int xlo, int xhi, int ylo, int yhi) -> bool {
    const float Ylo = ((float)(ylo * P.aa_y) - P.half_hy) * P.sy;
    const float Yhi = ((float)(yhi * P.aa_y + P.aa_y - 1) - P.half_hy) * P.sy;
    const float hy = 0.5f * (Yhi - Ylo);
    const float Xlo = (float)(xlo * P.aa_x) - P.half_wx;
    const float Xhi = (float)(xhi * P.aa_x + P.aa_x - 1) - P.half_wx;
    const float hx = 0.5f * (Xhi - Xlo);
    const f3 wc = mk(Xlo + hx, Ylo + hy, P.focal);
    const f3 duc = mk(dot3(r0, wc), dot3(r1, wc), dot3(r2, wc));
    const f3 eu = mk(1.0001f * (fabsf(r0.x) * hx + fabsf(r0.y) * hy), 1.0001f * (fabsf(r1.x) * hx + fabsf(r1.y) * hy),
                     1.0001f * (fabsf(r2.x) * hx + fabsf(r2.y) * hy));
    const float dumax = fmaxf(fmaxf(fabsf(duc.x) + eu.x, fabsf(duc.y) + eu.y), fabsf(duc.z) + eu.z);
    const bool clear = (dumax < 1e30f) && primary_clear(duc, eu, dumax, c, c4.w, pc, qc);
    return ballot(ok && !clear) != 0ull;
  };
  // blockIdx.y = a row of 4 x 4-cell blocks: a block first (what is clear for a set of rays is clear for every subset — a
  // tile of a mesh covers a few cells of the thousands), its sixteen cells only where that leaves something open
  const int cyb = blockIdx.y * 4;
  for (int cxb = 0; cxb < P.scx; cxb += 4) {
    const int bx1 = (cxb + 4 < P.scx ? cxb + 4 : P.scx), by1 = (cyb + 4 < P.scy ? cyb + 4 : P.scy);
    const int pxhi = bx1 * kScreenCell - 1 < P.W - 1 ? bx1 * kScreenCell - 1 : P.W - 1;
    const int pyhi = by1 * kScreenCell - 1 < P.H - 1 ? by1 * kScreenCell - 1 : P.H - 1;
    if (!rect_open(cxb * kScreenCell, pxhi, cyb * kScreenCell, pyhi)) continue;
    for (int cy = cyb; cy < by1; ++cy) {
      const int ylo = cy * kScreenCell, yhi = (ylo + kScreenCell - 1) < P.H ? (ylo + kScreenCell - 1) : (P.H - 1);
      for (int cx = cxb; cx < bx1; ++cx) {
        const int xlo = cx * kScreenCell, xhi = (xlo + kScreenCell - 1) < P.W ? (xlo + kScreenCell - 1) : (P.W - 1);
        if (rect_open(xlo, xhi, ylo, yhi) && lane == 0)
          atomicOr(&P.screen_masks[((size_t)cy * P.scx + cx) * P.nwords + (t >> 6)], 1ull << (t & 63));
      }
    }
  }
}

// Words of the three occupancy bitmaps (cells of edge 1, 2 and 4 grid cells), one after the other
__host__ __device__ inline int occ_offset(int G, int level) {
  const int w0 = (G * G * G + 31) / 32, w1 = ((G / 2) * (G / 2) * (G / 2) + 31) / 32;
  return level == 0 ? 0 : level == 1 ? w0 : w0 + w1;
}
__host__ __device__ inline int occ_words(int G) { return occ_offset(G, 2) + ((G / 4) * (G / 4) * (G / 4) + 31) / 32; }

// Which world cells can hold the start point of a shadow ray at all: every surface point lies on a triangle or on
// a sphere, and its start point X + 1e-4 (light - X) within 1e-4 |light - X| of it.  One thread per triangle (then
// per sphere) marks the cells its bounding box touches, widened by that distance, by the rounding of world_cell()
// and by 1 % of a cell; rt_bin_shadow skips every other cell (no task ever reads their masks).
// (A mesh puts tens of thousands of triangles into a few dozen cells: every workgroup ORs into a copy of the three bitmaps in
// LDS and sends what it has set to memory once — atomics on one address are serialised memory-side, and with one per wave and
// word the kernel spent 0.46 ms of configs[4]'s frame on them.)
constexpr int kOccThreads = 1024;
__global__ __launch_bounds__(kOccThreads) void rt_bin_occupancy(const FrameParams P) {
  extern __shared__ unsigned int s_occ[];
  const int i = blockIdx.x * kOccThreads + threadIdx.x;
  const int G = P.grid_g;
  const int n_occ_words = occ_words(G);
  for (int w = threadIdx.x; w < n_occ_words; w += kOccThreads) s_occ[w] = 0u;
  __syncthreads();
  f3 lo, hi;
  if (i < P.n) {
    const f3 a0 = xyz(P.records[i]), a1 = a0 + xyz(P.records[(size_t)P.n + i]), a2 = a0 + xyz(P.records[(size_t)2 * P.n + i]);
    lo = mk(fminf(fminf(a0.x, a1.x), a2.x), fminf(fminf(a0.y, a1.y), a2.y), fminf(fminf(a0.z, a1.z), a2.z));
    hi = mk(fmaxf(fmaxf(a0.x, a1.x), a2.x), fmaxf(fmaxf(a0.y, a1.y), a2.y), fmaxf(fmaxf(a0.z, a1.z), a2.z));
  } else if (i < P.n + P.nsph) {
    const DevSphere& sp = P.sph[i - P.n];
    const float r = sqrtf(fmaxf(sp.r2, 0.0f)) * 1.0001f;
    lo = mk(sp.cx - r, sp.cy - r, sp.cz - r); hi = mk(sp.cx + r, sp.cy + r, sp.cz + r);
  } else {
    lo = mk(0.f, 0.f, 0.f); hi = lo;
  }
  bool valid = i < P.n + P.nsph;
  const float amax = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
  const float lmax = P.light_inf;
  // A point on a triangle is v0 + u e1 + v e2 with u, v in [0,1]: inside the triangle's box whatever the ray was.  A hit
  // point on a SPHERE is X = start + x dir (kernels.cl:225) with x from the quadratic's discriminant b*b - 4*a*c, which
  // the reference rounds by up to ~16 eps |d|^2 |L|^2 (L = start - centre): x is off by up to 4 sqrt(eps) |L| / |d| =
  // 1e-3 |L| when the ray grazes the sphere — seen from a camera 50 000 units away the "hit point" lies tens of units
  // off the sphere, and every path must still shade it the same way.  So a sphere's cells are those within
  // 2e-3 (|L|max + R) of its box, |L|max over the camera and every possible bounce-ray origin (the scene itself).
  const float cmax = max_abs3(P.cam[0], P.cam[1], P.cam[2]);
  float sl = 2e-4f * (lmax + amax) + 1e-5f * (1.0f + amax + cmax);
  if (i >= P.n && i < P.n + P.nsph) {
    const DevSphere& sp = P.sph[i - P.n];
    const f3 Lc = mk(P.cam[0] - sp.cx, P.cam[1] - sp.cy, P.cam[2] - sp.cz);
    const float far = fmaxf(bsqrt(dot3(Lc, Lc)), 1.7321f * P.grid_cell * (float)G);
    sl += 2e-3f * (far + sqrtf(fmaxf(sp.r2, 0.0f)));
  }
  if (!(amax < 1e30f)) valid = false;               // non-finite vertices: no ray can hit such a triangle
  int c0[3], c1[3];
  const float lo3[3] = {lo.x, lo.y, lo.z}, hi3[3] = {hi.x, hi.y, hi.z};
  for (int k = 0; k < 3; ++k) {
    const float f0 = floorf((lo3[k] - sl - P.grid_lo[k]) * P.grid_inv - 0.01f), f1 = floorf((hi3[k] + sl - P.grid_lo[k]) * P.grid_inv + 0.01f);
    c0[k] = f0 < 0.0f ? 0 : (f0 > (float)(G - 1) ? G - 1 : (int)f0);
    c1[k] = f1 < 0.0f ? 0 : (f1 > (float)(G - 1) ? G - 1 : (int)f1);
  }
  // A mesh puts tens of thousands of triangles into a few cells: the lanes of a wave that touch one cell only
  // combine their bits and send one atomic per distinct word (atomics on one address are serialised memory-side)
  const bool single = valid && c0[0] == c1[0] && c0[1] == c1[1] && c0[2] == c1[2];
  for (int level = 0; level < 3; ++level) {
    const int g = G >> level;
    unsigned int* occ = s_occ + occ_offset(G, level);
    const int cell = (((c0[2] >> level) * g + (c0[1] >> level)) * g + (c0[0] >> level));
    const int word = single ? (cell >> 5) : -1;
    const unsigned int bit = single ? (1u << (cell & 31)) : 0u;
    for (unsigned long long rem = ballot(single); rem != 0ull;) {
      const int src = __builtin_ctzll(rem);
      const int w = __builtin_amdgcn_readlane(word, src);
      const unsigned long long same = ballot(word == w);
      unsigned int m = word == w ? bit : 0u;
      for (int off = 32; off > 0; off >>= 1) m |= (unsigned int)__shfl_xor((int)m, off, 64);
      if ((threadIdx.x & 63) == src) atomicOr(&occ[w], m);
      rem &= ~same;
    }
    if (valid && !single)
      for (int z = c0[2] >> level; z <= (c1[2] >> level); ++z)
        for (int y = c0[1] >> level; y <= (c1[1] >> level); ++y) {        // the cells of one x-run are consecutive bits
          const int b0 = (z * g + y) * g + (c0[0] >> level), b1 = (z * g + y) * g + (c1[0] >> level);
          for (int w = b0 >> 5; w <= (b1 >> 5); ++w) {
            const int lo_b = w == (b0 >> 5) ? (b0 & 31) : 0, hi_b = w == (b1 >> 5) ? (b1 & 31) : 31;
            const unsigned int m = (hi_b - lo_b == 31 ? ~0u : ((1u << (hi_b - lo_b + 1)) - 1u)) << lo_b;
            atomicOr(&occ[w], m);
          }
        }
  }
  __syncthreads();
  for (int w = threadIdx.x; w < n_occ_words; w += kOccThreads) {
    const unsigned int m = s_occ[w];
    if (m != 0u) atomicOr(&P.world_occ[w], m);
  }
}

__device__ __forceinline__ bool occupied(const FrameParams& P, int level, int ix, int iy, int iz) {
  if (P.mask_debug & 4) return true;
  const int g = P.grid_g >> level;
  const int cell = (iz * g + iy) * g + ix;
  return ((P.world_occ[occ_offset(P.grid_g, level) + (cell >> 5)] >> (cell & 31)) & 1u) != 0u;
}

// Grid: x = tiles / 4, y = z-slice of the world grid.
__global__ __launch_bounds__(64 * kMeshWaves) void rt_bin_shadow(const FrameParams P) {
  const int lane = threadIdx.x & 63;
  const int n = P.n, ntiles = (n + kTile - 1) / kTile;
  const int t = blockIdx.x * kMeshWaves + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const int gi = t * kTile + lane;
  const size_t g = gi < n ? gi : t * kTile;
  const bool ok = gi < n && P.records[(size_t)5 * n + g].w != -1.0f;          // glass casts no shadow, :247
  TriLane T1;
  T1.v0 = xyz(P.records[g]); T1.e1 = xyz(P.records[(size_t)n + g]); T1.e2 = xyz(P.records[(size_t)2 * n + g]);
  T1.c = xyz(P.records[(size_t)3 * n + g]);
  T1.c1 = norm1(T1.c); T1.e1_1 = norm1(T1.e1); T1.e2_1 = norm1(T1.e2);
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const float linf = P.light_inf;
  const float hbox = P.hbox;
  const float half = 0.5f * P.grid_cell;
  const int G = P.grid_g;
  const float tv0inf = fmaxf(fmaxf(fabsf(T1.v0.x), fabsf(T1.v0.y)), fabsf(T1.v0.z));
  // every start point that world_cell() maps to the cell with centre C and half edge h lies in C +- hs
  // (slack: the rounding of that mapping); its dir = light - X (kernels.cl:323) = (light - C) + (C - X),
  // |C - X| <= hs + 1e-4 |dir|  (start = X + 1e-4 dir, :324)
  auto cell_clear = [&](f3 C, float h) {
    const float cinf = norm_inf(C);
    const float hs = 1.001f * h + 1e-5f * (1.0f + cinf);
    const f3 D0 = light - C;
    const float d0len = bsqrt(dot3(D0, D0));
    const float dinf = norm_inf(D0);
    const float ed = hs + 1.1e-4f * (d0len + 2.0f * hs) + 1e-6f * (1.0f + dinf + cinf);
    const float dlen_max = (d0len + 1.7321f * ed) * 1.00001f;
    const float dlen_min = fmaxf(d0len - 1.7321f * ed, 0.0f) * 0.99999f;
    const float hh = 1.002f * hbox + 2e-6f * (dlen_max + hbox);
    return light_bundle_bound(T1, light, C, hs, D0, ed, hh, dlen_min, dlen_max, linf + tv0inf + dlen_max).clear;
  };
  // three levels, 4x4x4 -> 2x2x2 -> single cells: what is clear for a union of cells is clear for each of them
  // (grid.y, grid.z = y, z of the coarsest level; G is a multiple of 4)
  auto centre = [&](int ix, int iy, int iz, float edge) {
    return mk(P.grid_lo[0] + ((float)ix + 0.5f) * edge, P.grid_lo[1] + ((float)iy + 0.5f) * edge, P.grid_lo[2] + ((float)iz + 0.5f) * edge);
  };
  const int Gq = G >> 2, izq = blockIdx.z, iyq = blockIdx.y;         // one row of coarsest cells per wave
  // Where the coarse cell's own bundle is not clear, the tile as a whole is asked next, lane = one of the coarse cell's
  // 4 x 4 x 4 single cells: the bounce rays' tile certificate (tile_clear_for_bundle: box, normal cone and sliver measure of
  // the tile) for each single cell's bundle — start points C +- hs, directions (light - C) +- (ed + jitter half-width).  It
  // reasons differently from the per-triangle intervals (a separating plane through the ray instead of the signs of the
  // determinants), so it clears (tile, cell) pairs the hierarchy below cannot: 6 % fewer tiles per cell on configs[4].
  const bool tile_test = PC(tile_box) != nullptr && !(P.mask_debug & 256);
  for (int ixq = 0; ixq < Gq; ++ixq) {
      if (!occupied(P, 2, ixq, iyq, izq)) continue;
      if (ballot(ok && !cell_clear(centre(ixq, iyq, izq, 4.0f * P.grid_cell), 4.0f * half)) == 0ull) continue;
      // single cells (bit = 16 dz + 4 dy + dx) still to be decided: the occupied ones (all 64 looked up at once, lane = cell —
      // a block of cells is occupied exactly if one of its cells is, so the loops below need no further look-ups) ...
      unsigned long long open;
      {
        const int dx = lane & 3, dy = (lane >> 2) & 3, dz = lane >> 4;
        const int ix = 4 * ixq + dx, iy = 4 * iyq + dy, iz = 4 * izq + dz;
        bool want = occupied(P, 0, ix, iy, iz);
        if (tile_test && want) {                                        // ... that the tile as a whole is not clear of
          const f3 C = centre(ix, iy, iz, P.grid_cell);
          const float cinf = norm_inf(C);
          const float hs = 1.001f * half + 1e-5f * (1.0f + cinf);
          const f3 D0 = light - C;
          const float d0len = bsqrt(dot3(D0, D0));
          const float ed = hs + 1.1e-4f * (d0len + 2.0f * hs) + 1e-6f * (1.0f + norm_inf(D0) + cinf);
          const float dlen_max = (d0len + 1.7321f * ed) * 1.00001f;
          const float hh = 1.002f * hbox + 2e-6f * (dlen_max + hbox);
          want = !tile_clear_for_bundle(PC(tile_box) + (size_t)3 * t, C, D0, hs, ed * 1.0001f + hh, (dlen_max + 1.7321f * hh) * 1.0001f);
        }
        open = ballot(want);
        if (open == 0ull) continue;
      }
      for (int s2 = 0; s2 < 8; ++s2) {
        const int ixc = 2 * ixq + (s2 & 1), iyc = 2 * iyq + ((s2 >> 1) & 1), izc = 2 * izq + (s2 >> 2);
        // (its eight single cells: dx in {2 (s2&1), +1}, ...)
        const unsigned long long sub_bits = 0x0000000000330033ull << (2 * (s2 & 1) + 8 * ((s2 >> 1) & 1) + 32 * (s2 >> 2));
        const int n_open = __popcll(open & sub_bits);
        if (n_open == 0) continue;
        // (the tile test leaves 12 % of the single cells open: the bound of the 2 x 2 x 2 block is only worth its price where
        // it can spare three or more of them)
        if (n_open >= 3 && ballot(ok && !cell_clear(centre(ixc, iyc, izc, 2.0f * P.grid_cell), 2.0f * half)) == 0ull) continue;
        for (int sub = 0; sub < 8; ++sub) {
          const int ix = 2 * ixc + (sub & 1), iy = 2 * iyc + ((sub >> 1) & 1), iz = 2 * izc + (sub >> 2);
          if (((open >> (((iz & 3) << 4) | ((iy & 3) << 2) | (ix & 3))) & 1ull) == 0ull) continue;      // (unoccupied, or the tile test settled it)
          const unsigned long long m = ballot(ok && !cell_clear(centre(ix, iy, iz, P.grid_cell), half));
          if (m != 0ull && lane == 0)
            atomicOr(&P.world_masks[((size_t)(iz * G + iy) * G + ix) * P.nwords + (t >> 6)], 1ull << (t & 63));
        }
      }
  }
}

// Grid: the workgroups the device holds at once (persistent, see the job loop); block = 4 waves = 2x2 blocks of 8x8 pixels.
// COUNT: diagnostic build that also sums what the waves executed into P.counters[0..7] (rt_count_executed)
// PROF: diagnostic build (never timed) that sums s_memtime cycles per phase over the waves instead
#define MESH_STAMP(slot)                                                            \
  if (PROF) {                                                                       \
    unsigned long long now_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                              \
    xw[slot] += now_ - tlast;                                                       \
    tlast = now_;                                                                   \
  }
// AA_X, AA_Y, SS > 0: AA grid and sample count as compile-time constants (as rt_kernel_wave.hip); no such instantiation is
// shipped (see launch_mesh).
template <bool COUNT, bool PROF = false, int AA_X = 0, int AA_Y = 0, int SS = 0>
__global__ __launch_bounds__(64 * kMeshWaves, RT_MESH_MIN_BLOCKS) void rt_draw_mesh(const FrameParams P) {
  unsigned long long xw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
  extern __shared__ float4 lds[];
  float4* tile = lds;                                   // kBatch staged tiles x 4 records x kTile triangles
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wave_bytes = mesh_wave_lds_bytes((SS ? SS : P.S) > kPointSamples || P.nsph > 0);
  char* const wbase = reinterpret_cast<char*>(lds + kBatch * kSlot) + wave * wave_bytes;
  const MeshWaveLds L{reinterpret_cast<float4*>(wbase), reinterpret_cast<float4*>(wbase + 64 * 16),
                      reinterpret_cast<uint32_t*>(wbase + 64 * 32 + kMaxGroups * 64), reinterpret_cast<float4*>(wbase + 64 * 32)};
  const int n = P.n, ntiles = (n + kTile - 1) / kTile;
  const int nwords = (ntiles + 63) >> 6;
  // candidate-tile masks of this workgroup: primary rays (fixed for the frame), shadow rays (per task round)
  unsigned long long* pmask = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lds + kBatch * kSlot) +
                                                                    kMeshWaves * wave_bytes);
  unsigned long long* smask = pmask + nwords;
  // cooperative blocks (below): per lane, what each of the four waves found in its share of the tiles
  struct CoopHit { float t, u, v; int best, orig; };
  CoopHit* const coop_hit = reinterpret_cast<CoopHit*>(smask + nwords);                       // [kMeshWaves][64]
  unsigned long long* const coop_sh = reinterpret_cast<unsigned long long*>(coop_hit + kMeshWaves * 64);   // [kMeshWaves][64] blocked samples
  unsigned int* const coop_fl = reinterpret_cast<unsigned int*>(coop_sh + kMeshWaves * 64);  // [kMeshWaves][64] bit 0 blocked, bit 1 task_blocked
  const bool bins = PC(screen_masks) != nullptr;
  // Persistent workgroups: a job is one 16x16-pixel block of the frame (this workgroup's four 8x8 tasks-blocks); the
  // blocks that look at the mesh's silhouette or stand in its shadow cost a hundred times what a wall block costs,
  // and the frame ends when the last one does — so they are pulled from a queue, the blocks that were expensive in
  // the context's PREVIOUS frame first (PC(mesh_order), built by rt_mesh_order from the costs each job records; the
  // first frame goes from the middle rows outwards).  No pixel depends on the order.
  __shared__ int s_job;
  stage_spheres(P, tid);                               // (the job loop's first barrier publishes them)
  stage_rng_jump(tid);
  const int wgx_n = (P.W + 15) / 16, wgy_n = (P.owned_rows + 15) / 16, n_jobs = wgx_n * wgy_n;
  const int n_queue = PC(mesh_order) != nullptr ? (int)PC(mesh_queue_len)[0] : n_jobs;
  const int wg_mid = (wgy_n + 1) >> 1;
  for (;;) {
  __syncthreads();                                     // the previous job's LDS (and s_job) are no longer read
  if (tid == 0) s_job = (int)atomicAdd(PC(job_counter), 1u);
  __syncthreads();
  if (s_job >= n_queue) break;                         // the counter only grows: every workgroup gets here
  // An entry of the order list is (block << 3) | (cooperative << 2) | sub-block.  A block that was VERY expensive in the
  // previous frame comes as four cooperative entries, one per 8x8 sub-block: the four waves then work on the SAME 64
  // pixels and share the TILES (of every staged batch of four, wave w takes tile w), merging what they found per lane
  // through LDS — closest hit: smallest (t, original index); shadows: OR of the blocked-sample masks.  The longest
  // unit of work is then a quarter of a sub-block's tiles instead of a whole block's.
  const unsigned int entry = PC(mesh_order) != nullptr ? PC(mesh_order)[s_job] : ((unsigned int)s_job << 3);
  const int job = (int)(entry >> 3);
  const bool coop = (entry & 4u) != 0u;
  const int coop_q = (int)(entry & 3u);
  const unsigned long long job_t0 = (COUNT || PC(mesh_cost) != nullptr) ? __builtin_amdgcn_s_memtime() : 0ull;
  const int job_y = job / wgx_n, job_x = job - job_y * wgx_n;
  // without an order list: rows from the middle of the frame outwards (the last to start are the top and bottom ones)
  const int wg_row = (job_y & 1) ? wg_mid + (job_y >> 1) : wg_mid - 1 - (job_y >> 1);
  for (int w = tid; w < nwords; w += 64 * kMeshWaves) {
    unsigned long long m = (bins && !(PC(mask_debug) & 1)) ? 0ull : ~0ull;
    if (bins && !(PC(mask_debug) & 1)) {
      const int cx = (job_x * 16) >> kScreenCellLog;
      int last = -1;
      for (int r = 0; r < 16; ++r) {                      // the workgroup's 16 packed rows: global y may jump at a band edge
        const int lrr = wg_row * 16 + r;
        if (lrr >= P.owned_rows) break;
        const int cy = band_global_row_cold(P, lrr) >> kScreenCellLog;
        if (cy != last) m |= PC(screen_masks)[((size_t)cy * PC(scx) + cx) * nwords + w];
        last = cy;
      }
    }
    pmask[w] = m;
  }
  __syncthreads();
  const LdsScene G = lds_scene(P.records, n);           // the whole mesh, in HBM (hit finalisation, bounce rays)

  const int aa_x = AA_X ? AA_X : P.aa_x, aa_y = AA_Y ? AA_Y : P.aa_y;
  const float sy = (AA_X && AA_Y) ? (float)AA_X / (float)(AA_Y ? AA_Y : 1) : P.sy;
  const int aa = aa_x * aa_y;                           // <= 64 (mesh_kernel_supports()); lanes past PT * aa idle
  const float inv_S = SS ? ((SS & (SS - 1)) == 0 ? 1.0f / (float)(SS ? SS : 1) : 0.0f) : P.inv_S;
  const float inv_aa = (AA_X && AA_Y) ? (((AA_X * AA_Y) & (AA_X * AA_Y - 1)) == 0 ? 1.0f / (float)(AA_X * AA_Y ? AA_X * AA_Y : 1) : 0.0f) : P.inv_aa;
  const int PT = 64 / aa;                               // pixels per task
  const int ntask = (64 + PT - 1) / PT;                 // tasks per 8x8 block
  const int pt_magic = (65536 + PT - 1) / PT;           // q / PT == (q * pt_magic) >> 16 for q < 64
  BlockGeom B;
  const int sub = coop ? coop_q : wave;                 // the 8x8 sub-block this wave renders
  B.x0 = job_x * 16 + (sub & 1) * 8;
  B.lr0 = wg_row * 16 + (sub >> 1) * 8;                 // waves past the frame still walk the tiles (barriers)
  B.PT = PT;
  const int GP = PT < kRngPixels ? PT : kRngPixels;
  const int GL = GP * aa;
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const float hbox = P.hbox;
  const int NS = SS ? SS : P.S;
  const int n_pass = (NS + 63) >> 6;                    // more than 64 shadow samples: passes of 64 sample lanes
  Work wk;

  // Candidate tiles are staged kBatch at a time (one barrier round each): 4 records x 64 triangles per tile,
  // one float4 per thread and tile.  primary: c|det(cam-v0,e1,e2), cof(cam-v0,e2), cof(e1,cam-v0);
  // shadow: v0|material, e1, e2, c.
  auto next_tile = [&](const unsigned long long* mask, int& w, unsigned long long& m) -> int {
    while (m == 0ull) {
      if (++w >= nwords) return -1;
      m = uniform64(mask[w]);
    }
    const int t = w * 64 + __builtin_ctzll(m);
    m &= m - 1ull;
    return t < ntiles ? t : -1;
  };
  auto load_batch = [&](int t0, int t1, int t2, int t3, int cnt, bool primary) {
    const int rec = tid >> 6, i = tid & 63;
    const int src = primary ? (rec == 0 ? 3 : rec == 1 ? 6 : rec == 2 ? 7 : 1) : rec;    // primary slot 3: e1 | original index
    __syncthreads();
    for (int sl = 0; sl < cnt; ++sl) {
      const int t = sl == 0 ? t0 : sl == 1 ? t1 : sl == 2 ? t2 : t3;
      const int gi = t * kTile + i;
      tile[sl * kSlot + tid] = gi < n ? P.records[(size_t)src * n + gi] : make_float4(0.f, 0.f, 0.f, -1.0f);
    }
    __syncthreads();
  };

  f3 outc = mk(0.f, 0.f, 0.f);
  for (int k = 0; k < ntask; ++k) {
    // ---- phase 1: primary rays over all tiles -------------------------------------------------------
    const int p = (AA_X && AA_Y) ? lane / (AA_X * AA_Y ? AA_X * AA_Y : 1) : (lane * P.aa_magic) >> 16;    // pixel of this lane within the task (lane / aa)
    const int a = lane - p * aa;                // AA sample index dy*rx+dx, kernels.cl:395
    const bool in_task = p < PT && B.q(k, p) < 64;
    const int qz = in_task ? B.q(k, p) : B.q(k, 0);
    const int x = B.x0 + zorder_x(qz);
    const int lr = B.lr0 + zorder_y(qz);
    const bool valid = in_task && lr < P.owned_rows && x < P.W;
    const int y = band_global_row_cold(P, lr < P.owned_rows ? lr : 0);
    const int ay = AA_X ? a / (AA_X ? AA_X : 1) : (a * P.aax_magic) >> 16;        // a / aa_x (a < 256)
    Ray ray = primary_ray(P, x, y, a - ay * aa_x, ay, aa_x, aa_y, sy);
    f3 duc, eu;
    float dumax;
    {
      // sub-pixel rectangle of the task: the bounding box of its pixels (the global row of a packed row may jump at
      // a band boundary, and a run of the Z curve is no rectangle: take minima and maxima over the task's lanes)
      const int yl = band_global_row_cold(P, lr < P.owned_rows ? lr : (P.owned_rows > 0 ? P.owned_rows - 1 : 0));
      const float xmin = wave_min_pos((float)x), xmax = wave_max_pos((float)x), ymin = wave_min_pos((float)yl), ymax = wave_max_pos((float)yl);
      const float Xlo = xmin * (float)aa_x - P.half_wx;
      const float Xhi = (xmax * (float)aa_x + (float)(aa_x - 1)) - P.half_wx;
      const float Ylo = (ymin * (float)aa_y - P.half_hy) * sy;
      const float Yhi = ((ymax * (float)aa_y + (float)(aa_y - 1)) - P.half_hy) * sy;
      const float hx = 0.5f * (Xhi - Xlo), hy = 0.5f * (Yhi - Ylo);
      const f3 wc = mk(Xlo + hx, Ylo + hy, P.focal);
      const f3 r0 = mk(P.rot[0], P.rot[1], P.rot[2]), r1 = mk(P.rot[4], P.rot[5], P.rot[6]),
               r2 = mk(P.rot[8], P.rot[9], P.rot[10]);
      duc = mk(dot3(r0, wc), dot3(r1, wc), dot3(r2, wc));
      eu = mk(1.0001f * (fabsf(r0.x) * hx + fabsf(r0.y) * hy), 1.0001f * (fabsf(r1.x) * hx + fabsf(r1.y) * hy),
              1.0001f * (fabsf(r2.x) * hx + fabsf(r2.y) * hy));
      dumax = fmaxf(fmaxf(fabsf(duc.x) + eu.x, fabsf(duc.y) + eu.y), fabsf(duc.z) + eu.z);
    }
    // can any primary ray of the task touch a sphere at all? (else the quadratic tests per ray are skipped)
    const bool sph_task = P.nsph > 0 && (!(dumax < 1e30f) ||
                          ballot(sphere_bundle_maybe(P, lane, mk(P.cam[0], P.cam[1], P.cam[2]), 0.0f, duc, bsqrt(dot3(duc, duc)),
                                                     1.0001f * bsqrt(dot3(eu, eu)), false)) != 0ull);
    MESH_STAMP(0)
    float current_t = RT_MAXFLOAT, bu = 0.f, bv = 0.f;
    int best = -1, best_o = 0x7fffffff;          // best: position in the reordered mesh; best_o: its original index
    const f3 ndp = -ray.dir;
    auto primary_tile = [&](int t, const float4* tb) {
      const float4 *t_c = tb, *t_pc = tb + kTile, *t_qc = tb + 2 * kTile, *t_or = tb + 3 * kTile;
      const int nc = (n - t * kTile) < kTile ? (n - t * kTile) : kTile;
      unsigned long long Kp = nc == 64 ? ~0ull : ((1ull << nc) - 1ull);
      {
        const float4 c4 = t_c[lane];
        const bool clear = primary_clear(duc, eu, dumax, xyz(c4), c4.w, xyz(t_pc[lane]), xyz(t_qc[lane]));
        if (dumax < 1e30f) Kp &= ~ballot(clear);
      }
      if (COUNT) { xw[0]++; xw[1] += __popcll(Kp); }
      if (valid) {
        // (uniform64: a loop inside a divergent `if` otherwise keeps its wave-uniform mask in vector registers)
        for (unsigned long long m = uniform64(Kp); m != 0ull; m &= m - 1ull) {
          const int i = __builtin_ctzll(m);
          const float4 c4 = t_c[i];
          const f3 pc = xyz(t_pc[i]), qc = xyz(t_qc[i]);
          const float detA_recip = rcp_exact(detc(ndp, xyz(c4)));
          const float tt = c4.w * detA_recip;
          const float u = detc(ndp, pc) * detA_recip;
          const float v = detc(ndp, qc) * detA_recip;
          // the reference visits the triangles in their ORIGINAL order and replaces the hit only for a strictly smaller
          // t (kernels.cl:120): of equal t the lowest original index stays — whatever order the tiles come in
          const int oi = __float_as_int(t_or[i].w);
          if (u >= 0 && v >= 0 && (u + v) <= 1 && tt >= 0 && (tt < current_t || (tt == current_t && best >= 0 && oi < best_o))) {
            best = t * kTile + i; best_o = oi; bu = u; bv = v; current_t = tt;
          }
        }
      }
    };
    {
      int bw = -1;
      unsigned long long bm = 0ull;
      for (;;) {
        int t0 = -1, t1 = -1, t2 = -1, t3 = -1, cnt = 0;
        if ((t0 = next_tile(pmask, bw, bm)) >= 0) { cnt = 1;
          if ((t1 = next_tile(pmask, bw, bm)) >= 0) { cnt = 2;
            if ((t2 = next_tile(pmask, bw, bm)) >= 0) { cnt = 3;
              if ((t3 = next_tile(pmask, bw, bm)) >= 0) cnt = 4; } } }
        if (cnt == 0) break;
        MESH_STAMP(2)
        load_batch(t0, t1, t2, t3, cnt, true);
        MESH_STAMP(1)
        for (int sl = 0; sl < cnt; ++sl)
          if (!coop || sl == wave) primary_tile(sl == 0 ? t0 : sl == 1 ? t1 : sl == 2 ? t2 : t3, tile + sl * kSlot);
        if (cnt < kBatch) break;
      }
    }
    if (coop) {                                         // closest hit over all four waves' tiles: smallest (t, original index)
      coop_hit[wave * 64 + lane] = CoopHit{current_t, bu, bv, best, best_o};
      __syncthreads();
      for (int w = 0; w < kMeshWaves; ++w) {
        const CoopHit h = coop_hit[w * 64 + lane];
        if (h.best >= 0 && (best < 0 || h.t < current_t || (h.t == current_t && h.orig < best_o))) {
          current_t = h.t; bu = h.u; bv = h.v; best = h.best; best_o = h.orig;
        }
      }
      __syncthreads();
    }
    MESH_STAMP(2)
    bool lit = false, secondary = false;
    if (valid) {
      if (best >= 0) {
        ray.tri = best;
        ray.P = (xyz(G.v0[best]) + bu * xyz(G.e1[best])) + bv * xyz(G.e2[best]);
        ray.N = xyz(G.nrm[best]);
        ray.col = G.col[best];
      }
      if (sph_task) closest_spheres<false>(P, ray, current_t, wk);
      if (ray.tri != -1) {
        if (ray.col.w <= 0.0f) secondary = true;
        else lit = true;
      }
    }
    // ---- phase 2: mirror / glass bounces (secondary_light, kernels.cl:342-365), round by round, the workgroup's four
    // tasks together: every round streams ALL tiles through LDS once (bounce rays have no per-frame tile masks); per
    // tile, lane = triangle bounds the wave's bundle of bounce rays (origin box x direction box, as rt_kernel_wave.hip
    // does per round) and the closest-hit loop visits the survivors, resolving equal t by the original index.
    {
      bool bouncing = secondary;
      for (int b = 0; b < P.bounces; ++b) {
        const bool act = bouncing && ray.col.w <= 0.0f;               // this lane's loop condition, :345
        if (__syncthreads_or(act ? 1 : 0) == 0) break;                // workgroup-uniform
        if (act) ray = (ray.col.w == 0.0f) ? reflect_ray(ray) : refract_ray(ray);
        const unsigned long long actm = ballot(act);
        int mode = 2;                                                 // 0: nothing to test, 1: bound per tile, 2: every triangle
        f3 s0 = mk(0.f, 0.f, 0.f), D0 = s0;
        float es = 0.f, ed = 0.f, dl = 0.f;
        if (actm != 0ull) {
          const f3 o = ray.start, d = ray.dir;
          const float mag = fmaxf(norm_inf(o), norm_inf(d));
          const bool fin = act && mag < 1e30f;                         // false for NaN as well
          const bool isnan_ = act && !(mag == mag);                    // a NaN ray hits nothing whatever the set
          const unsigned long long finm = ballot(fin);
          const bool odd = ballot(act && !fin && !isnan_) != 0ull;     // infinite coordinates: no bound
          if (finm != 0ull && !odd) {
            const float big = 3.0e38f;
            const f3 olo = mk(wave_min(fin ? o.x : big), wave_min(fin ? o.y : big), wave_min(fin ? o.z : big));
            const f3 ohi = mk(wave_max(fin ? o.x : -big), wave_max(fin ? o.y : -big), wave_max(fin ? o.z : -big));
            const f3 dlo = mk(wave_min(fin ? d.x : big), wave_min(fin ? d.y : big), wave_min(fin ? d.z : big));
            const f3 dhi = mk(wave_max(fin ? d.x : -big), wave_max(fin ? d.y : -big), wave_max(fin ? d.z : -big));
            s0 = 0.5f * (olo + ohi); D0 = 0.5f * (dlo + dhi);
            es = 0.5001f * fmaxf(fmaxf(ohi.x - olo.x, ohi.y - olo.y), ohi.z - olo.z) + 1e-6f * norm1(s0);
            ed = 0.5001f * fmaxf(fmaxf(dhi.x - dlo.x, dhi.y - dlo.y), dhi.z - dlo.z) + 1e-6f * norm1(D0);
            const float dmx = fmaxf(fmaxf(fmaxf(fabsf(dlo.x), fabsf(dhi.x)), fmaxf(fabsf(dlo.y), fabsf(dhi.y))), fmaxf(fabsf(dlo.z), fabsf(dhi.z)));
            dl = 1.7321f * dmx * 1.0001f;                               // >= |d|_2 of every ray
            mode = 1;
          } else if (finm == 0ull && !odd) {
            mode = 0;                                                   // only NaN rays
          }
        }
        float cur = RT_MAXFLOAT, hu = 0.f, hv = 0.f;
        int hit = -1, hit_o = 0x7fffffff;
        const f3 ndb = -ray.dir;
        auto bounce_tile = [&](int t, const float4* tb) {
          const float4 *t_v0 = tb, *t_e1 = tb + kTile, *t_e2 = tb + 2 * kTile, *t_c = tb + 3 * kTile;
          if (actm == 0ull || mode == 0) return;                      // wave-uniform; the barriers are behind us
          const int nc = (n - t * kTile) < kTile ? (n - t * kTile) : kTile;
          unsigned long long Kb = nc == 64 ? ~0ull : ((1ull << nc) - 1ull);
          if (mode == 1) {
            TriLane Tb;
            Tb.v0 = xyz(t_v0[lane]); Tb.e1 = xyz(t_e1[lane]); Tb.e2 = xyz(t_e2[lane]); Tb.c = xyz(t_c[lane]);
            Tb.c1 = norm1(Tb.c); Tb.e1_1 = norm1(Tb.e1); Tb.e2_1 = norm1(Tb.e2);
            Kb &= ~ballot(task_bound(Tb, s0, D0, es, ed, 2e-6f * dl, 0.0f, dl).clear);
          }
          if (act)
            for (unsigned long long m = uniform64(Kb); m != 0ull; m &= m - 1ull) {
              const int i = __builtin_ctzll(m);
              const float4 e14 = t_e1[i];
              const f3 v0 = xyz(t_v0[i]), e1 = xyz(e14), e2 = xyz(t_e2[i]), c = xyz(t_c[i]);
              const f3 bb = ray.start - v0;
              const float detA_recip = rcp_exact(detc(ndb, c));
              const float tt = detc(bb, c) * detA_recip;
              const float u = detc(ndb, cof(bb, e2)) * detA_recip;
              const float v = detc(ndb, cof(e1, bb)) * detA_recip;
              const int oi = __float_as_int(e14.w);
              if (u >= 0 && v >= 0 && (u + v) <= 1 && tt >= 0 && (tt < cur || (tt == cur && hit >= 0 && oi < hit_o))) {
                hit = t * kTile + i; hit_o = oi; hu = u; hv = v; cur = tt;
              }
            }
        };
        // Which tiles?  lane = tile, 64 tiles per pass (tile_clear_for_bundle), the four waves' answers OR-ed into the
        // workgroup's tile mask; RT_FLAG_NO_TILE_BINS (no masks at all) streams every tile as before.
        __syncthreads();
        for (int w = tid; w < nwords; w += 64 * kMeshWaves) smask[w] = 0ull;
        __syncthreads();
        if (actm != 0ull && mode != 0) {
          const bool pretest = bins && mode == 1 && PC(tile_box) != nullptr && !(PC(mask_debug) & 64);
          // |d|_2 of the wave's rays (they are normalised: 1 within rounding; lanes without a ray contribute 0)
          const float d2 = pretest ? 1.0001f * bsqrt(wave_max_pos((act && mode == 1) ? dot3(ray.dir, ray.dir) : 0.0f)) * 1.0001f : 0.0f;
          for (int base = 0; base < ntiles; base += 64) {
            const int t = base + lane;
            bool visit = t < ntiles;
            if (pretest && visit) visit = !tile_clear_for_bundle(PC(tile_box) + (size_t)3 * t, s0, D0, es, ed, d2);
            const unsigned long long vm = ballot(visit);
            if (lane == 0 && vm != 0ull) atomicOr(&smask[base >> 6], vm);
          }
        }
        __syncthreads();
        {
          int bw = -1;
          unsigned long long bm = 0ull;
          for (;;) {
            int t0 = -1, t1 = -1, t2 = -1, t3 = -1, cnt = 0;
            if ((t0 = next_tile(smask, bw, bm)) >= 0) { cnt = 1;
              if ((t1 = next_tile(smask, bw, bm)) >= 0) { cnt = 2;
                if ((t2 = next_tile(smask, bw, bm)) >= 0) { cnt = 3;
                  if ((t3 = next_tile(smask, bw, bm)) >= 0) cnt = 4; } } }
            if (cnt == 0) break;
            load_batch(t0, t1, t2, t3, cnt, false);
            if (COUNT && actm != 0ull) xw[0] += cnt;                     // (diagnostic: bounce-round tile visits count as primary visits)
            for (int sl = 0; sl < cnt; ++sl)
              if (!coop || sl == wave) bounce_tile(sl == 0 ? t0 : sl == 1 ? t1 : sl == 2 ? t2 : t3, tile + sl * kSlot);
            if (cnt < kBatch) break;
          }
        }
        if (coop) {      // cooperative block: the four waves hold the SAME rays and took a tile each — closest hit = smallest (t, original index)
          coop_hit[wave * 64 + lane] = CoopHit{cur, hu, hv, hit, hit_o};
          __syncthreads();
          for (int w = 0; w < kMeshWaves; ++w) {
            const CoopHit h = coop_hit[w * 64 + lane];
            if (h.best >= 0 && (hit < 0 || h.t < cur || (h.t == cur && h.orig < hit_o))) { cur = h.t; hu = h.u; hv = h.v; hit = h.best; hit_o = h.orig; }
          }
          __syncthreads();
        }
        if (act) {
          if (hit >= 0) {
            ray.tri = hit;
            ray.P = (xyz(G.v0[hit]) + hu * xyz(G.e1[hit])) + hv * xyz(G.e2[hit]);
            ray.N = xyz(G.nrm[hit]);
            ray.col = G.col[hit];
          }
          closest_spheres<false>(P, ray, cur, wk);
          if (ray.tri != -1 && ray.col.w > 0.0f) { lit = true; bouncing = false; }
        }
      }
    }
    // per-lane light set-up of direct_light, kernels.cl:323-326
    const f3 dir = light - ray.P;
    const f3 start = ray.P + 0.0001f * dir;
    const float radius_sq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    const float term = (16.0f * fmaxf(dot3(dir, ray.N), 0.0f)) / (4.0f * 3.14159274f * radius_sq);
    __builtin_amdgcn_wave_barrier();
    L.h0[lane] = make_float4(start.x, start.y, start.z, radius_sq);
    L.h1[lane] = make_float4(dir.x, dir.y, dir.z, 0.f);
    __builtin_amdgcn_wave_barrier();

    // a point that faces away from the light (term == 0 exactly) sums +-0 whatever its samples' masks are: no shadow tests
    // for it (rt_kernel_wave.hip) — half the surface of a closed mesh, whose rays would otherwise cross the whole mesh
    const bool slit = lit && !(term == 0.0f);
    // ---- phase 3: shadows over all tiles ---------------------------------------------------------------
    const unsigned long long litmask = ballot(slit);
    const float dlen = bsqrt(radius_sq);
    const float hh = 1.002f * hbox + 2e-6f * (dlen + hbox);
    float dminlen = dlen - 1.7321f * hh;
    const bool sane = slit && (radius_sq > 1e-18f) && (radius_sq < 1e30f);
    if (!sane || !(dminlen > 0.0f)) dminlen = 0.0f;
    const float dk = dlen * 1.000004f;
    // Level 1 bounds a SET of surface points, and is only as tight as the set is compact.  A task whose pixels
    // straddle a silhouette holds points on surfaces far apart, so the lit points are split into groups by
    // world cell (the last group takes whatever is left) and each group is bounded on its own.
    const bool task_ok = litmask != 0ull && ballot(slit && !sane) == 0ull;
    int ngroups = 0, grp = -1;
    bool task_sph = P.nsph > 0 && !task_ok;
    if (task_ok) {
      const int ci = bins ? world_cell(P, start) : 0;
      const float linf_l = P.light_inf;
      for (unsigned long long rem = litmask; rem != 0ull; ++ngroups) {
        const int jr = __builtin_ctzll(rem);
        const int cj = __builtin_amdgcn_readlane(ci, jr);
        const unsigned long long gm = ngroups == kMaxGroups - 1 ? rem : (rem & ballot(ci == cj));
        const bool one_cell = bins && (gm & ~ballot(ci == cj)) == 0ull;      // false for an overflow group over several cells
        rem &= ~gm;
        const bool in = ((gm >> lane) & 1ull) != 0ull;
        const f3 s0 = mk(rl(start.x, jr), rl(start.y, jr), rl(start.z, jr));
        const f3 D0 = mk(rl(dir.x, jr), rl(dir.y, jr), rl(dir.z, jr));
        const f3 ds = start - s0, dd = dir - D0;
        const float es = wave_max_pos(in ? norm_inf(ds) : 0.0f);
        const float ed = wave_max_pos(in ? norm_inf(dd) : 0.0f);
        const float dlen_max = wave_max_pos(in ? dlen : 0.0f);
        const float dlen_min = wave_min_pos(in ? dlen : 3.0e38f);
        if (in) grp = ngroups;
        if (P.nsph > 0) {     // may any shadow ray of the group touch a shadow-casting sphere?
          const float hh_g = 1.002f * hbox + 2e-6f * (dlen_max + hbox);
          const float s0inf = norm_inf(s0), d0inf = norm_inf(D0);
          task_sph = task_sph || ballot(sphere_bundle_maybe(P, lane, s0, 1.001f * es + 2e-6f * (s0inf + es), D0, rl(dlen, jr) * 1.000001f,
                                                            1.7321f * (1.001f * ed + 2e-6f * d0inf + hh_g), true)) != 0ull;
        }
        if (lane == 0) {
          L.grp[4 * ngroups] = make_float4(s0.x, s0.y, s0.z, es);
          L.grp[4 * ngroups + 1] = make_float4(D0.x, D0.y, D0.z, ed);
          L.grp[4 * ngroups + 2] = make_float4(dlen_min, dlen_max, 1.002f * hbox + 2e-6f * (dlen_max + hbox), linf_l + dlen_max);
          // .z: the group's world cell (all its points start there), -1 for the overflow group, which holds several
          L.grp[4 * ngroups + 3] = make_float4(__uint_as_float((unsigned)gm), __uint_as_float((unsigned)(gm >> 32)),
                                               __int_as_float(one_cell ? cj : -1), 0.f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    SphereBound sb;
    sb.maybe = false; sb.all_blocked = false;
    if (task_sph && sane) sb = spheres_point(P, start, dir, dlen, hh);
    const unsigned long long sphmask = ballot(slit && P.nsph > 0 && (sb.maybe || !sane));
    // this lane's pixel's xorshift stream after the seed call (kernels.cl:319), for the point-parallel level 3
    uint32_t rs0 = 0u, rs1 = 0u, rs2 = 0u;
    if (NS <= kPointSamples) {
      const int gid = pixel_global_id(P, x, y);
      rs0 = xorshift((uint32_t)gid); rs1 = xorshift((uint32_t)((float)gid * 91.0f)); rs2 = xorshift((uint32_t)((float)gid * 19.0f));
    }
    unsigned long long my_sh = 0ull;            // blocked samples of THIS lane's surface point (current pass), across tiles
    unsigned long long active = 0ull;           // sample lanes of the current pass
    int first_s = 0, cnt_s = 0;                 // its first sample and its number of samples
    int unshadowed = 0;
    bool blocked = sane && sb.all_blocked, task_blocked = false;
    int rng_group = -1;                         // which pixel group's streams the scratch currently holds
    // candidate tiles of the workgroup's four tasks: OR of the world-cell masks of their lit surface points
    __syncthreads();
    for (int w = tid; w < nwords; w += 64 * kMeshWaves) smask[w] = 0ull;
    __syncthreads();
    if (litmask != 0ull) {
      if (!bins || (PC(mask_debug) & 2) || ballot(slit && !sane) != 0ull) {
        for (int w = lane; w < nwords; w += 64) atomicOr(&smask[w], ~0ull);
      } else {
        const int ci = world_cell(P, start);
        for (unsigned long long rem = litmask; rem != 0ull;) {
          const int cj = __builtin_amdgcn_readlane(ci, __builtin_ctzll(rem));
          rem &= ~ballot(ci == cj);
          const unsigned long long* src = PC(world_masks) + (size_t)cj * nwords;
          for (int w = lane; w < nwords; w += 64) {
            const unsigned long long v = src[w];
            if (v != 0ull) atomicOr(&smask[w], v);
          }
        }
      }
    }
    __syncthreads();
    MESH_STAMP(3)
    // the world-cell masks of this wave's groups, one 64-tile word at a time (lane g = group g): the tiles come in rising
    // order, so a word is loaded once per 64 tiles instead of once per tile and group in front of every bound
    int gm_word = -1;
    unsigned long long gm_bits = ~0ull;
    auto shadow_tile = [&](int t, const float4* tbase) {
      const float4 *t_v0 = tbase, *t_e1 = tbase + kTile, *t_e2 = tbase + 2 * kTile, *t_c = tbase + 3 * kTile;
      if (litmask == 0ull || task_blocked) return;              // wave-uniform; the barriers are behind us
      unsigned long long gwant = ~0ull;                          // groups whose own cell names this tile
      if (task_ok) {
        if ((t >> 6) != gm_word) {
          gm_word = t >> 6;
          unsigned long long v = ~0ull;
          if (lane < ngroups) {
            const int cellg = __float_as_int(L.grp[4 * lane + 3].z);
            if (cellg >= 0 && !(PC(mask_debug) & 2)) v = PC(world_masks)[(size_t)cellg * nwords + gm_word];
          }
          gm_bits = v;
        }
        gwant = ballot(lane < ngroups && ((gm_bits >> (t & 63)) & 1ull) != 0ull);
        if (gwant == 0ull) return;                               // (the tile is in the workgroup's union for another wave's sake)
      }
      const int nc = (n - t * kTile) < kTile ? (n - t * kTile) : kTile;
      const unsigned long long casts = ballot(lane < nc && t_v0[lane].w != -1.0f);    // glass casts no shadow, :247
      unsigned long long K = casts;
      unsigned long long mymask = casts;          // the tile triangles THIS lane's surface point may still need
      if (task_ok) {                                               // level 1, lane = triangle, per group of points
        TriLane T1;
        T1.v0 = xyz(t_v0[lane]); T1.e1 = xyz(t_e1[lane]); T1.e2 = xyz(t_e2[lane]); T1.c = xyz(t_c[lane]);
        T1.c1 = norm1(T1.c); T1.e1_1 = norm1(T1.e1); T1.e2_1 = norm1(T1.e2);
        const float v0n = norm1(T1.v0);
        const unsigned long long alive = ballot(slit && !blocked && (my_sh & active) != active);
        K = 0ull; mymask = 0ull;
        for (int g = 0; g < ngroups; ++g) {
          const float4 g0 = L.grp[4 * g], g1 = L.grp[4 * g + 1], g2 = L.grp[4 * g + 2], g3 = L.grp[4 * g + 3];
          const unsigned long long gm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(g3.y)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(g3.x));
          if ((gm & alive) == 0ull) continue;
          // the group's own cell has certified this whole tile clear (rt_bin_shadow): nothing to bound, nothing to test.
          // (The workgroup visits the union of its 256 points' cells' tiles; a task that straddles a silhouette holds
          // six groups, and most tiles of the union matter to one of them.)
          if (((gwant >> g) & 1ull) == 0ull) continue;
          unsigned long long Kg = casts;
          if (g0.w < 1e30f && g1.w < 1e30f) {
            const Bound tb = light_bundle_bound(T1, light, mk(g0.x, g0.y, g0.z), g0.w, mk(g1.x, g1.y, g1.z), g1.w, g2.z, g2.x, g2.y,
                                                g2.w + v0n);
            Kg = casts & ~ballot(tb.clear);
            if ((casts & ballot(tb.all_blocked)) != 0ull) {          // one triangle blocks every sample of the group
              if (grp == g) blocked = true;
              continue;
            }
          }
          if (grp == g) mymask = Kg;
          K |= Kg;
        }
        if (ballot(slit && !blocked) == 0ull) { task_blocked = true; return; }
      }
      MESH_STAMP(5)
      if (COUNT) { xw[2]++; xw[3] += __popcll(K); }
      if (K == 0ull) return;
      K = uniform64(K);                                            // the loops over K then run on the scalar unit
      unsigned long long need = 0ull;                              // level 2, lane = surface point; bit = tile triangle
      // with one or two samples per point the bound costs more than the samples: test every candidate
      if (NS <= kDirectSamples) need = mymask;
      else for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull) {
        const int kq = __builtin_ctzll(kk);
        const bool part = ((mymask >> kq) & 1ull) != 0ull;
        if (ballot(part && slit && !blocked) == 0ull) continue;
        const Bound pb = point_bound(start, dir, hh, dlen, dminlen, dk, xyz(t_v0[kq]), xyz(t_e1[kq]), xyz(t_e2[kq]), xyz(t_c[kq]), part && slit && !blocked);
        if (part && (!pb.clear || !sane)) need |= 1ull << kq;
        blocked = blocked || (part && sane && pb.all_blocked);
      }
      MESH_STAMP(6)
      const bool mine = slit && !blocked && need != 0ull && (my_sh & active) != active;
      const unsigned long long work = ballot(mine);
      if (NS <= kPointSamples) {
        // level 3 with few samples, lane = surface point: every lane runs the reference's test (kernels.cl:251-272)
        // for its own point, sample after sample, against the tile triangles its `need` names — one pass serves
        // 64 points (lane = sample would keep NS of 64 lanes busy)
        if (work != 0ull) {
          uint32_t r0 = rs0, r1 = rs1, r2 = rs2;
          for (int sm = 0; sm < NS; ++sm) {
            r0 = xorshift(r0); r1 = xorshift(r1); r2 = xorshift(r2);                     // rand = random(rand), :331
            const bool todo = mine && ((my_sh >> sm) & 1ull) == 0ull;
            if (ballot(todo) == 0ull) continue;
            const f3 d = dir + mk(crush1(r0, P.spread), crush1(r1, P.spread), crush1(r2, P.spread));   // :333
            const f3 nd = -d;
            bool hit = false;
            for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull) {
              const int kq = __builtin_ctzll(kk);
              const bool go = todo && !hit && ((need >> kq) & 1ull) != 0ull;
              if (ballot(go) == 0ull) continue;
              if (COUNT) xw[5]++;
              if (go) {
                const f3 v0 = xyz(t_v0[kq]), c = xyz(t_c[kq]);
                const f3 b = start - v0;
                const float detA_recip = rcp_exact(detc(nd, c));
                const float tt = detc(b, c) * detA_recip;
                const f3 dv = tt * d;
                const float dist = dv.x * dv.x + dv.y * dv.y + dv.z * dv.z;
                if (tt >= 0 && dist < radius_sq) {
                  const f3 e1 = xyz(t_e1[kq]), e2 = xyz(t_e2[kq]);
                  const float u = detc(nd, cof(b, e2)) * detA_recip;
                  const float v = detc(nd, cof(e1, b)) * detA_recip;
                  if (u >= 0 && v >= 0 && (u + v) <= 1) hit = true;
                }
              }
            }
            if (hit) my_sh |= 1ull << sm;
          }
          if (COUNT) xw[4]++;
        }
        return;
      }
      for (int g = 0; g * GL < 64 && work != 0ull; ++g) {          // level 3, lane = shadow sample
        const unsigned long long gm = (work >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
        if (gm == 0ull) continue;
        if (rng_group != g) { generate_streams(P, L, lane, GP, first_s, cnt_s, B, k, g * GP); rng_group = g; }
        for (int pp = 0; pp < GP; ++pp) {
          unsigned long long pm = (gm >> (pp * aa)) & (aa == 64 ? ~0ull : ((1ull << aa) - 1ull));
          if (pm == 0ull) continue;
          const uint32_t* src = L.rng + pp * kRngStride + lane * 4;
          const f3 jit = mk(crush1(src[0], P.spread), crush1(src[1], P.spread), crush1(src[2], P.spread));
          const int base = g * GL + pp * aa;
          while (pm != 0ull) {
            const int j = base + __builtin_ctzll(pm);
            pm &= pm - 1ull;
            const int j2 = pm != 0ull ? base + __builtin_ctzll(pm) : j;
            pm &= pm - 1ull;
            const Mask2 m2 = tile_test_pair(t_v0, t_e1, t_e2, t_c, L, j, j2, K, readlane64(need, j) | readlane64(need, j2), jit,
                                            active, readlane64(my_sh, j), readlane64(my_sh, j2), xw[5]);
            if (COUNT) xw[4]++;
            if (lane == j2) my_sh = m2.b;
            if (lane == j) my_sh = m2.a;
          }
        }
      }
    };
    // More than 64 shadow samples: the whole traversal once per pass of 64 sample lanes (the certificates "all samples
    // blocked" hold for every sample, so a point blocked in one pass stays out of the later ones)
    for (int pass = 0; pass < n_pass; ++pass) {
    first_s = pass << 6;
    cnt_s = NS - first_s < 64 ? NS - first_s : 64;
    active = cnt_s == 64 ? ~0ull : ((1ull << cnt_s) - 1ull);
    my_sh = 0ull;
    rng_group = -1;
    {
      int bw = -1;
      unsigned long long bm = 0ull;
      for (;;) {
        int t0 = -1, t1 = -1, t2 = -1, t3 = -1, cnt = 0;
        if ((t0 = next_tile(smask, bw, bm)) >= 0) { cnt = 1;
          if ((t1 = next_tile(smask, bw, bm)) >= 0) { cnt = 2;
            if ((t2 = next_tile(smask, bw, bm)) >= 0) { cnt = 3;
              if ((t3 = next_tile(smask, bw, bm)) >= 0) cnt = 4; } } }
        if (cnt == 0) break;
        MESH_STAMP(7)
        load_batch(t0, t1, t2, t3, cnt, false);
        MESH_STAMP(4)
        for (int sl = 0; sl < cnt; ++sl)
          if (!coop || sl == wave) shadow_tile(sl == 0 ? t0 : sl == 1 ? t1 : sl == 2 ? t2 : t3, tile + sl * kSlot);
        if (cnt < kBatch) break;
      }
    }
    if (coop) {                                         // any-hit over all four waves' tiles: OR of what each wave found
      coop_sh[wave * 64 + lane] = my_sh;
      coop_fl[wave * 64 + lane] = (blocked ? 1u : 0u) | (task_blocked ? 2u : 0u);
      __syncthreads();
      for (int w = 0; w < kMeshWaves; ++w) {
        my_sh |= coop_sh[w * 64 + lane];
        const unsigned int f = coop_fl[w * 64 + lane];
        blocked = blocked || (f & 1u) != 0u;
        task_blocked = task_blocked || (f & 2u) != 0u;
      }
      __syncthreads();
    }
    // shadow-casting spheres (kernels.cl:278-307) for the points whose rays can reach one
    {
      const unsigned long long sw = ballot(slit && !blocked && !task_blocked && ((sphmask >> lane) & 1ull) != 0ull &&
                                           (my_sh & active) != active);
      for (int g = 0; g * GL < 64 && sw != 0ull; ++g) {
        const unsigned long long gm = (sw >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
        if (gm == 0ull) continue;
        if (rng_group != g) { generate_streams(P, L, lane, GP, first_s, cnt_s, B, k, g * GP); rng_group = g; }
        for (int pp = 0; pp < GP; ++pp) {
          unsigned long long pm = (gm >> (pp * aa)) & (aa == 64 ? ~0ull : ((1ull << aa) - 1ull));
          if (pm == 0ull) continue;
          const uint32_t* src = L.rng + pp * kRngStride + lane * 4;
          const f3 jit = mk(crush1(src[0], P.spread), crush1(src[1], P.spread), crush1(src[2], P.spread));
          for (; pm != 0ull; pm &= pm - 1ull) {
            const int j = g * GL + pp * aa + __builtin_ctzll(pm);
            const float4 h0 = L.h0[j], h1 = L.h1[j];
            const unsigned long long shj = readlane64(my_sh, j);
            bool s1 = (shj >> lane) & 1ull;
            if (!s1) s1 = shadow_spheres<false>(P, mk(h0.x, h0.y, h0.z), mk(h1.x, h1.y, h1.z) + jit, h0.w, wk);
            const unsigned long long upd = shj | (active & ballot(s1));
            if (lane == j) my_sh = upd;
          }
        }
      }
    }
    unshadowed += __popcll(active & ~my_sh);
    }                                             // passes
    if (blocked || task_blocked) unshadowed = 0;

    // ---- phase 4: shading and the AA sum, as in rt_kernel_wave.hip ---------------------------------------
    f3 contrib = mk(0.f, 0.f, 0.f);
    if (lit) {
      float total = 0.0f;
      if (unshadowed < NS) total += 0.0f * term;
#pragma unroll 8
      for (int i = 0; i < NS; ++i) if (i < unshadowed) total += term;       // (one add per trip = one taken branch per add)
      const float l = 0.5f + div_count(total, NS, inv_S);
      if (secondary) { const float kk = 0.9f * l; contrib = mk(kk * ray.col.x, kk * ray.col.y, kk * ray.col.z); }
      else contrib = mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
    }
    const f3 acc = aa_sum(contrib, aa, (p < PT ? p : 0) * aa);
    {   // output lane l owns block pixel (l & 7, l >> 3): take its sum from the task and pixel that cover it
      const int qo = zorder_of(lane & 7, lane >> 3);
      const int ok = (qo * pt_magic) >> 16;
      const int op = qo - ok * PT;
      const f3 v = mk(shfl(acc.x, op * aa), shfl(acc.y, op * aa), shfl(acc.z, op * aa));
      if (ok == k) outc = v;
    }
  }
  MESH_STAMP(7)
  if (COUNT) xw[7] += ntask;
  {
    const int x = B.x0 + (lane & 7);
    const int lr = B.lr0 + (lane >> 3);
    if (!COUNT && !PROF && (!coop || wave == 0) && lr < P.owned_rows && x < P.W) {      // the counting pass has no framebuffer
      const f3 c = mk(div_count(outc.x, aa, inv_aa), div_count(outc.y, aa, inv_aa), div_count(outc.z, aa, inv_aa));
      const size_t o = (size_t)(PC(out_global) ? band_global_row_cold(P, lr) : lr) * P.W + x;
      PC(out_argb)[o] = pack_argb(c);
      if (PC(out_rgb)) PC(out_rgb)[o] = make_float4(c.x, c.y, c.z, 1.0f);
    }
  }
  if (COUNT && tid == 0) atomicMax(&PC(counters)[6], __builtin_amdgcn_s_memtime() - job_t0);   // the longest block, in ticks
  if (PC(mesh_cost) != nullptr && tid == 0) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - job_t0;
    // zeroed per frame.  A cooperative sub-block job counts four-fold (what it would have taken one wave): the block
    // then stays above the threshold that made it cooperative instead of alternating between the two forms.
    const unsigned long long d4 = coop ? 4ull * dt : dt;
    atomicAdd(&PC(mesh_cost)[job], d4 > 0x3fffffffull ? 0x3fffffffu : (unsigned int)d4);
  }
  }                                                    // ---- end of the job loop ----------------------------------
  if (PROF) { if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&PC(counters)[q], xw[q]); return; }
  if (COUNT) {
    if (lane == 0) for (int q = 0; q < 8; ++q) if (xw[q] && q != 6) atomicAdd(&PC(counters)[q], xw[q]);
  }
}

// Next frame's job order from this frame's costs: expensive blocks first (64 linear cost classes; inside a class the
// order is whatever the atomics make it — no pixel depends on it); a block that cost more than 8x the mean (and more than
// a quarter of the dearest) is entered as four cooperative sub-block jobs.  One workgroup.
__global__ __launch_bounds__(1024) void rt_mesh_order(const unsigned int* cost, unsigned int* order, unsigned int* queue_len, int n_jobs,
                                                      int coop_all) {
  __shared__ unsigned int smax, hist[64], base[64];
  __shared__ unsigned long long ssum;
  const int tid = threadIdx.x;
  if (tid == 0) { smax = 0u; ssum = 0ull; }
  if (tid < 64) hist[tid] = 0u;
  __syncthreads();
  unsigned int m = 0u;
  unsigned long long sum = 0ull;
  for (int i = tid; i < n_jobs; i += 1024) { m = cost[i] > m ? cost[i] : m; sum += cost[i]; }
  atomicMax(&smax, m);
  atomicAdd(&ssum, sum);
  __syncthreads();
  const unsigned long long scale = (unsigned long long)smax + 1ull;
  const unsigned long long mean8 = 8ull * (ssum / (unsigned long long)(n_jobs > 0 ? n_jobs : 1));
  // coop_all (UOB_RT_MESH_COOP=1, tests): every block comes back cooperative
  const unsigned long long heavy = coop_all ? 0ull : (mean8 > (unsigned long long)smax / 4ull ? mean8 : (unsigned long long)smax / 4ull);
  for (int i = tid; i < n_jobs; i += 1024)
    atomicAdd(&hist[63 - (int)((unsigned long long)cost[i] * 64ull / scale)], (unsigned long long)cost[i] > heavy ? 4u : 1u);
  __syncthreads();
  if (tid == 0) {
    unsigned int at = 0u;
    for (int b = 0; b < 64; ++b) { base[b] = at; at += hist[b]; }
    queue_len[0] = at;
  }
  __syncthreads();
  for (int i = tid; i < n_jobs; i += 1024) {
    const int b = 63 - (int)((unsigned long long)cost[i] * 64ull / scale);
    if ((unsigned long long)cost[i] > heavy) {
      const unsigned int at = atomicAdd(&base[b], 4u);
      for (unsigned int q = 0; q < 4u; ++q) order[at + q] = ((unsigned int)i << 3) | 4u | q;
    } else {
      order[atomicAdd(&base[b], 1u)] = (unsigned int)i << 3;
    }
  }
}

// First frame of a context (no costs yet): a block's cost is guessed as the number of candidate tiles of its screen cell —
// the blocks on the mesh's silhouette are the expensive ones — so that the first frame, too, starts its long blocks first
// (and splits the dearest four ways) instead of meeting them in row order.  Scheduling only.
// (Adding a shadow side to the guess — the block's centre ray traced against the first tile, which holds the walls and the
// floor, and the candidate shadow tiles of the world cell its hit point starts from counted twice — was built and measured
// on configs[4]: first frame 14.7 ms against 9.5 ms with the primary side alone; the blocks in the mesh's shadow are not
// the dear ones, the ones on its silhouette are.  profiles/r03_mesh.txt)
__global__ void rt_mesh_estimate(const FrameParams P, unsigned int* cost, int n_jobs) {
  const int job = blockIdx.x * blockDim.x + threadIdx.x;
  if (job >= n_jobs) return;
  const int wgx_n = (P.W + 15) / 16, wg_rows = (P.owned_rows + 15) / 16, wg_mid = (wg_rows + 1) >> 1;
  const int job_y = job / wgx_n, job_x = job - job_y * wgx_n;
  const int wg_row = (job_y & 1) ? wg_mid + (job_y >> 1) : wg_mid - 1 - (job_y >> 1);
  const int lrr = wg_row * 16 < P.owned_rows ? wg_row * 16 : P.owned_rows - 1;
  const int cx = (job_x * 16) >> kScreenCellLog, cy = band_global_row(P, lrr) >> kScreenCellLog;
  unsigned int c = 1u;
  for (int w = 0; w < P.nwords; ++w) c += (unsigned int)__popcll(P.screen_masks[((size_t)cy * P.scx + cx) * P.nwords + w]);
  // A block that looks at a mirror or glass sphere sends bounce rays through every tile for up to `bounces` rounds: by far
  // the dearest blocks of a frame (configs[4] + the reference's spheres: a block on a sphere's rim 15 ms, the frame without
  // them 8 ms).  Counted as all tiles, twice: rt_mesh_order then enters them as cooperative jobs, whose four waves share each
  // round's tiles — in the first frame already (30 -> ms, profiles/r03_mesh.txt).
  if (P.bounces > 0) {
    const float xc = (float)(job_x * 16 + 8), yc = (float)(band_global_row(P, lrr + 8 < P.owned_rows ? lrr + 8 : P.owned_rows - 1));
    for (int i = 0; i < P.nsph; ++i) {
      if (P.sph[i].col[3] > 0.0f || !(P.sph[i].r2 > 0.0f)) continue;
      const f3 v = mk(P.sph[i].cx - P.cam[0], P.sph[i].cy - P.cam[1], P.sph[i].cz - P.cam[2]);
      // d = R^T v (the primary ray through pixel (x, y) is R (x aa_x - half_wx, (y aa_y - half_hy) sy, focal), rt_trace.h)
      const f3 q = mk(P.rot[0] * v.x + P.rot[4] * v.y + P.rot[8] * v.z, P.rot[1] * v.x + P.rot[5] * v.y + P.rot[9] * v.z,
                      P.rot[2] * v.x + P.rot[6] * v.y + P.rot[10] * v.z);
      if (!(q.z > 1e-6f)) continue;
      const float sx = (q.x / q.z * P.focal + P.half_wx) / (float)P.aa_x, sy_ = (q.y / q.z * P.focal / P.sy + P.half_hy) / (float)P.aa_y;
      const float rp = sqrtf(P.sph[i].r2) / q.z * P.focal / (float)P.aa_x * 1.1f + 12.0f;     // projected radius, generous
      if ((xc - sx) * (xc - sx) + (yc - sy_) * (yc - sy_) <= rp * rp) c += 2u * (unsigned int)(64 * P.nwords);
    }
  }
  cost[job] = c;
}

bool mesh_kernel_supports(const FrameParams& P) {
  const int aa = P.aa_x * P.aa_y;
  return P.records != nullptr && P.S >= 1 && P.S <= 4096 && aa >= 1 && aa <= 64 && P.n > 64 && P.spread >= 0.0f;
}

int mesh_tiles(int n) { return (n + kTile - 1) / kTile; }
int mesh_occ_words(int grid) { return occ_words(grid); }
int mesh_screen_cells(int pixels) { return (pixels + kScreenCell - 1) / kScreenCell; }

// P.records must hold this frame's records (launch_stage_records) before the masks are built.
// `aux`, `ev_fork`, `ev_join` (all nullable): a second stream and two events of the context, so that the primary-ray tile
// masks are built beside the world-cell occupancy + shadow-ray masks instead of in front of them.
void launch_mesh(const FrameParams& P, bool count, bool prof, hipStream_t stream, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
  const dim3 block(64 * kMeshWaves);
  const int ntiles = mesh_tiles(P.n), nwords = (ntiles + 63) / 64;
  if (P.screen_masks != nullptr) {
    hipMemsetAsync(P.screen_masks, 0, (size_t)P.scx * P.scy * nwords * 8, stream);
    hipMemsetAsync(P.world_masks, 0, (size_t)P.grid_g * P.grid_g * P.grid_g * nwords * 8, stream);
    hipMemsetAsync(P.world_occ, 0, (size_t)occ_words(P.grid_g) * sizeof(unsigned int), stream);
    const bool fork = aux != nullptr && ev_fork != nullptr && ev_join != nullptr &&
                      hipEventRecord(ev_fork, stream) == hipSuccess && hipStreamWaitEvent(aux, ev_fork, 0) == hipSuccess;
    hipLaunchKernelGGL(rt_bin_primary, dim3((ntiles + kMeshWaves - 1) / kMeshWaves, (P.scy + 3) / 4), block, 0, fork ? aux : stream, P);
    if (fork) hipEventRecord(ev_join, aux);
    hipLaunchKernelGGL(rt_bin_occupancy, dim3((P.n + P.nsph + kOccThreads - 1) / kOccThreads), dim3(kOccThreads), (size_t)occ_words(P.grid_g) * sizeof(unsigned int), stream, P);
    hipLaunchKernelGGL(rt_bin_shadow, dim3((ntiles + kMeshWaves - 1) / kMeshWaves, P.grid_g / 4, P.grid_g / 4), block, 0, stream, P);
    if (fork) hipStreamWaitEvent(stream, ev_join, 0);
  }
  const int n_jobs = ((P.W + 15) / 16) * ((P.owned_rows + 15) / 16);
  const int resident = P.mesh_blocks > 0 ? P.mesh_blocks : 256 * RT_MESH_MIN_BLOCKS;
  const dim3 grid(n_jobs < resident ? (n_jobs > 0 ? n_jobs : 1) : resident);
  hipMemsetAsync(P.job_counter, 0, sizeof(unsigned int), stream);
  const size_t lds_bytes = kBatch * kSlot * sizeof(float4) + kMeshWaves * (size_t)mesh_wave_lds_bytes(P.S > kPointSamples || P.nsph > 0) + 2 * (size_t)nwords * 8 +
                           kMeshWaves * 64 * (20 + 8 + 4);          // + the cooperative blocks' merge area
  FrameParams Q = P;
  if (!count && P.mesh_order == nullptr && P.mesh_cost != nullptr && P.mesh_order_out != nullptr && P.screen_masks != nullptr &&
      !(P.mask_debug & 16)) {                           // the context's first frame: order it by the guess (mask_debug 16: do not)
    hipLaunchKernelGGL(rt_mesh_estimate, dim3((n_jobs + 255) / 256), dim3(256), 0, stream, P, P.mesh_cost, n_jobs);
    hipLaunchKernelGGL(rt_mesh_order, dim3(1), dim3(1024), 0, stream, P.mesh_cost, P.mesh_order_out, P.mesh_queue_len, n_jobs, (P.mask_debug & 8) ? 1 : 0);
    Q.mesh_order = P.mesh_order_out;
  }
  if (!count && P.mesh_cost != nullptr) hipMemsetAsync(P.mesh_cost, 0, (size_t)n_jobs * 4, stream);
  if (count && prof) hipLaunchKernelGGL((rt_draw_mesh<false, true>), grid, block, lds_bytes, stream, Q);
  else if (count) hipLaunchKernelGGL((rt_draw_mesh<true>), grid, block, lds_bytes, stream, Q);
  // (instantiations specialised on (AA grid, samples) — <1,1,1> for configs[4], <2,2,10> — were built and measured: 7.98 ms
  // against 7.87 ms generic; this kernel is bound by the tile stream through LDS and its barriers, not by instruction issue)
  else hipLaunchKernelGGL((rt_draw_mesh<false>), grid, block, lds_bytes, stream, Q);
  if (!count && P.mesh_cost != nullptr && P.mesh_order_out != nullptr)
    hipLaunchKernelGGL(rt_mesh_order, dim3(1), dim3(1024), 0, stream, P.mesh_cost, P.mesh_order_out, P.mesh_queue_len, n_jobs, (P.mask_debug & 8) ? 1 : 0);
}

int mesh_blocks_per_cu() { return RT_MESH_MIN_BLOCKS; }

}  // namespace uobrt
