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
// (64-ray task, triangle) plus the few real tests.  Mirror / glass bounce rays (divergent directions) take
// the general per-lane loop over the HBM records.
#include "rt_wave_common.h"

namespace uobrt {

namespace {

constexpr int kTile = 64;                   // triangles per LDS tile
constexpr int kMeshWaves = 4;               // waves (= tasks) per workgroup sharing a tile

struct MeshWaveLds {
  float4* h0;   // start.xyz | radius_sq
  float4* h1;   // dir.xyz
  uint32_t* rng;
};
constexpr int kMeshWaveLdsBytes = 64 * 32 + kRngPixels * kRngStride * 4;

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
                                                unsigned long long sha, unsigned long long shb) {
  const float4 ha0 = L.h0[ja], ha1 = L.h1[ja], hb0 = L.h0[jb], hb1 = L.h1[jb];     // LDS broadcasts
  const f3 sa = mk(ha0.x, ha0.y, ha0.z), sb = mk(hb0.x, hb0.y, hb0.z);
  const float ra = ha0.w, rb = hb0.w;
  const f3 da = mk(ha1.x, ha1.y, ha1.z) + jit, db = mk(hb1.x, hb1.y, hb1.z) + jit;   // dir + crush(...), :333
  const f3 nda = -da, ndb = -db;
  int pos = 0;
  for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull, ++pos) {
    if (((need >> pos) & 1ull) == 0ull) continue;
    const int k = __builtin_ctzll(kk);
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

// xorshift streams of the GP pixels of RNG group g of the current task into the wave's scratch (:319,:331)
// Geometry of a wave's 8x8 pixel block: task k covers the PTx x PTy sub-block number k (row-major over the
// (8/PTx) x (8/PTy) grid of sub-blocks); pixel p of a task is (p % PTx, p / PTx) inside it.
struct BlockGeom {
  int x0, lr0;        // first pixel column / first packed local row of the 8x8 block
  int ptx_log, pty_log;
  __device__ __forceinline__ int PTx() const { return 1 << ptx_log; }
  __device__ __forceinline__ int PTy() const { return 1 << pty_log; }
  __device__ __forceinline__ int bx(int k, int p) const { return ((k & ((8 >> ptx_log) - 1)) << ptx_log) + (p & (PTx() - 1)); }
  __device__ __forceinline__ int by(int k, int p) const { return ((k >> (3 - ptx_log)) << pty_log) + (p >> ptx_log); }
};

__device__ __forceinline__ void generate_streams(const FrameParams& P, const MeshWaveLds& L, int lane, int GP, int NS,
                                                 const BlockGeom& B, int k, int first_p) {
  if (lane < 3 * GP) {
    const int pp = lane / 3, comp = lane % 3;
    const int px = B.x0 + B.bx(k, first_p + pp);
    const int py = band_global_row(B.lr0 + B.by(k, first_p + pp), P.band_rows, P.band_index, P.band_count);
    const int gid = pixel_global_id(P, px, py);
    const uint32_t seed = comp == 0 ? (uint32_t)gid : (uint32_t)((float)gid * (comp == 1 ? 91.0f : 19.0f));
    uint32_t s = xorshift(seed);
    uint32_t* dst = L.rng + pp * kRngStride + comp;
    for (int it = 0; it < NS; ++it) { s = xorshift(s); dst[it * 4] = s; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// Grid: x = ceil(W/16), y = ceil(owned_rows/16); block = 4 waves = 2x2 blocks of 8x8 pixels.
__global__ __launch_bounds__(64 * kMeshWaves) void rt_draw_mesh(const FrameParams P) {
  extern __shared__ float4 lds[];
  float4* tile = lds;                                   // 8 records x kTile triangles
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const MeshWaveLds L{reinterpret_cast<float4*>(reinterpret_cast<char*>(lds + 8 * kTile) + wave * kMeshWaveLdsBytes),
                      reinterpret_cast<float4*>(reinterpret_cast<char*>(lds + 8 * kTile) + wave * kMeshWaveLdsBytes + 64 * 16),
                      reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds + 8 * kTile) + wave * kMeshWaveLdsBytes + 64 * 32)};
  const int n = P.n, ntiles = (n + kTile - 1) / kTile;
  const LdsScene G = lds_scene(P.records, n);           // the whole mesh, in HBM (hit finalisation, bounce rays)
  const float4 *t_v0 = tile, *t_e1 = tile + kTile, *t_e2 = tile + 2 * kTile, *t_c = tile + 3 * kTile,
               *t_col = tile + 5 * kTile, *t_pc = tile + 6 * kTile, *t_qc = tile + 7 * kTile;

  const int aa = P.aa_x * P.aa_y;                       // a power of two <= 64 (mesh_kernel_supports())
  const int la = __builtin_ctz(aa);
  const int PT = 64 >> la;                              // pixels per task
  BlockGeom B;
  B.x0 = blockIdx.x * 16 + (wave & 1) * 8;
  B.lr0 = blockIdx.y * 16 + (wave >> 1) * 8;            // waves past the frame still walk the tiles (barriers)
  B.ptx_log = (6 - la + 1) >> 1;                        // PTx >= PTy, PTx * PTy = PT
  B.pty_log = (6 - la) - B.ptx_log;
  const int GP = PT < kRngPixels ? PT : kRngPixels;
  const int GL = GP * aa;
  const f3 light = mk(P.light[0], P.light[1], P.light[2]);
  const float hbox = P.spread / 2.f;
  const int NS = P.S;
  const unsigned long long active = NS == 64 ? ~0ull : ((1ull << NS) - 1ull);
  Work wk;

  // cooperative tile load: 8 records x 64 triangles = 512 float4, two per thread
  auto load_tile = [&](int t) {
    __syncthreads();
    for (int r = tid; r < 8 * kTile; r += 64 * kMeshWaves) {
      const int rec = r >> 6, gi = t * kTile + (r & 63);
      tile[r] = gi < n ? P.records[(size_t)rec * n + gi] : make_float4(0.f, 0.f, 0.f, -1.0f);
    }
    __syncthreads();
  };

  f3 outc = mk(0.f, 0.f, 0.f);
  for (int k = 0; k < aa; ++k) {
    // ---- phase 1: primary rays over all tiles -------------------------------------------------------
    const int p = lane >> la;                   // pixel of this lane within the task
    const int a = lane & (aa - 1);
    const int x = B.x0 + B.bx(k, p);
    const int lr = B.lr0 + B.by(k, p);
    const bool valid = lr < P.owned_rows && x < P.W;
    const int y = band_global_row(lr < P.owned_rows ? lr : 0, P.band_rows, P.band_index, P.band_count);
    Ray ray = primary_ray(P, x, y, a % P.aa_x, a / P.aa_x);
    f3 duc, eu;
    float dumax;
    {
      // sub-pixel rectangle of the task: columns are contiguous; rows are the task's PTy packed rows, whose
      // global y may jump at a band boundary, so take their min and max
      const float Xlo = (float)((B.x0 + B.bx(k, 0)) * P.aa_x) - ((float)P.W * (float)P.aa_x) / 2.0f;
      int ymin = 0x7fffffff, ymax = 0;
      for (int r = 0; r < B.PTy(); ++r) {
        const int lrr = B.lr0 + B.by(k, 0) + r;
        const int yy = band_global_row(lrr < P.owned_rows ? lrr : (P.owned_rows > 0 ? P.owned_rows - 1 : 0), P.band_rows,
                                       P.band_index, P.band_count);
        ymin = yy < ymin ? yy : ymin; ymax = yy > ymax ? yy : ymax;
      }
      const float Ylo = ((float)(ymin * P.aa_y) - ((float)P.H * (float)P.aa_y) / 2.0f) * P.sy;
      const float Yhi = ((float)(ymax * P.aa_y + P.aa_y - 1) - ((float)P.H * (float)P.aa_y) / 2.0f) * P.sy;
      const float hx = 0.5f * (float)(B.PTx() * P.aa_x - 1), hy = 0.5f * (Yhi - Ylo);
      const f3 wc = mk(Xlo + hx, Ylo + hy, P.focal);
      const f3 r0 = mk(P.rot[0], P.rot[1], P.rot[2]), r1 = mk(P.rot[4], P.rot[5], P.rot[6]),
               r2 = mk(P.rot[8], P.rot[9], P.rot[10]);
      duc = mk(dot3(r0, wc), dot3(r1, wc), dot3(r2, wc));
      eu = mk(1.0001f * (fabsf(r0.x) * hx + fabsf(r0.y) * hy), 1.0001f * (fabsf(r1.x) * hx + fabsf(r1.y) * hy),
              1.0001f * (fabsf(r2.x) * hx + fabsf(r2.y) * hy));
      dumax = fmaxf(fmaxf(fabsf(duc.x) + eu.x, fabsf(duc.y) + eu.y), fabsf(duc.z) + eu.z);
    }
    float current_t = RT_MAXFLOAT, bu = 0.f, bv = 0.f;
    int best = -1;
    const f3 ndp = -ray.dir;
    for (int t = 0; t < ntiles; ++t) {
      load_tile(t);
      const int nc = (n - t * kTile) < kTile ? (n - t * kTile) : kTile;
      unsigned long long Kp = nc == 64 ? ~0ull : ((1ull << nc) - 1ull);
      {
        const float4 c4 = t_c[lane];
        const bool clear = primary_clear(duc, eu, dumax, xyz(c4), c4.w, xyz(t_pc[lane]), xyz(t_qc[lane]));
        if (dumax < 1e30f) Kp &= ~ballot(clear);
      }
      if (valid) {
        for (unsigned long long m = Kp; m != 0ull; m &= m - 1ull) {
          const int i = __builtin_ctzll(m);
          const float4 c4 = t_c[i];
          const float detA_recip = rcp_exact(detc(ndp, xyz(c4)));
          const float tt = c4.w * detA_recip;
          const float u = detc(ndp, xyz(t_pc[i])) * detA_recip;
          const float v = detc(ndp, xyz(t_qc[i])) * detA_recip;
          if (tt < current_t && u >= 0 && v >= 0 && (u + v) <= 1 && tt >= 0) {
            best = t * kTile + i; bu = u; bv = v; current_t = tt;
          }
        }
      }
    }
    bool lit = false, secondary = false;
    if (valid) {
      if (best >= 0) {
        ray.tri = best;
        ray.P = (xyz(G.v0[best]) + bu * xyz(G.e1[best])) + bv * xyz(G.e2[best]);
        ray.N = xyz(G.nrm[best]);
        ray.col = G.col[best];
      }
      closest_spheres<false>(P, ray, current_t, wk);
      if (ray.tri != -1) {
        if (ray.col.w <= 0.0f) { secondary = true; lit = bounce_to_diffuse<false>(G, P, ray, wk); }
        else lit = true;
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

    // ---- phase 3: shadows over all tiles ---------------------------------------------------------------
    const unsigned long long litmask = ballot(lit);
    const float dlen = sqrtf(radius_sq);
    const float hh = 1.002f * hbox + 2e-6f * (dlen + hbox);
    float dminlen = dlen - 1.7321f * hh;
    const bool sane = lit && (radius_sq > 1e-18f) && (radius_sq < 1e30f);
    if (!sane || !(dminlen > 0.0f)) dminlen = 0.0f;
    const float dk = dlen * 1.000004f;
    const unsigned long long sphmask = ballot(lit && P.nsph > 0 && spheres_maybe(P, start, dir, dlen, hh));
    f3 s0 = mk(0.f, 0.f, 0.f), D0 = mk(0.f, 0.f, 0.f);
    float es = 0.f, ed = 0.f, dlen_max = 0.f, dlen_min = 0.f, hh_task = 0.f;
    bool task_ok = false;
    if (litmask != 0ull) {
      const int jr = __builtin_ctzll(litmask);
      s0 = mk(rl(start.x, jr), rl(start.y, jr), rl(start.z, jr));
      D0 = mk(rl(dir.x, jr), rl(dir.y, jr), rl(dir.z, jr));
      const f3 ds = start - s0, dd = dir - D0;
      es = wave_max(lit ? fmaxf(fmaxf(fabsf(ds.x), fabsf(ds.y)), fabsf(ds.z)) : 0.0f);
      ed = wave_max(lit ? fmaxf(fmaxf(fabsf(dd.x), fabsf(dd.y)), fabsf(dd.z)) : 0.0f);
      dlen_max = wave_max(lit ? dlen : 0.0f);
      dlen_min = wave_min(lit ? dlen : 3.0e38f);
      task_ok = (ballot(lit && !sane) == 0ull) && es < 1e30f && ed < 1e30f;
      hh_task = 1.002f * hbox + 2e-6f * (dlen_max + hbox);
    }
    unsigned long long my_sh = 0ull;            // blocked samples of THIS lane's surface point, across tiles
    bool blocked = false, task_blocked = false;
    int rng_group = -1;                         // which pixel group's streams the scratch currently holds
    for (int t = 0; t < ntiles; ++t) {
      load_tile(t);
      if (litmask == 0ull || task_blocked) continue;            // wave-uniform; the barriers are behind us
      const int nc = (n - t * kTile) < kTile ? (n - t * kTile) : kTile;
      const unsigned long long casts = ballot(lane < nc && t_col[lane].w != -1.0f);    // glass casts no shadow, :247
      unsigned long long K = casts;
      if (task_ok) {                                               // level 1, lane = triangle
        TriLane T1;
        T1.v0 = xyz(t_v0[lane]); T1.e1 = xyz(t_e1[lane]); T1.e2 = xyz(t_e2[lane]); T1.c = xyz(t_c[lane]);
        T1.c1 = norm1(T1.c); T1.e1_1 = norm1(T1.e1); T1.e2_1 = norm1(T1.e2);
        const Bound tb = task_bound(T1, s0, D0, es, ed, hh_task, dlen_min, dlen_max);
        K = casts & ~ballot(tb.clear);
        if ((casts & ballot(tb.all_blocked)) != 0ull) { task_blocked = true; continue; }
      }
      if (K == 0ull) continue;
      unsigned long long need = 0ull;                              // level 2, lane = surface point
      int pos = 0;
      for (unsigned long long kk = K; kk != 0ull; kk &= kk - 1ull, ++pos) {
        const int kq = __builtin_ctzll(kk);
        const Bound pb = point_bound(start, dir, hh, dlen, dminlen, dk, xyz(t_v0[kq]), xyz(t_e1[kq]), xyz(t_e2[kq]), xyz(t_c[kq]));
        if (!pb.clear || !sane) need |= 1ull << pos;
        blocked = blocked || (sane && pb.all_blocked);
      }
      const unsigned long long work = ballot(lit && !blocked && need != 0ull && (my_sh & active) != active);
      for (int g = 0; g * GL < 64 && work != 0ull; ++g) {          // level 3, lane = shadow sample
        const unsigned long long gm = (work >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
        if (gm == 0ull) continue;
        if (rng_group != g) { generate_streams(P, L, lane, GP, NS, B, k, g * GP); rng_group = g; }
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
                                            active, readlane64(my_sh, j), readlane64(my_sh, j2));
            if (lane == j2) my_sh = m2.b;
            if (lane == j) my_sh = m2.a;
          }
        }
      }
    }
    // shadow-casting spheres (kernels.cl:278-307) for the points whose rays can reach one
    {
      const unsigned long long sw = ballot(lit && !blocked && !task_blocked && ((sphmask >> lane) & 1ull) != 0ull &&
                                           (my_sh & active) != active);
      for (int g = 0; g * GL < 64 && sw != 0ull; ++g) {
        const unsigned long long gm = (sw >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
        if (gm == 0ull) continue;
        if (rng_group != g) { generate_streams(P, L, lane, GP, NS, B, k, g * GP); rng_group = g; }
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
    const int unshadowed = (blocked || task_blocked) ? 0 : __popcll(active & ~my_sh);

    // ---- phase 4: shading and the AA sum, as in rt_kernel_wave.hip ---------------------------------------
    f3 contrib = mk(0.f, 0.f, 0.f);
    if (lit) {
      float total = 0.0f;
      if (unshadowed < NS) total += 0.0f * term;
      for (int i = 0; i < NS; ++i) if (i < unshadowed) total += term;
      const float l = 0.5f + total / (float)NS;
      if (secondary) { const float kk = 0.9f * l; contrib = mk(kk * ray.col.x, kk * ray.col.y, kk * ray.col.z); }
      else contrib = mk(ray.col.x * l, ray.col.y * l, ray.col.z * l);
    }
    f3 acc = mk(0.f, 0.f, 0.f);
    const int first = (lane >> la) << la;
    for (int r = 0; r < aa; ++r)
      acc = acc + mk(shfl(contrib.x, first + r), shfl(contrib.y, first + r), shfl(contrib.z, first + r));
    {   // output lane l owns block pixel (l & 7, l >> 3): take its sum from the task and pixel that cover it
      const int obx = lane & 7, oby = lane >> 3;
      const int ok = ((oby >> B.pty_log) << (3 - B.ptx_log)) + (obx >> B.ptx_log);
      const int op = ((oby & (B.PTy() - 1)) << B.ptx_log) + (obx & (B.PTx() - 1));
      const f3 v = mk(shfl(acc.x, op << la), shfl(acc.y, op << la), shfl(acc.z, op << la));
      if (ok == k) outc = v;
    }
  }
  const int x = B.x0 + (lane & 7);
  const int lr = B.lr0 + (lane >> 3);
  if (lr < P.owned_rows && x < P.W) {
    const float inv = (float)aa;
    const f3 c = mk(outc.x / inv, outc.y / inv, outc.z / inv);
    const size_t o = (size_t)lr * P.W + x;
    P.out_argb[o] = pack_argb(c);
    if (P.out_rgb) P.out_rgb[o] = make_float4(c.x, c.y, c.z, 1.0f);
  }
}

bool mesh_kernel_supports(const FrameParams& P) {
  const int aa = P.aa_x * P.aa_y;
  return P.records != nullptr && P.S >= 1 && P.S <= 64 && aa >= 1 && aa <= 64 && (64 % aa) == 0 && P.n > 64 &&
         P.spread >= 0.0f;
}

void launch_mesh(const FrameParams& P, hipStream_t stream) {
  const dim3 block(64 * kMeshWaves);
  const dim3 grid((P.W + 15) / 16, (P.owned_rows + 15) / 16);
  const size_t lds_bytes = 8 * kTile * sizeof(float4) + kMeshWaves * (size_t)kMeshWaveLdsBytes;
  hipLaunchKernelGGL(rt_draw_mesh, grid, block, lds_bytes, stream, P);
}

}  // namespace uobrt
