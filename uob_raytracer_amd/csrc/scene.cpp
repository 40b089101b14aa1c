// scene.cpp — host-side scene format of the ray tracer (C ABI: rt_scene_*, rt_triangle_*, rt_rotation_matrix).
//
// Replaces, behind include/uob_rt.h, what the reference builds on the host before the device boundary:
//   * the Triangle AoS and ComputeNormal      (Source/TestModelH.h:11-38)
//   * LoadTestModel, the 26-triangle Cornell Box (Source/TestModelH.h:44-219)
//   * load_obj                                   (Source/Loader.cpp:11-59)
//   * the AoS -> packed float4 upload format     (Source/skeleton.cpp:474-484)
//   * rot_matrix[12]                             (Source/skeleton.cpp:149-151)
// Re-designed as data tables (corner sets + one face list) rather than a push_back script; the float
// arithmetic keeps the reference's (GLM 0.9.7.2) operation order so the packed buffers are bit-identical
// (tests pin their FNV-1a hashes).  Built with -ffp-contract=off.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uob_rt.h"

namespace uobrt {
void set_error(const char* fmt, ...);
}

namespace {

struct P3 { float x, y, z; };

// Material colours (TestModelH.h:50-62); w is the material flag (>0 diffuse, 0 mirror, <0 glass)
const float kDarkGrey[4]   = {0.25f, 0.25f, 0.25f, 1.0f};
const float kDarkPurple[4] = {0.25f, 0.0f, 0.25f, 1.0f};
const float kDarkGreen[4]  = {0.0f, 0.25f, 0.0f, 1.0f};
const float kDarkYellow[4] = {0.3f, 0.3f, 0.0f, 1.0f};
const float kWhite[4]      = {0.75f, 0.75f, 0.75f, 1.0f};
const float kRed[4]        = {0.6f, 0.0f, 0.0f, 1.0f};
const float kBlue[4]       = {0.0f, 0.2f, 0.5f, 1.0f};
const float kObjBlue[4]    = {0.0f, 0.2f, 0.4f, 0.5f};   // Loader.cpp:20

// Corner naming of a box, as in TestModelH.h: A B C D on the floor (y=0), E F G H above them.
enum { cA, cB, cC, cD, cE, cF, cG, cH };

struct Face { int a, b, c; const float* color; };

// Room: 10 triangles (TestModelH.h:84-103; the front wall is commented out in the reference)
const Face kRoomFaces[] = {
    {cC, cB, cA, kDarkGrey},   {cC, cD, cB, kDarkGrey},      // floor
    {cA, cE, cC, kDarkPurple}, {cC, cE, cG, kDarkPurple},    // left wall
    {cF, cB, cD, kDarkGreen},  {cH, cF, cD, kDarkGreen},     // right wall
    {cE, cF, cG, kDarkYellow}, {cF, cH, cG, kDarkYellow},    // ceiling
    {cG, cD, cC, kWhite},      {cG, cH, cD, kWhite},         // back wall
};
// A block: 8 triangles (front, right, left, top; back and bottom are commented out, TestModelH.h:128-152)
const int kBlockFaces[][3] = {
    {cE, cB, cA}, {cE, cF, cB}, {cF, cD, cB}, {cF, cH, cD}, {cG, cE, cC}, {cE, cA, cC}, {cG, cF, cE}, {cG, cH, cF},
};

void box_corners(const float floor_xz[4][2], float height, P3 out[8]) {
  for (int k = 0; k < 4; ++k) {
    out[k] = P3{floor_xz[k][0], 0.0f, floor_xz[k][1]};
    out[4 + k] = P3{floor_xz[k][0], height, floor_xz[k][1]};
  }
}

void set_tri(rt_triangle* t, P3 a, P3 b, P3 c, const float* color) {
  const P3 v[3] = {a, b, c};
  float* dst[3] = {t->v0, t->v1, t->v2};
  for (int k = 0; k < 3; ++k) { dst[k][0] = v[k].x; dst[k][1] = v[k].y; dst[k][2] = v[k].z; dst[k][3] = 1.0f; }
  memcpy(t->color, color, sizeof t->color);
}

}  // namespace

extern "C" {

// TestModelH.h:26-35 with GLM's cross (func_geometric.inl:134-142) and normalize = x * (1/sqrt(dot))
// (func_geometric.inl:154-159, func_exponential.inl:150-153; dot = x*x + y*y + z*z left to right).
void rt_triangle_compute_normal(rt_triangle* t) {
  const float e1x = t->v1[0] - t->v0[0], e1y = t->v1[1] - t->v0[1], e1z = t->v1[2] - t->v0[2];
  const float e2x = t->v2[0] - t->v0[0], e2y = t->v2[1] - t->v0[1], e2z = t->v2[2] - t->v0[2];
  // cross(e2, e1)
  const float nx = e2y * e1z - e1y * e2z;
  const float ny = e2z * e1x - e1z * e2x;
  const float nz = e2x * e1y - e1x * e2y;
  const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
  t->normal[0] = nx * inv; t->normal[1] = ny * inv; t->normal[2] = nz * inv; t->normal[3] = 1.0f;
}

int rt_scene_cornell_box(rt_triangle* out, int32_t cap) {
  if (!out || cap < 0) { uobrt::set_error("rt_scene_cornell_box: bad arguments"); return RT_E_INVALID; }
  const float L = 555;   // side of the Cornell Box, TestModelH.h:69
  const float room_xz[4][2] = {{L, 0}, {0, 0}, {L, L}, {0, L}};
  const float short_xz[4][2] = {{290, 114}, {130, 65}, {240, 272}, {82, 225}};    // TestModelH.h:112-115
  const float tall_xz[4][2] = {{423, 247}, {265, 296}, {472, 406}, {314, 456}};   // TestModelH.h:159-162
  std::vector<rt_triangle> tris;
  P3 c[8];
  box_corners(room_xz, L, c);
  for (const Face& f : kRoomFaces) { rt_triangle t; set_tri(&t, c[f.a], c[f.b], c[f.c], f.color); tris.push_back(t); }
  box_corners(short_xz, 165, c);
  for (const auto& f : kBlockFaces) { rt_triangle t; set_tri(&t, c[f[0]], c[f[1]], c[f[2]], kRed); tris.push_back(t); }
  box_corners(tall_xz, 330, c);
  for (const auto& f : kBlockFaces) { rt_triangle t; set_tri(&t, c[f[0]], c[f[1]], c[f[2]], kBlue); tris.push_back(t); }

  // Scale to [-1,1]^3 and flip x,y (TestModelH.h:195-218): v*(2/L) - 1, then negate x and y.
  const float s = 2 / L;
  for (rt_triangle& t : tris) {
    float* vs[3] = {t.v0, t.v1, t.v2};
    for (float* v : vs) {
      v[0] = -(v[0] * s - 1.0f);
      v[1] = -(v[1] * s - 1.0f);
      v[2] = v[2] * s - 1.0f;
      v[3] = 1.0f;
    }
    rt_triangle_compute_normal(&t);
  }
  const int n = (int)tris.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = tris[i];
  return n;
}

// Loader.cpp:11-59: `v x y z` and `f a b c` lines (1-based indices); every other line is ignored.  Beyond the
// reference's parser, `f` also takes "i/t/n" / "i//n" tokens, negative (relative) indices and polygons (below).
// The constants the reference hard-codes — colour blue (0,0.2,0.4,0.5) :20, scale 1.5 :42, translation
// (-0.4,1.15,-0.7) :48-52 — are the defaults of the _ex form.
int rt_scene_load_obj(const char* path, rt_triangle* out, int32_t cap) {
  return rt_scene_load_obj_ex(path, nullptr, 1.5f, nullptr, out, cap);
}

int rt_scene_load_obj_ex(const char* path, const float color[4], float scale, const float translate[3],
                         rt_triangle* out, int32_t cap) {
  if (!path || (!out && cap > 0)) { uobrt::set_error("rt_scene_load_obj: bad arguments"); return RT_E_INVALID; }
  const float kObjMove[3] = {-0.4f, 1.15f, -0.7f};                 // Loader.cpp:48
  const float* const col = color ? color : kObjBlue;
  const float* const mv = translate ? translate : kObjMove;
  FILE* f = fopen(path, "r");
  if (!f) { uobrt::set_error("rt_scene_load_obj: cannot open %s", path); return RT_E_IO; }
  std::vector<P3> verts;
  int n = 0, rc = RT_OK;
  // Lines of any length (an n-gon's `f` record can be thousands of characters): a fixed buffer would cut a record in two
  // and read the second half as a record of its own, indices split mid-number.
  std::string linebuf;
  char chunk[1024];
  long lineno = 0;
  for (;;) {
    linebuf.clear();
    bool got = false;
    while (fgets(chunk, sizeof chunk, f)) {
      got = true;
      linebuf += chunk;
      if (!linebuf.empty() && linebuf.back() == '\n') break;
    }
    if (!got) break;
    const char* const line = linebuf.c_str();
    ++lineno;
    char tag[8] = {0};
    int used = 0;
    if (sscanf(line, "%7s%n", tag, &used) != 1) continue;
    if (!strcmp(tag, "v")) {
      float x, y, z;
      if (sscanf(line + used, "%f %f %f", &x, &y, &z) != 3) { rc = RT_E_IO; break; }
      verts.push_back(P3{scale * x, scale * y, scale * z});     // Loader.cpp:42 (1.5f * v)
    } else if (!strcmp(tag, "f")) {
      // Loader.cpp:44-45 reads three plain 1-based indices.  Hardening beyond the reference (which reads
      // garbage there): "i/t/n" and "i//n" tokens (the vertex index is taken), negative = relative indices,
      // and polygons with more than three corners, fan-triangulated in order.
      std::vector<int> idx;
      int k = 0;
      const int nv = (int)verts.size();
      const char* p = line + used;
      bool bad = false;
      for (;;) {
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '\0' || *p == '\n' || *p == '\r' || *p == '#') break;
        char* end = nullptr;
        long v = strtol(p, &end, 10);
        if (end == p) { bad = true; break; }
        p = end;
        while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') {      // "/t/n" suffix
          if (*p != '/' && *p != '-' && (*p < '0' || *p > '9')) { bad = true; break; }
          ++p;
        }
        if (bad) break;
        if (v < 0) v = nv + 1 + v;
        if (v < 1 || v > nv) { bad = true; break; }
        idx.push_back((int)v); ++k;
      }
      if (bad || k < 3) { rc = RT_E_IO; break; }
      for (int j = 1; j + 1 < k; ++j) {
        const int a = idx[0], b = idx[j], c = idx[j + 1];
        if (n < cap) {
          rt_triangle t;
          set_tri(&t, verts[a - 1], verts[b - 1], verts[c - 1], col);
          rt_triangle_compute_normal(&t);                          // normal of the UN-negated triangle, :46
          float* vs[3] = {t.v0, t.v1, t.v2};
          for (float* v : vs) {                                    // (-1)*v + (-0.4, 1.15, -0.7, 1), :48-52
            v[0] = (-1.f) * v[0] + mv[0];
            v[1] = (-1.f) * v[1] + mv[1];
            v[2] = (-1.f) * v[2] + mv[2];
            v[3] = (-1.f) * v[3] + 1.0f;
          }
          out[n] = t;
        }
        ++n;
      }
    }
  }
  fclose(f);
  if (rc != RT_OK) { uobrt::set_error("rt_scene_load_obj: %s:%ld: malformed or out-of-range record", path, lineno); return rc; }
  return n;
}

// skeleton.cpp:474-484
void rt_scene_pack(const rt_triangle* tris, int32_t n, float* vertices4, float* normals4, float* colors4) {
  for (int i = 0; i < n; ++i) {
    const float* vs[3] = {tris[i].v0, tris[i].v1, tris[i].v2};
    for (int k = 0; k < 3; ++k) {
      float* d = vertices4 + (size_t)(3 * i + k) * 4;
      d[0] = vs[k][0]; d[1] = vs[k][1]; d[2] = vs[k][2]; d[3] = 0.0f;
    }
    float* nr = normals4 + (size_t)i * 4;
    nr[0] = tris[i].normal[0]; nr[1] = tris[i].normal[1]; nr[2] = tris[i].normal[2]; nr[3] = 0.0f;
    memcpy(colors4 + (size_t)i * 4, tris[i].color, 16);
  }
}

// skeleton.cpp:149-151 (float cos/sin of float yaw/pitch)
void rt_rotation_matrix(float yaw, float pitch, float rot[12]) {
  const float cy = cosf(yaw), sy = sinf(yaw), cp = cosf(pitch), sp = sinf(pitch);
  const float m[12] = {cy, sp * sy, sy * cp, 0.0f, 0.0f, cp, -sp, 0.0f, -sy, cy * sp, cp * cy, 0.0f};
  memcpy(rot, m, sizeof m);
}

}  // extern "C"
