// TEST INFRASTRUCTURE ONLY — sanitizer build of the CPU-side code ("Race detection / sanitizers":
// -fsanitize=address,undefined on the CPU restatement; the GPU pool offers no device sanitizer).
//   make -C oracle asan   builds oracle/asan_check from rt_oracle.c + uob_raytracer_amd/csrc/scene.cpp, both instrumented
//   oracle/asan_check <dir>   (1) renders a 64x48 frame of the Cornell Box with the oracle (reference constants + glass and
//                             mirror spheres: every function of the restatement runs), prints its FNV-1a-64;
//                             (2) feeds the OBJ reader a file with a 5000-character `f` record, a 70-gon, slash tokens,
//                             relative indices, an out-of-range index, a record cut short.
// Any sanitizer report aborts with a non-zero exit (halt_on_error); tests/test_asan.py runs it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/uob_rt.h"

extern "C" int rto_render(const rt_config* cfg, const float* verts4, const float* normals4, const float* colors4, int n,
                          const float* rot12, const float* cam3, const float* light3, float focal, uint32_t* out_argb,
                          float* out_rgb, const int* pix, long npix, int nthreads, rt_work* work);

namespace uobrt {
void set_error(const char*, ...) {}          // scene.cpp reports through the product's error string; not needed here
}

static int fail(const char* what) { fprintf(stderr, "asan_check: %s\n", what); return 1; }

int main(int argc, char** argv) {
  if (argc != 2) return fail("usage: asan_check <scratch-dir>");
  // ---- (1) the oracle ----------------------------------------------------------------------------------------------
  std::vector<rt_triangle> tris(64);
  const int n = rt_scene_cornell_box(tris.data(), 64);
  if (n != 26) return fail("cornell box");
  std::vector<float> v(12 * n), nr(4 * n), col(4 * n);
  rt_scene_pack(tris.data(), n, v.data(), nr.data(), col.data());
  rt_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.width = 64; cfg.height = 48; cfg.aa_x = 2; cfg.aa_y = 2; cfg.shadow_samples = 10; cfg.light_spread = 0.05f;
  cfg.max_bounces = 10; cfg.num_spheres = 2;
  const rt_sphere glass = {{0.3f, 0.1f, -0.5f}, 0.075f, {0.f, 0.f, 0.f, -1.f}}, mirror = {{-0.4f, 0.8f, -0.5f}, 0.05f, {0.f, 0.f, 0.f, 0.f}};
  cfg.spheres[0] = glass; cfg.spheres[1] = mirror;
  cfg.band_rows = 48; cfg.band_index = 0; cfg.band_count = 1; cfg.device = -1;
  float rot[12];
  rt_rotation_matrix(0.2f, -0.1f, rot);
  const float cam[3] = {0.1f, 0.0f, -3.0f}, light[3] = {-0.2f, -0.5f, -0.7f};
  std::vector<uint32_t> argb(64 * 48);
  std::vector<float> rgb(3 * 64 * 48);
  rt_work work;
  if (rto_render(&cfg, v.data(), nr.data(), col.data(), n, rot, cam, light, 1100.0f * 48 / 1024 * 2, argb.data(), rgb.data(),
                 nullptr, 0, 2, &work) != RT_OK) return fail("rto_render");
  uint64_t h = 1469598103934665603ull;
  for (uint32_t w : argb) { h ^= w; h *= 1099511628211ull; }
  printf("oracle 64x48 fnv %016llx primary %llu bounce %llu shadow %llu\n", (unsigned long long)h,
         (unsigned long long)work.primary_rays, (unsigned long long)work.bounce_rays, (unsigned long long)work.shadow_rays);

  // ---- (2) the OBJ reader on hostile input ------------------------------------------------------------------------------
  const std::string path = std::string(argv[1]) + "/hostile.obj";
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return fail("cannot write the OBJ");
  for (int i = 0; i < 80; ++i) fprintf(f, "v %d.5 %d.25 %d\n", i % 9, i % 7, i % 5);
  fprintf(f, "f");                                              // a 70-gon: 68 triangles
  for (int i = 1; i <= 70; ++i) fprintf(f, " %d", i);
  fprintf(f, "\nf");                                            // > 5000 characters: slash tokens padded with texture / normal indices
  for (int i = 1; i <= 75; ++i) fprintf(f, " %d/%030d/%030d", i, i, i);
  fprintf(f, "\nf -1 -2 -3\nf 1//2 2//3 3//4\n");
  fclose(f);
  const int m = rt_scene_load_obj(path.c_str(), nullptr, 0);
  if (m != 68 + 73 + 1 + 1) { fprintf(stderr, "asan_check: OBJ gave %d triangles\n", m); return 1; }
  std::vector<rt_triangle> mesh(m);
  if (rt_scene_load_obj(path.c_str(), mesh.data(), m) != m) return fail("second OBJ pass");
  // fewer slots than triangles: only `cap` are written
  std::vector<rt_triangle> few(5);
  if (rt_scene_load_obj(path.c_str(), few.data(), 5) != m) return fail("capped OBJ pass");
  if (memcmp(few.data(), mesh.data(), 5 * sizeof(rt_triangle)) != 0) return fail("capped pass differs");
  const char* bad[] = {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n", "v 0 0\n", "v 0 0 0\nf 1 x 3\n"};
  for (const char* text : bad) {
    f = fopen(path.c_str(), "w");
    fputs(text, f);
    fclose(f);
    if (rt_scene_load_obj(path.c_str(), nullptr, 0) != RT_E_IO) return fail("a malformed OBJ was accepted");
  }
  printf("obj reader: %d triangles from the hostile file, 4 malformed files rejected\n", m);
  return 0;
}
