// TEST INFRASTRUCTURE ONLY.  Compiles the reference's scene builders *in place*
// (/root/reference/Source/TestModelH.h:44 LoadTestModel, /root/reference/Source/Loader.cpp:11
// load_obj, with the vendored /root/reference/glm headers) and exports their output as flat floats
// so tests can pin the product's own scene generator bit-for-bit.  Output: oracle/_ref/libref_scene.so.
#include "Loader.cpp"   // resolved by -I/root/reference/Source ; itself includes TestModelH.h

static int flatten(const std::vector<Triangle>& t, float* out, int cap) {
  const int n = (int)t.size();
  for (int i = 0; i < n && i < cap; ++i) {
    const glm::vec4 v[5] = {t[i].v0, t[i].v1, t[i].v2, t[i].normal, t[i].color};
    for (int k = 0; k < 5; ++k) { out[20 * i + 4 * k] = v[k].x; out[20 * i + 4 * k + 1] = v[k].y; out[20 * i + 4 * k + 2] = v[k].z; out[20 * i + 4 * k + 3] = v[k].w; }
  }
  return n;
}
// AoS, 20 floats per triangle: v0 v1 v2 normal color (TestModelH.h:14-18)
extern "C" int ref_load_test_model(float* out, int cap) {
  std::vector<Triangle> t; LoadTestModel(t); return flatten(t, out, cap);
}
extern "C" int ref_load_obj(const char* path, float* out, int cap) {
  std::vector<Triangle> t = load_obj(path); return flatten(t, out, cap);
}
