// skeleton_host.cpp — the reference's application loop (Source/skeleton.cpp:93-144) over libuob_rt.so.
//
// Same globals and function names as the reference: focal_length, camera_position, yaw, pitch,
// light_position (skeleton.cpp:61-67), update() (:282-361), draw via offload_rendering() (:146-182),
// opencl_initialise() (:366-497) — the last two now four lines each over the C ABI.
// SDL events do not exist headless: update() keeps the light animation (:290-298) bit for bit and takes its events from
// an optional script that stands for SDL's event queue: key names ("up down left right i o k j esc", :311-352), mouse
// motion "m:dx,dy" (SDL_MOUSEMOTION xrel / yrel, :306-309) and "." (the queue is empty for the rest of this frame).
// As in the reference's polling loop a frame consumes mouse events until it meets a key (handled, then update()
// returns: what is left waits for the next frame), a "." or the end of the script.
//
//   uob_raytracer [--size N] [--frames K] [--aa X Y] [--shadows S] [--keys "left m:12,-3 . left i"] [--out file.bmp]
//                 [--obj mesh.obj]            append load_obj(mesh.obj) to the box, as skeleton.cpp:102-103 does
//                 [--gpus N | --devices a,b,..]  render every frame on several GPUs inside the one context
//                 [--copy-back]                  device buffer + blocking read-back instead of rt_register_output
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/uob_rt.h"
#include "screen.h"

using namespace std;
using namespace std::chrono;

int SCREEN_WIDTH = 1024, SCREEN_HEIGHT = 1024;                 // skeleton.cpp:32-33

float focal_length = 2200.0;                                   // :61
float camera_position[4] = {0.0f, 0.0f, -3.2f, 1.0f};          // :62 (glm::vec4)
float pitch = 0.0f, yaw = 0.0f;                                // :65-66
float light_position[4] = {0.0f, -0.5f, -0.7f, 1.0f};          // :67
bool quit = false;
bool lor = true;                                               // :74
vector<rt_triangle> triangles;                                 // :72
static rt_ctx* g_rt = nullptr;
static vector<string> g_keys;                                  // scripted key presses, one per frame
static size_t g_key_at = 0;

static void die(const char* op) {                              // checkError(), :499-507
  fprintf(stderr, "Error during operation '%s': %s\n", op, rt_last_error());
  exit(EXIT_FAILURE);
}

void opencl_initialise(const rt_config& cfg) {                 // :366-497
  const int n = (int)triangles.size();
  vector<float> v(12 * (size_t)n), nr(4 * (size_t)n), col(4 * (size_t)n);
  rt_scene_pack(triangles.data(), n, v.data(), nr.data(), col.data());     // :474-484
  if (rt_init(&cfg, v.data(), nr.data(), col.data(), n, &g_rt) != RT_OK) die("rt_init");
}

void offload_rendering(screen* screen) {                       // :146-182
  float rot[12];
  rt_rotation_matrix(yaw, pitch, rot);                         // :149-151
  if (rt_render(g_rt, rot, camera_position, light_position, focal_length, screen->buffer, nullptr) != RT_OK)
    die("rt_render");
}

bool update() {                                                // :282-361
  if (lor) {                                                   // light oscillation, :290-298
    float diff = -0.5f - light_position[0];
    if (diff > -0.001f) lor = false;
    light_position[0] += diff / 20.0f;
  } else {
    float diff = 0.5f - light_position[0];
    if (diff < 0.001f) lor = true;
    light_position[0] += diff / 20.0f;
  }
  while (g_key_at < g_keys.size()) {                           // while(SDL_PollEvent(&e)), :300-301
    const string& k = g_keys[g_key_at++];
    if (k == ".") return false;                                // no more events this frame
    if (k.compare(0, 2, "m:") == 0) {                          // SDL_MOUSEMOTION, :306-309
      int xrel = 0, yrel = 0;
      if (sscanf(k.c_str() + 2, "%d,%d", &xrel, &yrel) != 2) { fprintf(stderr, "bad mouse event '%s' (m:dx,dy)\n", k.c_str()); exit(2); }
      yaw += xrel * 0.0009f;
      pitch -= yrel * 0.0009f;
      continue;
    }
    if (k == "up") pitch -= 0.1;
    else if (k == "down") pitch += 0.1;
    else if (k == "left") yaw += 0.1;
    else if (k == "right") yaw -= 0.1;
    else if (k == "i") camera_position[2] += 0.1;
    else if (k == "o") camera_position[2] -= 0.1;
    else if (k == "k") camera_position[0] += 0.1;
    else if (k == "j") camera_position[0] -= 0.1;
    else if (k == "esc") { quit = true; return false; }
    return true;
  }
  return false;
}

int main(int argc, char* argv[]) {
  bool direct_out = true;
  int frames = 10;
  const char* out = "screenshot.bmp";
  const char* obj = nullptr;
  rt_config cfg;
  rt_config_default(&cfg);
  for (int i = 1; i < argc; ++i) {
    string a = argv[i];
    if (a == "--size" && i + 1 < argc) SCREEN_WIDTH = SCREEN_HEIGHT = atoi(argv[++i]);
    else if (a == "--frames" && i + 1 < argc) frames = atoi(argv[++i]);
    else if (a == "--aa" && i + 2 < argc) { cfg.aa_x = atoi(argv[++i]); cfg.aa_y = atoi(argv[++i]); }
    else if (a == "--shadows" && i + 1 < argc) cfg.shadow_samples = atoi(argv[++i]);
    else if (a == "--keys" && i + 1 < argc) { istringstream in(argv[++i]); string k; while (in >> k) g_keys.push_back(k); }
    else if (a == "--out" && i + 1 < argc) out = argv[++i];
    else if (a == "--obj" && i + 1 < argc) obj = argv[++i];
    else if (a == "--copy-back") direct_out = false;           // render into device memory + blocking copy, as the reference reads back
    else if (a == "--gpus" && i + 1 < argc) {
      cfg.num_devices = atoi(argv[++i]);
      if (cfg.num_devices < 1 || cfg.num_devices > RT_MAX_DEVICES) { fprintf(stderr, "--gpus must be in [1,%d]\n", RT_MAX_DEVICES); return 2; }
      for (int d = 0; d < cfg.num_devices; ++d) cfg.devices[d] = d;
    } else if (a == "--devices" && i + 1 < argc) {
      istringstream in(argv[++i]); string tok; cfg.num_devices = 0;
      while (getline(in, tok, ',') && cfg.num_devices < RT_MAX_DEVICES) cfg.devices[cfg.num_devices++] = atoi(tok.c_str());
    }
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  cfg.width = SCREEN_WIDTH; cfg.height = SCREEN_HEIGHT; cfg.band_rows = SCREEN_HEIGHT;
  focal_length = 1100.0f * (float)SCREEN_WIDTH / 1024.0f * (float)cfg.aa_x;   // 2200 at the reference's 1024 / 2x2

  screen* screen = InitializeSDL(SCREEN_WIDTH, SCREEN_HEIGHT, false);         // :98
  triangles.resize(64);
  const int n = rt_scene_cornell_box(triangles.data(), 64);                   // LoadTestModel, :101
  if (n < 0) die("rt_scene_cornell_box");
  triangles.resize(n);
  if (obj) {                                                                   // load_obj + insert, :102-103
    const int m = rt_scene_load_obj(obj, nullptr, 0);
    if (m < 0) die("rt_scene_load_obj");
    triangles.resize((size_t)n + m);
    if (rt_scene_load_obj(obj, triangles.data() + n, m) != m) die("rt_scene_load_obj");
  }
  printf("Triangles Length size %lu\n", triangles.size());                    // :104
  opencl_initialise(cfg);                                                      // :106
  // the device(s) write finished pixels straight into screen->buffer (no read-back after the kernel)
  if (direct_out &&
      rt_register_output(g_rt, screen->buffer, (size_t)SCREEN_WIDTH * SCREEN_HEIGHT * sizeof(uint32_t)) != RT_OK)
    die("rt_register_output");

  offload_rendering(screen);                                                   // initial scene, :109-110
  SDL_Renderframe(screen);
  for (int f = 0; f < frames && !quit; ++f) {                                  // :117-138
    update();
    auto start = high_resolution_clock::now();
    offload_rendering(screen);
    auto stop = high_resolution_clock::now();
    auto offload_duration = duration_cast<microseconds>(stop - start);
    cout << "\nOffloaded GPU Rendertime: " << offload_duration.count() << " micro seconds" << endl;
    cout << "Frame Rate: " << 1000000.0f / ((float)offload_duration.count()) << "FPS" << endl;
    SDL_Renderframe(screen);
  }
  SDL_SaveImage(screen, out);                                                  // :139
  printf("light_position.x %.9g yaw %.9g pitch %.9g camera %.9g %.9g\n", light_position[0], yaw, pitch,
         camera_position[0], camera_position[2]);
  rt_destroy(g_rt);
  KillSDL(screen);
  return 0;
}
