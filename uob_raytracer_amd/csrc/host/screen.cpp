// screen.cpp — headless implementation of the reference's SDLauxiliary.h surface (see screen.h).
#include "screen.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

screen* InitializeSDL(int width, int height, bool /*fullscreen*/) {
  screen* s = new screen;
  memset(s, 0, sizeof *s);
  s->width = width;
  s->height = height;
  s->buffer = new uint32_t[(size_t)width * height];           // SDLauxiliary.h:105-106
  memset(s->buffer, 0, (size_t)width * height * sizeof(uint32_t));
  return s;
}

// SDLauxiliary.h:150-162: clamp(255*c, 0, 255) truncated, alpha 128
void PutPixelSDL(screen* s, int x, int y, float r, float g, float b) {
  if (x < 0 || x >= s->width || y < 0 || y >= s->height) {
    std::cout << "apa" << std::endl;
    return;
  }
  auto q = [](float c) { float v = 255 * c; v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v); return (uint32_t)v; };
  s->buffer[y * s->width + x] = (128u << 24) + (q(r) << 16) + (q(g) << 8) + q(b);
}

void SDL_Renderframe(screen* s) { s->frames_presented++; }    // SDLauxiliary.h:65-71 blits; headless: count

void KillSDL(screen* s) {                                     // SDLauxiliary.h:56-63
  delete[] s->buffer;
  delete s;
}

// SDLauxiliary.h:24-54: SDL_CreateRGBSurfaceFrom(buffer, ARGB masks) + SDL_SaveBMP
void SDL_SaveImage(screen* s, const char* filename) {
  FILE* f = fopen(filename, "wb");
  if (!f) {
    std::cout << "Failed to save image: cannot open " << filename << std::endl;
    exit(1);
  }
  const uint32_t w = (uint32_t)s->width, h = (uint32_t)s->height, image = w * h * 4, off = 14 + 108;
  uint8_t hdr[14 + 108];
  memset(hdr, 0, sizeof hdr);
  auto put32 = [&](int at, uint32_t v) { memcpy(hdr + at, &v, 4); };
  auto put16 = [&](int at, uint16_t v) { memcpy(hdr + at, &v, 2); };
  hdr[0] = 'B'; hdr[1] = 'M';
  put32(2, off + image); put32(10, off);
  put32(14, 108);                       // BITMAPV4HEADER
  put32(18, w); put32(22, h);           // positive height: bottom-up
  put16(26, 1); put16(28, 32);
  put32(30, 3);                         // BI_BITFIELDS
  put32(34, image);
  put32(54, 0x00FF0000u); put32(58, 0x0000FF00u); put32(62, 0x000000FFu); put32(66, 0xFF000000u);   // R G B A masks
  put32(70, 0x57696E20u);               // LCS_WINDOWS_COLOR_SPACE, as SDL writes
  fwrite(hdr, 1, sizeof hdr, f);
  for (uint32_t y = 0; y < h; ++y) fwrite(s->buffer + (size_t)(h - 1 - y) * w, 4, w, f);
  fclose(f);
}
