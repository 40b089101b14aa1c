// screen.h — the reference's presentation surface (Source/SDLauxiliary.h), headless.
//
// Same struct name, field names and function names as the reference so that its main loop compiles against
// this header unchanged: screen{height,width,buffer} (SDLauxiliary.h:9-16; the SDL_Window/Renderer/Texture
// members are opaque placeholders here), InitializeSDL (:73), PutPixelSDL (:150), SDL_Renderframe (:65),
// SDL_SaveImage (:24), KillSDL (:56).  SDL2 is not available in the build image, so "presenting" a frame
// only counts it; SDL_SaveImage writes the same kind of file SDL_SaveBMP writes for an ARGB8888 surface:
// a bottom-up 32-bit BI_BITFIELDS BMP with a BITMAPV4HEADER (byte-level parity with SDL2 is unpinned —
// no SDL2 here to compare with; tests read the file back and compare pixels).
#pragma once
#include <cstdint>

struct screen {
  void* window;      // SDL_Window*   in the reference
  void* renderer;    // SDL_Renderer*
  void* texture;     // SDL_Texture*
  int height;
  int width;
  uint32_t* buffer;  // width*height ARGB8888, row-major (SDLauxiliary.h:105)
  long frames_presented;
};

screen* InitializeSDL(int width, int height, bool fullscreen = false);
void PutPixelSDL(screen* s, int x, int y, float r, float g, float b);   // glm::vec3 colour in the reference
void SDL_Renderframe(screen* s);
void KillSDL(screen* s);
void SDL_SaveImage(screen* s, const char* filename);
