// rt_kernel_wave.hip — placeholder until the wave-per-hit-point kernel lands.
#include "rt_device.h"
namespace uobrt {
bool wave_kernel_supports(const FrameParams&) { return false; }
void launch_wave(const FrameParams&, hipStream_t) {}
}  // namespace uobrt
