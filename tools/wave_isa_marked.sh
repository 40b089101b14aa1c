#!/bin/bash
# Diagnostic: ISA of the shipped wave kernel with "; RTMARK n" comments where the PROF build takes its phase stamps.
# usage: tools/wave_isa_marked.sh out.s
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d); mkdir -p $tmp/a/b $tmp/include
cp $root/include/uob_rt.h $tmp/include/; cp $root/uob_raytracer_amd/csrc/*.h $root/uob_raytracer_amd/csrc/rt_kernel_wave.hip $tmp/a/b/
cd $tmp/a/b
python3 - <<'PY'
s=open('rt_kernel_wave.hip').read()
s=s.replace('#define RT_STAMP(slot)                                                              \\\n  if (PROF) {','#define RT_STAMP(slot)                                                              \\\n  asm volatile("; RTMARK " #slot);                                       \\\n  if (PROF) {')
open('rt_kernel_wave.hip','w').write(s)
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 --cuda-device-only -S rt_kernel_wave.hip -o mk.s 2>/dev/null
awk '/^_ZN5uobrt12rt_draw_waveILb1ELb0ELb0ELi32ELb0EEEvNS_11FrameParamsE:/,/s_endpgm/' mk.s > "$1"
grep -n RTMARK "$1"
rm -rf $tmp
