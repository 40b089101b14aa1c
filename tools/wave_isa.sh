#!/bin/bash
# Diagnostic: resource usage and ISA of the shipped wave-kernel instantiation.  usage: tools/wave_isa.sh out.s [extra -D flags]
set -e
cd "$(dirname "$0")/../uob_raytracer_amd/csrc"
out=$1; shift
K='_ZN5uobrt12rt_draw_waveILb1ELb0ELb0ELi32ELb0EEEvNS_11FrameParamsE'
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 --cuda-device-only "$@" -S rt_kernel_wave.hip -o "$out.all" 2>/dev/null
awk "/^$K:/,/s_endpgm/" "$out.all" > "$out"
awk "/^$K:/,/\.end_amdhsa_kernel/" "$out.all" | grep -E "; (NumVgprs|ScratchSize|SGPRBlocks|Occupancy)|sgpr_spill|vgpr_spill" | tr '\n' ' '; echo
grep -A40 "\.name: *$K" "$out.all" | grep -E "sgpr_spill_count|vgpr_spill_count|vgpr_count" | tr '\n' ' '; echo
echo "lines $(wc -l < "$out")  valu $(grep -cE '^\s+v_' "$out")  readlane $(grep -c v_readlane "$out")  writelane $(grep -c v_writelane "$out")"
