#!/usr/bin/env python3
"""Diagnostic: how the persistent waves of the wave kernel spend the headline frame (UOB_RT_TIMELINE=1, rt_debug_wave_timeline).
usage: wave_timeline.py [band_count [band_index]]   -- band_count > 1: one rank's 32-row bands of a band_count-rank job"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UOB_RT_TIMELINE"] = "1"
import torch
from uob_raytracer_amd import abi, runtime as rt
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=bi, band_count=bc, flags=int(os.environ.get("TL_FLAGS", "0")))
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
buf = torch.empty((tr.rows, 4096), dtype=torch.int32, device="cuda")
for i in range(30):
    tr.render_device(rot, cam, light, 17600.0, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = tr.wave_timeline()
    if i < 3 or i % 5 == 4: print("frame %d: kernel %.3f ms  " % (i, tr.last_kernel_ms()), t, flush=True)
