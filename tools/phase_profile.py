#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the wave kernel on the headline frame (UOB_RT_PHASE_PROFILE=1)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UOB_RT_PHASE_PROFILE"] = "1"
from uob_raytracer_amd import abi, runtime as rt
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(os.environ.get("PP_SIZE", "4096")); AX, AY, SS = int(os.environ.get("PP_AAX", "4")), int(os.environ.get("PP_AAY", "2")), int(os.environ.get("PP_S", "64"))
cfg = abi.make_config(width=W, height=W, aa_x=AX, aa_y=AY, shadow_samples=SS, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc,
                      **({"spheres": ()} if os.environ.get("PP_NOSPH") else {}))
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
out = (C.c_uint64 * 8)()
rt._check(rt.lib().rt_count_executed(tr._h, rt._fp(rot), rt._fp(__import__("numpy").array(cam, "f4")),
                                      rt._fp(__import__("numpy").array(light, "f4")), C.c_float(1100.0 * W / 1024.0 * AX), out))
names = ["job set-up (primary bounds)", "primary+bounce", "light setup + level 1", "level 2", "xorshift streams", "level 3 sample tests",
         "shading + AA sum (dead code in this build)", "hand-out wait (+ staging once)"]
tot = sum(out)
for n, v in zip(names, out):
    print("%-26s %14d  %5.1f %%" % (n, v, 100.0 * v / max(tot, 1)))
