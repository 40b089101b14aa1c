#!/usr/bin/env python3
"""Diagnostic: executed-work counters (rt_count_executed) of one 32-row band of the headline frame.
usage: count_band.py band_count band_index"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uob_raytracer_amd import abi, runtime as rt
bc, bi = int(sys.argv[1]), int(sys.argv[2])
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=bi, band_count=bc)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
for k, v in tr.count_executed(rot, cam, light, 17600.0).items():
    print("%-32s %d" % (k, v))
