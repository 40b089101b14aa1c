#!/usr/bin/env python3
"""Diagnostic: headline frame with the light at several x (static, 25 frames each; median of the last 10) — what part of
the animated-light time (bench.py `animated_light`) is the position of the light rather than the frame-to-frame change."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from uob_raytracer_amd import abi, runtime as rt
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam = rt.rotation_matrix(0, 0), [0, 0, -3.2]
buf = torch.empty((4096, 4096), dtype=torch.int32, device="cuda")
for lx in [0.0, -0.1, -0.2, -0.3, -0.4, -0.5, 0.0]:
    ts = []
    for i in range(25):
        tr.render_device(rot, cam, [lx, -0.5, -0.7], 17600.0, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        ts.append(tr.last_kernel_ms())
    print("light x %+.2f: first frame at this position %.3f ms, then median %.3f ms" % (lx, ts[0], float(np.median(ts[15:]))), flush=True)
