import os, sys
sys.path.insert(0, os.getcwd())
from uob_raytracer_amd import abi, runtime as rt
import numpy as np
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
for i in range(3):
    tr.render(rot, cam, light, 17600.0)
    print(os.environ.get("UOB_RT_DEBUG_STOP"), "kernel ms", tr.last_kernel_ms())
