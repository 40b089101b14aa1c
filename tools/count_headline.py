import sys; sys.path.insert(0,'/root/repo')
import torch
from uob_raytracer_amd import abi, runtime as rt
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
print(tr.count_executed(rt.rotation_matrix(0,0), [0,0,-3.2], [0,-0.5,-0.7], 1100.0*4*4))
