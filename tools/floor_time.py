"""Cost floor of the wave kernel: a view in which every task is trivially lit (camera close to the back wall)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from uob_raytracer_amd import abi, runtime as rt
box = rt.Scene.cornell_box()
for name, cam, sph, scene in (("default view", [0, 0, -3.2], abi.REFERENCE_SPHERES, box), ("default view, no spheres", [0, 0, -3.2], (), box),
                              ("empty room (10 wall triangles)", [0, 0, -3.2], (), rt.Scene(box.aos[:10].copy()))):
    cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, spheres=sph)
    tr = rt.RayTracer(cfg, scene)
    buf = torch.empty((4096, 4096), dtype=torch.int32, device="cuda")
    rot = rt.rotation_matrix(0, 0)
    ts = []
    for i in range(8):
        tr.render_device(rot, cam, [0, -0.5, -0.7], 1100.0 * 16, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ts.append(tr.last_kernel_ms())
    ex = tr.count_executed(rot, cam, [0, -0.5, -0.7], 1100.0 * 16)
    print("%-28s median %.3f ms  sampled %d  tasks_whole %d  nonblack %.3f" % (name, float(np.median(ts[2:])), ex["surface_points"], ex["tasks_resolved_whole"], float((buf != -16777216).float().mean())), flush=True)
    tr.close()
