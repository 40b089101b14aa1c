"""One rank's bands of an N-rank job in three scenes (default, no spheres, empty room): where a short frame's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from uob_raytracer_amd import abi, runtime as rt
box = rt.Scene.cornell_box()
for bc in [int(v) for v in (sys.argv[1:] or ["1", "8"])]:
    for name, sph, scene in (("default", abi.REFERENCE_SPHERES, box), ("no spheres", (), box), ("empty room", (), rt.Scene(box.aos[:10].copy()))):
        cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, spheres=sph, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc)
        tr = rt.RayTracer(cfg, scene)
        buf = torch.empty((tr.rows, 4096), dtype=torch.int32, device="cuda")
        rot = rt.rotation_matrix(0, 0)
        ts = []
        for i in range(12):
            tr.render_device(rot, [0, 0, -3.2], [0, -0.5, -0.7], 1100.0 * 16, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ts.append(tr.last_kernel_ms())
        print("bands %d  %-12s median %.3f ms  (x%d = %.3f)" % (bc, name, float(np.median(ts[3:])), bc, bc * float(np.median(ts[3:]))), flush=True)
        tr.close()
