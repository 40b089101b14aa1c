import os, sys
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from uob_raytracer_amd import abi, runtime as rt
for bc in [int(v) for v in (sys.argv[1:] or ['1', '2'])]:
    # band_count 2 with band_rows = H/2... emulate: render only this rank's half but compare per-row cost
    cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32, band_index=0, band_count=bc)
    tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
    buf = torch.empty((tr.rows, 4096), dtype=torch.int32, device="cuda")
    ts = []
    for i in range(45):      # (the device needs ~15 frames to reach its clocks)
        tr.render_device(rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7], 1100.0 * 16, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        ts.append(tr.last_kernel_ms())
    print("band_count", bc, "rows", tr.rows, "median ms %.3f" % float(np.median(ts[20:])), "-> per full frame %.3f" % (float(np.median(ts[20:])) * bc))
