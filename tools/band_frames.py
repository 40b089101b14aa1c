"""K frames of ONE rank's bands of an N-rank job (for rocprofv3 passes): python3 tools/band_frames.py N [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uob_raytracer_amd import abi, runtime as rt
bc = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
buf = torch.empty((tr.rows, 4096), dtype=torch.int32, device="cuda")
for _ in range(K):
    tr.render_device(rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7], 17600.0, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("bands", bc, "rows", tr.rows, "last kernel ms %.3f" % tr.last_kernel_ms())
