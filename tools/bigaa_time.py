import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from uob_raytracer_amd import abi, runtime as rt
for flags in (0, abi.RT_FLAG_GENERIC_KERNEL):
    cfg = abi.make_config(width=1024, height=1024, aa_x=9, aa_y=9, shadow_samples=64, flags=flags)
    tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
    buf = torch.empty((1024,1024),dtype=torch.int32,device="cuda"); ts=[]
    for i in range(8):
        tr.render_device(rt.rotation_matrix(0,0),[0,0,-3.2],[0,-0.5,-0.7],1100.0*9,buf.data_ptr(),None,torch.cuda.current_stream().cuda_stream); ts.append(tr.last_kernel_ms())
    print("1024^2, 9x9 AA, 64 samples, flags", flags, "median ms %.2f" % float(np.median(ts[3:])))
