"""Frame time against the number of 64-ray tasks per job (UOB_RT_JOB_TASKS), headline and configs[1] frames."""
import os, subprocess, sys
code = r'''
import sys; sys.path.insert(0, %r)
import torch, numpy as np
from uob_raytracer_amd import abi, runtime as rt
for name, kw, fs in (("headline", dict(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64), 16.0),
                     ("cfg2", dict(width=1024, height=1024, aa_x=2, aa_y=2, shadow_samples=16, spheres=()), 2.0)):
    cfg = abi.make_config(**kw)
    tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
    buf = torch.empty((kw["height"], kw["width"]), dtype=torch.int32, device="cuda")
    ts = []
    for i in range(14):
        tr.render_device(rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7], 1100.0 * kw["width"] / 1024 * kw["aa_x"], buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); ts.append(tr.last_kernel_ms())
    print("%%s median ms %%.3f checksum %%d;" %% (name, float(np.median(ts[3:])), int(buf.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())), end=" ")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for jt in (8, 4, 2, 1):
    env = dict(os.environ, UOB_RT_JOB_TASKS=str(jt))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("job_tasks", jt, out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else "", flush=True)
