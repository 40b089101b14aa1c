"""A/B kernel timing of builds of the library on the small BASELINE.json configurations (configs[1] 1024^2 / 16 samples /
no spheres, configs[2] 1920x1080 mirror wall / 10 samples / depth 5, the reference as shipped 1024^2 / 10 samples): each build
in its own child process, rounds interleaved.   usage: ab_small.py libA.so libB.so ..."""
import json, os, subprocess, sys
libs = sys.argv[1:]
child = r'''
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from uob_raytracer_amd import abi, runtime as rt
box = rt.Scene.cornell_box()
mirror = box.with_color([8, 9], (1.0, 1.0, 1.0, 0.0))
cases = {"cfg2": (dict(width=1024, height=1024, shadow_samples=16, spheres=()), box), "cfg3": (dict(width=1920, height=1080, max_bounces=5), mirror),
         "reference": (dict(width=1024, height=1024), box)}
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
out = {}
for name, (kw, sc) in cases.items():
    cfg = abi.make_config(**kw)
    tr = rt.RayTracer(cfg, sc)
    buf = torch.empty((kw["height"], kw["width"]), dtype=torch.int32, device="cuda")
    f = 1100.0 * min(kw["width"], kw["height"]) / 1024.0 * 2
    for i in range(300): tr.render_device(rot, cam, light, f, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for i in range(100):
        tr.render_device(rot, cam, light, f, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        ts.append(tr.last_kernel_ms())
    ts.sort(); out[name] = ts[50]
    out[name + "_sum"] = int(buf.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    tr.close()
print(json.dumps(out))
'''
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        o = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, UOB_RT_LIB=os.path.abspath(l)), capture_output=True, text=True)
        if not o.stdout.strip():
            print(l, "FAILED", o.stderr[-1500:]); sys.exit(1)
        res[l].append(json.loads(o.stdout.strip().splitlines()[-1]))
for l in libs:
    for r in res[l]:
        print(l, " ".join("%s %s" % (k, ("%.4f" % v) if isinstance(v, float) else v) for k, v in r.items()))
