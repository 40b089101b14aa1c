#!/usr/bin/env python3
"""First contact: the reference's own kernel (oracle/_ref/*.co) on the box's OpenCL device vs the CPU oracle and the HIP product.
Writes gpurun_out/ref_gpu_try.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref, ref_gpu            # noqa: E402
from uob_raytracer_amd import abi, runtime as rt   # noqa: E402

out = {"probe": ref_gpu.probe()}
print(out["probe"], flush=True)
scene = rt.Scene.cornell_box()
v, n, c = scene.packed()
rot = rt.rotation_matrix(0.0, 0.0)
cam, light = [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]


def diffstats(a, b):
    a = a.astype(np.uint32).ravel(); b = b.astype(np.uint32).ravel()
    ch = lambda x, s: ((x >> s) & 255).astype(np.int32)
    d = np.maximum.reduce([np.abs(ch(a, s) - ch(b, s)) for s in (0, 8, 16)])
    return {"pixels": int(a.size), "identical": int((d == 0).sum()), "within_1lsb": int((d <= 1).sum()),
            "beyond_1lsb": int((d > 1).sum()), "max_channel_diff": int(d.max())}


for variant, W, H, kw in (("default256", 256, 256, {}), ("default", 1024, 1024, {}), ("default_plain", 1024, 1024, {}),
                          ("cfg1", 256, 256, dict(aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=())),
                          ("s64_512", 512, 512, dict(shadow_samples=64))):
    if not ref_gpu.have(variant):
        continue
    cfg = abi.make_config(width=W, height=H, **kw)
    focal = 1100.0 * min(W, H) / 1024.0 * cfg.aa_x
    try:
        argb, info = ref_gpu.run(variant, W, H, v, n, c, rot, cam, light, focal, reps=5)
    except Exception as e:      # noqa: BLE001
        out[variant] = {"error": str(e)}
        print(variant, out[variant], flush=True)
        continue
    o_argb, _ = pyref.Oracle().render(cfg, v, n, c, rot, cam, light, focal, nthreads=16)
    tr = rt.RayTracer(cfg, scene)
    h_argb = tr.render(rot, cam, light, focal)
    hms = tr.last_kernel_ms()
    tr.close()
    out[variant] = {"info": info, "ref_gpu_vs_oracle": diffstats(argb, o_argb), "hip_vs_oracle": diffstats(h_argb, o_argb),
                    "hip_kernel_ms": hms, "fnv_ref_gpu": "%016x" % pyref.fnv1a64_words(argb) if W * H <= 1 << 20 else None}
    print(variant, json.dumps(out[variant]), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "ref_gpu_%s.npy" % variant), argb.reshape(H, W))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "ref_gpu_try.json"), "w") as f:
    json.dump(out, f, indent=1)
