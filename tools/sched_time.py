"""How much the wave kernel's frame time depends on what its scheduler knows: for each build of the library (in its own
child process, rounds interleaved) the headline frame
  static     median of 40 identical frames after 60 warm-up frames (the scheduler's best case)
  first      a fresh context's first frame (no list of expensive jobs), mean of 5 contexts on a warm GPU
  animated   light x moved as the reference's update() does (skeleton.cpp:290-298), mean / max of 20 frames
  jump       the frame after the light jumps by 0.2 (a stale list), mean of 5
  rank8      one rank's bands of an 8-rank split (512 rows), median of 40 static frames
usage: sched_time.py libA.so [libB.so ...]   (environment knobs such as UOB_RT_STEAL=0 reach the children)"""
import json
import os
import subprocess
import sys

libs = sys.argv[1:]
child = r'''
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from uob_raytracer_amd import abi, runtime as rt
W = 4096
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
st = torch.cuda.current_stream().cuda_stream
def ctx(bc=1):
    cfg = abi.make_config(width=W, height=W, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc)
    tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
    return tr, torch.empty((tr.rows, W), dtype=torch.int32, device="cuda")
def frame(tr, buf, li):
    tr.render_device(rot, cam, li, 17600.0, buf.data_ptr(), None, st)
    return tr.last_kernel_ms()
out = {}
tr, buf = ctx()
for i in range(60): frame(tr, buf, light)
ts = sorted(frame(tr, buf, light) for i in range(40))
out["static"] = ts[20]
firsts = []
for i in range(5):
    t2, b2 = ctx()
    firsts.append(frame(t2, b2, light))
    for j in range(3): frame(tr, buf, light)        # keep the clocks up between the contexts
    t2.close()
out["first"] = float(np.mean(firsts))
lx, lor, lights = np.float32(0.0), True, []
for _ in range(24):
    diff = (np.float32(-0.5) if lor else np.float32(0.5)) - lx
    if lor and diff > np.float32(-0.001): lor = False
    elif not lor and diff < np.float32(0.001): lor = True
    lx = np.float32(lx + diff / np.float32(20.0))
    lights.append([float(lx), -0.5, -0.7])
an = [frame(tr, buf, li) for li in lights][4:]
out["animated"] = float(np.mean(an)); out["animated_max"] = float(np.max(an))
jm = []
for i in range(5):
    for j in range(4): frame(tr, buf, [0.2 * (i % 2), -0.5, -0.7])
    jm.append(frame(tr, buf, [0.2 * ((i + 1) % 2), -0.5, -0.7]))
out["jump"] = float(np.mean(jm))
tr.close()
tr, buf = ctx(8)
for i in range(100): frame(tr, buf, light)
ts = sorted(frame(tr, buf, light) for i in range(40))
out["rank8"] = ts[20]
print(json.dumps(out))
'''
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, UOB_RT_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        if not out.stdout.strip():
            print(l, "FAILED", out.stderr[-1500:])
            sys.exit(1)
        res[l].append(json.loads(out.stdout.strip().splitlines()[-1]))
for l in libs:
    for r in res[l]:
        print(l, " ".join("%s %.3f" % (k, v) for k, v in r.items()))
