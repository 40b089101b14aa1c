"""A/B kernel timing in one process: interleaved rounds of the headline frame for several builds of the library
(cdna_hip_programming.md rule 24).  usage: ab_time.py libA.so libB.so ...  (each loaded in a subprocess-free way is
impossible with one ctypes binding, so each library runs in its own child; rounds are interleaved by the parent)"""
import os, subprocess, sys, json
libs = sys.argv[1:]
child = r'''
import os, sys, json
import torch
sys.path.insert(0, os.getcwd())
from uob_raytracer_amd import abi, runtime as rt
bc = int(os.environ.get("AB_BANDS", "1"))       # time ONE rank's bands of a bc-rank job (strong scaling of the render)
cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc)
tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
buf = torch.empty((tr.rows, 4096), dtype=torch.int32, device="cuda")
for i in range(60): tr.render_device(rot, cam, light, 17600.0, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ts = []
for i in range(40):
    tr.render_device(rot, cam, light, 17600.0, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    ts.append(tr.last_kernel_ms())
ts.sort()
print(json.dumps({"median": ts[len(ts)//2], "min": ts[0]}))
'''
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, UOB_RT_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        if not out.stdout.strip():
            print(l, 'FAILED', out.stderr[-1500:]); sys.exit(1)
        res[l].append(json.loads(out.stdout.strip().splitlines()[-1]))
for l in libs:
    print(l, "median ms", ["%.3f" % r["median"] for r in res[l]], "min", ["%.3f" % r["min"] for r in res[l]])
