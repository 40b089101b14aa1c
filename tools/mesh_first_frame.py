"""configs[4] (box + 100 026-triangle mesh, 2048^2, 1 spp, 1 shadow ray): kernel time of a context's FIRST frame (ordered by
rt_mesh_estimate's guess) and of the frames after it (ordered by measured block costs), five fresh contexts each.
UOB_RT_MASK_DEBUG=32: guess from the primary side alone; 16: no guess (row order); NOSPH=0: with the reference's two spheres."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uob_raytracer_amd import abi, meshgen, runtime as rt
path = os.path.join(tempfile.mkdtemp(), "m.obj")
meshgen.write_sphere_obj(path, 250, 201)
scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
W = 2048
sph = () if os.environ.get("NOSPH", "1") == "1" else abi.REFERENCE_SPHERES
if os.environ.get("SPHMAT"):      # the reference's spheres with another material: 1 diffuse, 0 mirror, -1 glass (which cost is the bounce rays'?)
    sph = tuple((c, r2, (0.5, 0.5, 0.5, float(os.environ["SPHMAT"]))) for c, r2, _ in abi.REFERENCE_SPHERES)
cfg = abi.make_config(width=W, height=W, aa_x=1, aa_y=1, shadow_samples=1, spheres=sph, max_bounces=int(os.environ.get("BOUNCES", "10")))
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
warm = rt.RayTracer(cfg, scene)
for _ in range(12):
    warm.render(rot, cam, light, 1100.0 * W / 1024)          # clocks
firsts, laters = [], []
for rep in range(5):
    tr = rt.RayTracer(cfg, scene)
    ts = []
    for i in range(6):
        tr.render(rot, cam, light, 1100.0 * W / 1024)
        ts.append(tr.last_kernel_ms())
    firsts.append(ts[0]); laters.append(float(np.median(ts[2:])))
    tr.close()
print("triangles %d, MASK_DEBUG=%s NOSPH=%s SPHMAT=%s BOUNCES=%s:" % (len(scene), os.environ.get("UOB_RT_MASK_DEBUG", "0"), os.environ.get("NOSPH", "1"), os.environ.get("SPHMAT"), os.environ.get("BOUNCES")), end="")
print(" %s first frame of a context %.2f ms (min %.2f, max %.2f), later frames %.2f ms" % (
    "", float(np.median(firsts)), min(firsts), max(firsts), float(np.median(laters))))
