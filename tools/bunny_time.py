"""The reference's main() scene shape: Cornell Box + a ~200-triangle OBJ (it loads bunny_200.obj, skeleton.cpp:102),
reference constants (1024^2, 2x2 AA, 10 shadow rays, 2 spheres).  Times the tiled mesh kernel and the others."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uob_raytracer_amd import abi, meshgen, runtime as rt
path = os.path.join(tempfile.mkdtemp(), "m.obj")
nf = meshgen.write_sphere_obj(path, 10, 11)
scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
ref = None
for name, flags in (("tiled mesh kernel", 0), ("every tile", abi.RT_FLAG_NO_TILE_BINS), ("generic kernel", abi.RT_FLAG_GENERIC_KERNEL)):
    tr = rt.RayTracer(abi.make_config(flags=flags, spheres=() if os.environ.get("NOSPH") else abi.REFERENCE_SPHERES), scene)
    ts = []
    for i in range(8):
        a = tr.render(rot, cam, light, 2200.0); ts.append(tr.last_kernel_ms())
    if ref is None: ref = a
    print("%-18s triangles %d  median %.3f ms  identical %s" % (name, len(scene), float(np.median(ts[2:])), bool(np.array_equal(a, ref))), flush=True)
box = rt.RayTracer(abi.make_config(), rt.Scene.cornell_box())
ts = []
for i in range(8):
    box.render(rot, cam, light, 2200.0); ts.append(box.last_kernel_ms())
print("box only (wave kernel)  median %.3f ms" % float(np.median(ts[2:])))
