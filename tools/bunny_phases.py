"""Diagnostic: s_memtime shares per phase of the mesh kernel on the reference's main() scene shape (box + ~200-triangle OBJ,
reference constants), and its executed-work counters.   UOB_RT_PHASE_PROFILE is set for the first context only."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uob_raytracer_amd import abi, meshgen, runtime as rt
path = os.path.join(tempfile.mkdtemp(), "m.obj")
meshgen.write_sphere_obj(path, 10, 11)
scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
names = ["0 job + task set-up", "1 primary tiles", "2 primary load / hit", "3 shadow set-up", "4 tile load + barrier", "5 level 1", "6 level 2", "7 level 3 + rest"]
os.environ["UOB_RT_PHASE_PROFILE"] = "1"
tr = rt.RayTracer(abi.make_config(), scene)
v = list(tr.count_executed(rot, cam, light, 2200.0).values())
tot = sum(v)
for n, x in zip(names, v):
    print("%-24s %5.1f %%" % (n, 100.0 * x / tot))
tr.close()
del os.environ["UOB_RT_PHASE_PROFILE"]
tr = rt.RayTracer(abi.make_config(), scene)
print(tr.count_executed(rot, cam, light, 2200.0))
