"""Which path is wrong?  One --wide fuzz case rendered by every device path and by the CPU oracle.
    python tools/fuzz_debug.py <seed> [lib.so ...]      (UOB_RT_LIB is set per child for each library given)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
if len(sys.argv) > 2:
    for lib in sys.argv[2:]:
        print("==== " + lib, flush=True)
        subprocess.run([sys.executable, __file__, sys.argv[1]], env=dict(os.environ, UOB_RT_LIB=os.path.abspath(lib)))
    sys.exit(0)
import numpy as np
from uob_raytracer_amd import abi, runtime as rt
from oracle import pyref
from fuzz_paths import wide_case
seed = int(sys.argv[1])
scene, kw, rot, cam, light, focal, info = wide_case(seed)
print(info, {k: v for k, v in kw.items() if k != "spheres"}, "spheres", kw["spheres"], "cam", cam, "light", light, "focal", focal)
v, n, c = scene.packed()
o_argb, o_rgb = pyref.Oracle().render(abi.make_config(**kw), v, n, c, rot, cam, light, focal, nthreads=16)
o_argb = o_argb.reshape(kw["height"], kw["width"])
names = {0: "default", abi.RT_FLAG_NO_TILE_BINS: "no tile masks", abi.RT_FLAG_NO_CULL: "no cull", abi.RT_FLAG_GENERIC_KERNEL: "generic"}
for fl in (0, abi.RT_FLAG_NO_TILE_BINS if len(scene) > 64 else abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL):
    tr = rt.RayTracer(abi.make_config(flags=fl, **kw), scene)
    a = tr.render(rot, cam, light, focal)
    tr.close()
    bad = np.argwhere(a != o_argb)
    print("%-14s vs oracle: %d pixels differ" % (names[fl], len(bad)), bad[:6].tolist(),
          [(hex(a[tuple(b)]), hex(o_argb[tuple(b)])) for b in bad[:3]])
