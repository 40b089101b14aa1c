"""Time BASELINE.json configs[4]: Cornell Box + ~100k-triangle OBJ mesh, 2048x2048, 1 spp, 1 shadow ray."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uob_raytracer_amd import abi, meshgen, runtime as rt
n_lon, n_lat, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
path = os.path.join(tempfile.mkdtemp(), "m.obj")
nf = meshgen.write_cubesphere_obj(path, n_lon) if os.environ.get("CUBE") else meshgen.write_sphere_obj(path, n_lon, n_lat)     # CUBE=1: a cube-sphere of 12 n_lon^2 triangles (no polar slivers)
scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
cfg = abi.make_config(width=W, height=W, aa_x=1, aa_y=1, shadow_samples=1, spheres=() if os.environ.get("NOSPH", "1") == "1" else abi.REFERENCE_SPHERES)   # configs[4] as bench.py --workload cfg5
tr = rt.RayTracer(cfg, scene)
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
for i in range(2):
    t = time.time(); a = tr.render(rot, cam, light, 1100.0 * W / 1024); dt = time.time() - t
    print("triangles", len(scene), "size", W, "kernel ms %.1f" % tr.last_kernel_ms(), "wall %.3f s" % dt, "nonblack", float((a != 0xFF000000).mean()), flush=True)
if len(sys.argv) > 4:
    for k, v in tr.count_executed(rot, cam, light, 1100.0 * W / 1024).items():
        print("  %-28s %d" % (k, v))
if os.environ.get("COSTS"):
    import numpy as np
    tr.render(rot, cam, light, 1100.0 * W / 1024)
    c = tr.block_costs().astype(np.float64)
    print("blocks", c.shape, "sum %.3g ticks" % c.sum(), "max %.3g" % c.max(), "mean %.3g" % c.mean(), "blocks above max/4:", int((c > c.max() / 4).sum()))
    ys, xs = np.unravel_index(np.argsort(c.ravel())[::-1][:12], c.shape)
    for y, x in zip(ys, xs):
        print("  block row %d col %d: %.3g ticks" % (y, x, c[y, x]))
    hist, edges = np.histogram(np.log10(c.ravel() + 1), bins=12)
    print("log10(ticks) histogram", [(round(float(e), 1), int(h)) for e, h in zip(edges, hist)])
