"""Diagnostic: the shadow-ray tile masks of configs[4] (box + 100 000-triangle mesh): how many tiles each occupied world cell names."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uob_raytracer_amd import abi, meshgen, runtime as rt
path = os.path.join(tempfile.mkdtemp(), "m.obj")
meshgen.write_cubesphere_obj(path, 91) if os.environ.get("CUBE") else meshgen.write_sphere_obj(path, 250, 201)
scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
W = 2048
cfg = abi.make_config(width=W, height=W, aa_x=1, aa_y=1, shadow_samples=1, spheres=())
tr = rt.RayTracer(cfg, scene)
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
tr.render(rot, cam, light, 1100.0 * W / 1024)
masks = tr.world_masks()
bits = np.unpackbits(masks.view(np.uint8), axis=-1, bitorder="little")          # [G, G, G, tiles]
m = bits.sum(-1)
occ = m[m > 0]
per_tile = bits.reshape(-1, bits.shape[-1])[m.ravel() > 0].sum(0)
print("tiles named by more than 90 %% of the occupied cells: %d; by more than half: %d" % ((per_tile > 0.9 * occ.size).sum(), (per_tile > 0.5 * occ.size).sum()))
print("  the most named tiles:", np.argsort(per_tile)[::-1][:48].tolist())
print("cells", m.size, "with a mask", occ.size, "tiles per such cell: mean %.1f median %d max %d" % (occ.mean(), np.median(occ), occ.max()))
print("histogram", np.histogram(occ, bins=[1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049])[0])
G = m.shape[0]
for z in (G - 1, G // 2):
    print("slice z =", z)
    print(m[z, ::2, ::2])
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "world_mask_counts.npy"), m)
