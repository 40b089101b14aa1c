"""Wide-domain fuzz cases (tools/fuzz_paths.py wide_case: scales 2^-10 .. 2^14, translations, far cameras, slivers, lights on
planes / vertices) rendered by the default device path and compared with the CPU ORACLE, bit for bit — what a change to
arithmetic shared by every device path (rt_math.h) needs, since the paths can then only agree with each other.
usage: fuzz_oracle.py [cases] [first seed]        (the oracle is test infrastructure: this tool is one of its callers' kin, tools/ only)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from uob_raytracer_amd import abi, runtime as rt
from oracle import pyref
from fuzz_paths import wide_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
orc = pyref.Oracle()
bad_cases, pixels = 0, 0
for seed in range(first, first + cases):
    scene, kw, rot, cam, light, focal, info = wide_case(seed)
    if kw["width"] * kw["height"] * kw["aa_x"] * kw["aa_y"] * (1 + kw["shadow_samples"]) * len(scene) > 6e9:
        continue                                        # keep the oracle's share of the run to seconds per case
    v, n, c = scene.packed()
    cfg = abi.make_config(**kw)
    o_argb, o_rgb = orc.render(cfg, v, n, c, rot, cam, light, focal, nthreads=16)
    tr = rt.RayTracer(cfg, scene)
    a, f = tr.render(rot, cam, light, focal, want_rgb=True)
    tr.close()
    pixels += a.size
    if not (np.array_equal(a.ravel(), o_argb) and np.array_equal(f[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))):
        bad_cases += 1
        print("MISMATCH with the oracle: wide seed %d: %d pixels; %s" % (seed, int((a.ravel() != o_argb).sum()), info), flush=True)
    if (seed - first) % 25 == 24:
        print("... %d cases, %d mismatching" % (seed - first + 1, bad_cases), flush=True)
print("fuzz vs oracle: seeds %d..%d, %d pixels, %d cases differ from the oracle" % (first, first + cases - 1, pixels, bad_cases))
sys.exit(1 if bad_cases else 0)
