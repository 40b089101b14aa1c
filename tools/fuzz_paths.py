"""Fuzz the exact culls: random scenes / lights / cameras / sampling, every device path must give the same bits.

    python tools/fuzz_paths.py [cases] [first_seed]

Small scenes (wave kernel, n <= 64): cull on  ==  cull off (RT_FLAG_NO_CULL)  ==  generic kernel.
Meshes (n > 64): tile masks on  ==  tile masks off (RT_FLAG_NO_TILE_BINS)  ==  generic kernel.
The generic kernel is the one pinned to the CPU oracle (tests/test_gpu_parity.py); this tool widens the set of
configurations in which the other paths have been compared with it.  Prints the failing seed, if any.
"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from uob_raytracer_amd import abi, meshgen, runtime as rt
from test_gpu_cull import _random_scene, _render


def sphere_table(rng):
    k = int(rng.integers(0, 4))
    out = []
    for _ in range(k):
        mat = float(rng.choice([-1.0, 0.0, 1.0]))
        out.append((tuple(rng.uniform(-0.7, 0.7, 3).tolist()), float(rng.uniform(0.005, 0.12)),
                    (float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), mat)))
    return tuple(out)


def one_case(seed):
    rng = np.random.default_rng(seed)
    mesh = seed % 4 == 3
    if mesh:
        path = os.path.join(tempfile.mkdtemp(), "m.obj")
        meshgen.write_sphere_obj(path, int(rng.integers(8, 70)), int(rng.integers(6, 50)),
                                 radius=float(rng.uniform(0.05, 0.25)), bumps=float(rng.uniform(0, 0.3)))
        scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
        if rng.random() < 0.3:
            scene = scene + _random_scene(rng, int(rng.integers(5, 60)), box=False)
    else:
        scene = _random_scene(rng, int(rng.integers(2, 38)), box=bool(rng.integers(0, 2)))
    aa = [(1, 1), (2, 1), (2, 2), (4, 2), (4, 4), (8, 8), (3, 3), (3, 2), (5, 1), (7, 9), (6, 6)][int(rng.integers(0, 11))]
    S = int(rng.choice([1, 2, 3, 8, 9, 10, 16, 33, 64, 65, 100, 200]))
    W, H = int(rng.integers(40, 200)), int(rng.integers(30, 140))
    bc = int(rng.choice([1, 1, 2, 3]))
    kw = dict(width=W, height=H, aa_x=aa[0], aa_y=aa[1], shadow_samples=S,
              light_spread=float(rng.choice([0.0, 0.05, 0.2, 0.6])), max_bounces=int(rng.choice([0, 2, 10])),
              spheres=sphere_table(rng), band_rows=int(rng.choice([1, 7, 16, H])), band_index=int(rng.integers(0, bc)), band_count=bc)
    light = rng.uniform(-0.95, 0.95, 3).tolist() if rng.random() < 0.85 else rng.uniform(-4, 4, 3).tolist()
    cam = [float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-3.4, -1.0))]
    rot = rt.rotation_matrix(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.4, 0.4)))
    flags = [0, abi.RT_FLAG_NO_TILE_BINS if len(scene) > 64 else abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL]
    ref = None
    for fl in flags:
        a, f = _render(kw, fl, scene, rot, cam, light)
        if ref is None:
            ref = (a, f)
        elif not (np.array_equal(a, ref[0]) and np.array_equal(f.view(np.uint32), ref[1].view(np.uint32))):
            bad = np.argwhere(a != ref[0])
            print("MISMATCH seed %d flags %d: %d pixels, first %s; n=%d kw=%s" % (seed, fl, len(bad), bad[:1].tolist(), len(scene), kw), flush=True)
            return False
    return True


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ok = 0
    for s in range(first, first + cases):
        ok += one_case(s)
        if (s - first) % 50 == 49:
            print("... %d cases, %d ok" % (s - first + 1, ok), flush=True)
    print("fuzz: %d / %d cases identical on all paths" % (ok, cases))
    return 0 if ok == cases else 1


if __name__ == "__main__":
    sys.exit(main())
