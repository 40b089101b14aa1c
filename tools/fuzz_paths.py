"""Fuzz the exact culls: random scenes / lights / cameras / sampling, every device path must give the same bits.

    python tools/fuzz_paths.py [cases] [first_seed] [--wide]

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


def coop_frames_equal(kw, scene, rot, cam, light, focal, ref):
    """Mesh kernel: three frames of ONE context, every block of frames 2 and 3 rendered as four cooperative sub-block jobs
    (UOB_RT_MASK_DEBUG=8, read once in rt_init): the last frame must equal the single-frame result."""
    os.environ["UOB_RT_MASK_DEBUG"] = "8"
    try:
        tr = rt.RayTracer(abi.make_config(**kw), scene)
    finally:
        del os.environ["UOB_RT_MASK_DEBUG"]
    for _ in range(3):
        a, f = tr.render(rot, cam, light, focal, want_rgb=True)
    tr.close()
    return np.array_equal(a, ref[0]) and np.array_equal(f.view(np.uint32), ref[1].view(np.uint32))


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
    if len(scene) > 64 and kw["aa_x"] * kw["aa_y"] <= 64:
        from conftest import focal_for
        if not coop_frames_equal(kw, scene, rot, cam, light, focal_for(abi.make_config(**kw)), ref):
            print("MISMATCH seed %d: cooperative frames differ; n=%d kw=%s" % (seed, len(scene), kw), flush=True)
            return False
    return True


def _transform(scene, spheres, k, t):
    """x -> k x + t on triangles and spheres (float32 arithmetic; normals recomputed by the product's ComputeNormal)."""
    import ctypes as C
    aos = scene.aos.copy()
    aos[:, :3, :3] = (np.float32(k) * aos[:, :3, :3] + np.asarray(t, np.float32)).astype(np.float32)
    tri = aos.ctypes.data_as(C.POINTER(abi.RtTriangle))
    for i in range(aos.shape[0]):
        rt.lib().rt_triangle_compute_normal(C.byref(tri[i]))
    sph = tuple((tuple((np.float32(k) * np.asarray(c, np.float32) + np.asarray(t, np.float32)).tolist()),
                 float(np.float32(k) * np.float32(k) * np.float32(r2)), col) for c, r2, col in spheres)
    return rt.Scene(aos), sph


def wide_case(seed):
    """The accepted DOMAIN of rt_init (|coordinate| <= 2^16), not just unit scenes: whole-scene scales 2^-10 .. 2^14,
    translations up to 3e4, cameras up to 4e4 away (focal scaled to keep the scene in view), jitter spreads up to the
    scene size, lights on a triangle's plane / at a vertex / on a surface / far away, slivers of aspect 1e6.  The culled
    paths must still equal the unculled ones and the generic kernel, bit for bit."""
    rng = np.random.default_rng(900000 + seed)
    mesh = seed % 5 == 4
    if mesh:
        path = os.path.join(tempfile.mkdtemp(), "m.obj")
        meshgen.write_sphere_obj(path, int(rng.integers(8, 60)), int(rng.integers(6, 40)),
                                 radius=float(rng.uniform(0.05, 0.25)), bumps=float(rng.uniform(0, 0.3)))
        scene = rt.Scene.cornell_box() + rt.Scene.load_obj(path)
    else:
        scene = _random_scene(rng, int(rng.integers(2, 38)), box=bool(rng.integers(0, 2)))
    if rng.random() < 0.5:                       # slivers: two nearly parallel edges, aspect up to 1e6
        aos = scene.aos.copy()
        for i in rng.choice(len(scene), size=min(3, len(scene)), replace=False):
            e = aos[i, 1, :3] - aos[i, 0, :3]
            perp = np.cross(e, rng.uniform(-1, 1, 3)).astype(np.float32)
            perp /= max(float(np.linalg.norm(perp)), 1e-20)
            aos[i, 2, :3] = aos[i, 0, :3] + np.float32(rng.uniform(0.3, 1.5)) * e + np.float32(10.0 ** rng.uniform(-7, -3)) * perp
        scene = rt.Scene(aos)
    spheres = sphere_table(rng)
    # unit-scale view and light first
    cam = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(-3.4, -1.0)], np.float32)
    far = float(rng.choice([1.0, 1.0, 30.0, 1000.0]))                       # camera pulled back, focal scaled with it
    cam[2] *= np.float32(far)
    kind = int(rng.integers(0, 6))
    tri = scene.aos[int(rng.integers(0, len(scene)))]
    v0, e1, e2 = tri[0, :3], tri[1, :3] - tri[0, :3], tri[2, :3] - tri[0, :3]
    if kind == 0:   light = v0 + np.float32(rng.uniform(-0.5, 1.5)) * e1 + np.float32(rng.uniform(-0.5, 1.5)) * e2      # on a plane
    elif kind == 1: light = tri[int(rng.integers(0, 3)), :3].copy()                                                       # at a vertex
    elif kind == 2: light = v0 + np.float32(0.3) * e1 + np.float32(0.3) * e2                                               # on a surface
    elif kind == 3: light = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(1, 3)                                             # far away
    else:           light = rng.uniform(-0.95, 0.95, 3)
    light = np.asarray(light, np.float32)
    spread = float(rng.choice([0.0, 0.05, 0.3, 1.0, 2.5]))
    k = float(2.0 ** int(rng.choice([-10, -6, -3, 0, 0, 4, 10, 14])))
    # translations keep at least ~9 bits of the scene's extent representable (beyond that every triangle collapses)
    t = (rng.uniform(-1, 1, 3) * rng.choice([0.0, 0.0, 1.0e3, 3.0e4]) * min(k, 1.0)).astype(np.float32)
    if max(abs(float(x)) for x in (np.float32(k) * cam + t)) > 6.0e4 or max(abs(float(x)) for x in (np.float32(k) * light + t)) > 6.0e4:
        k = 1.0
    scene, spheres = _transform(scene, spheres, k, t)
    cam = (np.float32(k) * cam + t).astype(np.float32).tolist()
    light = (np.float32(k) * light + t).astype(np.float32).tolist()
    aa = [(1, 1), (2, 2), (4, 2), (2, 1), (3, 3), (4, 4)][int(rng.integers(0, 6))]
    S = int(rng.choice([1, 3, 8, 16, 33, 64, 64, 100]))
    W, H = int(rng.integers(40, 160)), int(rng.integers(30, 110))
    kw = dict(width=W, height=H, aa_x=aa[0], aa_y=aa[1], shadow_samples=S, light_spread=spread * k,
              max_bounces=int(rng.choice([0, 2, 10])), spheres=spheres)
    rot = rt.rotation_matrix(float(rng.uniform(-0.5, 0.5)) / far, float(rng.uniform(-0.4, 0.4)) / far)
    focal = 1100.0 * min(W, H) / 1024.0 * aa[0] * far
    info = "n=%d scale %g shift %s far %g light-kind %d spread %g" % (len(scene), k, t.tolist(), far, kind, spread * k)
    return scene, kw, rot, cam, light, focal, info


def one_case_wide(seed, stats=None):
    scene, kw, rot, cam, light, focal, info = wide_case(seed)
    flags = [0, abi.RT_FLAG_NO_TILE_BINS if len(scene) > 64 else abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL]
    ref = None
    for fl in flags:
        cfg = abi.make_config(flags=fl, **kw)
        tr = rt.RayTracer(cfg, scene)
        a, f = tr.render(rot, cam, light, focal, want_rgb=True)
        tr.close()
        if ref is None:
            ref = (a, f)
            if stats is not None:
                stats.append(float((a != 0xFF000000).mean()))
        elif not (np.array_equal(a, ref[0]) and np.array_equal(f.view(np.uint32), ref[1].view(np.uint32))):
            bad = np.argwhere(a != ref[0])
            print("MISMATCH wide seed %d flags %d: %d pixels, first %s; %s kw=%s"
                  % (seed, fl, len(bad), bad[:1].tolist(), info, {q: kw[q] for q in kw if q != "spheres"}), flush=True)
            return False
    if len(scene) > 64 and not coop_frames_equal(kw, scene, rot, cam, light, focal, ref):
        print("MISMATCH wide seed %d: cooperative frames differ; %s" % (seed, info), flush=True)
        return False
    return True


def main():
    wide = "--wide" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--wide"]
    cases = int(args[0]) if len(args) > 0 else 100
    first = int(args[1]) if len(args) > 1 else 0
    ok = 0
    stats = []
    for s in range(first, first + cases):
        ok += one_case_wide(s, stats) if wide else one_case(s)
        if (s - first) % 50 == 49:
            print("... %d cases, %d ok" % (s - first + 1, ok), flush=True)
    print("fuzz%s: %d / %d cases identical on all paths" % (" --wide" if wide else "", ok, cases))
    if stats:
        st = np.array(stats)
        print("non-black fraction of the frames: mean %.2f, frames above 10%%: %d of %d" % (st.mean(), int((st > 0.1).sum()), len(st)))
    return 0 if ok == cases else 1


if __name__ == "__main__":
    sys.exit(main())
