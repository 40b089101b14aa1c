"""Directed fuzz: lights placed ON the decision thresholds of the exact culls.

    python tools/fuzz_thresholds.py [cases] [first_seed]

tools/fuzz_paths.py draws scenes at random; a random draw almost never lands where a certificate of the interval cull
(rt_wave_common.h point_bound / spheres_point / sphere_bundle_maybe) changes its mind.  This tool AIMS there.  Per case:
a random small scene and view, one target pixel and its surface point X, one shadow-casting triangle T (or a sphere), one
of the certificates' decision functions g, evaluated by a float64 model of the kernel's formula:

  A1+ / A1- / A2+ / A2-   det(A1) (det(A2)) centre against hh |p|_1 (hh |q|_1): the sign-consistency clauses (kernels.cl:268-272)
  W                       u + v <= 1 as one linear function: W0 - hh |w|_1 against slackW                    (kernels.cl:272)
  R                       |t d|^2 against radius_sq: |det(A0)| dminlen against hiD dk                          (kernels.cl:266)
  S                       a sphere's miss certificate: crn^2 - hh |cr x L|_1 against R (|dir| + jm) 1.002 crn (kernels.cl:285)
  S40 / S73               the switch |L| / R = 40 that turns the sphere certificates off (and 73, where the margin argument ends)

The light is moved along one axis until g = 0 (bisection in the model), then the frame is rendered with the light at that
coordinate + k ulp for k in -8..8 and +-16, 32, 64 (the model and the kernel's FP32 evaluation differ by a few ulp), by
three device paths — interval cull on, cull off (RT_FLAG_NO_CULL), generic kernel: identical bits required.  On both
sides of a flip the frame must not change with the path; a certificate that claims more than it proves shows up here
first.  `case(seed)` is importable (tests/test_gpu_thresholds.py runs a slice of it).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from uob_raytracer_amd import abi, runtime as rt

KINDS = ("A1+", "A1-", "A2+", "A2-", "W", "R", "S", "S40", "S73")
OFFSETS = sorted(set(range(-8, 9)) | {-64, -32, -16, 16, 32, 64})


def cof(a, b):          # the reference's 2x2 cofactors (rt_math.h cof): det(m0, m1, m2) = m0.x c.x - m0.y c.y + m0.z c.z
    return np.array([a[1] * b[2] - a[2] * b[1], a[0] * b[2] - a[2] * b[0], a[0] * b[1] - a[1] * b[0]])


def detc(m, c):
    return m[0] * c[0] - m[1] * c[1] + m[2] * c[2]


def n1(v):
    return float(np.abs(v).sum())


def g_triangle(kind, X, L, tri, h):
    """float64 model of point_bound's decision functions for surface point X, light L, triangle (v0, v1, v2), jitter half-width h."""
    v0, e1, e2 = tri[0], tri[1] - tri[0], tri[2] - tri[0]
    d = L - X
    start = X + 1e-4 * d
    dlen = float(np.linalg.norm(d))
    hh = 1.002 * h + 2e-6 * (dlen + h)
    b = start - v0
    c = cof(e1, e2)
    p, q = cof(b, e2), cof(e1, b)
    md = -d
    D0, N1, N2 = detc(md, c), detc(md, p), detc(md, q)
    if kind == "A1+": return N1 + hh * n1(p)
    if kind == "A1-": return N1 - hh * n1(p)
    if kind == "A2+": return N2 + hh * n1(q)
    if kind == "A2-": return N2 - hh * n1(q)
    sg = 1.0 if D0 >= 0 else -1.0
    if kind == "W":
        w = cof(b - e1, e2 - e1)
        e1n = n1(e1)
        slack = 4e-6 * (dlen + hh) * ((n1(p) + n1(q) + n1(c)) + (n1(b) + e1n) * (e1n + n1(e2)))
        return sg * detc(md, w) - hh * n1(w) - slack
    if kind == "R":
        nA0 = detc(b, c)
        dmin = max(dlen - 1.7321 * hh, 0.0)
        return abs(nA0) * dmin - (abs(D0) + hh * n1(c)) * dlen * 1.000004
    raise ValueError(kind)


def g_sphere(X, L, ctr, r2, h):
    """float64 model of spheres_point's miss certificate."""
    d = L - X
    start = X + 1e-4 * d
    dlen = float(np.linalg.norm(d))
    hh = 1.002 * h + 2e-6 * (dlen + h)
    Lv = start - ctr
    cr = np.cross(Lv, d)
    cl = np.cross(cr, Lv)
    crn = float(np.linalg.norm(cr))
    return crn * crn - hh * n1(cl) - np.sqrt(r2) * (dlen + 1.7321 * hh) * 1.002 * crn


def primary_hit(scene_aos, rot, cam, focal, W, H, aa, px, py):
    """float64 closest hit of the first AA ray of pixel (px, py) over the triangles: (index, X) or (-1, None)."""
    R = np.asarray(rot, np.float64).reshape(3, 4)[:, :3]
    dvec = np.array([px * aa - W * aa / 2.0, py * aa - H * aa / 2.0, focal])
    d = R @ dvec
    d /= np.linalg.norm(d)
    best, bt, X = -1, 1e300, None
    for i, t in enumerate(scene_aos):
        v0, e1, e2 = t[0, :3].astype(np.float64), (t[1, :3] - t[0, :3]).astype(np.float64), (t[2, :3] - t[0, :3]).astype(np.float64)
        pv = np.cross(d, e2)
        det = e1 @ pv
        if abs(det) < 1e-14:
            continue
        tv = np.asarray(cam, np.float64) - v0
        u = (tv @ pv) / det
        qv = np.cross(tv, e1)
        v = (d @ qv) / det
        tt = (e2 @ qv) / det
        if u >= 0 and v >= 0 and u + v <= 1 and 0 <= tt < bt:
            best, bt, X = i, tt, v0 + u * e1 + v * e2
    return best, X


def find_root(f, lo, hi, steps=240):
    xs = np.linspace(lo, hi, steps)
    vals = [f(x) for x in xs]
    for i in range(steps - 1):
        if np.isfinite(vals[i]) and np.isfinite(vals[i + 1]) and (vals[i] > 0) != (vals[i + 1] > 0):
            a, b, fa = xs[i], xs[i + 1], vals[i]
            for _ in range(80):
                m = 0.5 * (a + b)
                fm = f(m)
                if (fm > 0) == (fa > 0):
                    a, fa = m, fm
                else:
                    b = m
            return 0.5 * (a + b)
    return None


def ulp_step(x, k):
    x = np.float32(x)
    for _ in range(abs(k)):
        x = np.nextafter(x, np.float32(np.inf if k > 0 else -np.inf))
    return float(x)


def build_case(seed):
    """-> dict(scene, kw, rot, cam, focal, lights=[...], info) or None when the draw offers no threshold to aim at."""
    from test_gpu_cull import _random_scene
    rng = np.random.default_rng(7_000_000 + seed)
    kind = KINDS[seed % len(KINDS)]
    scene = _random_scene(rng, int(rng.integers(2, 9)), box=True)
    aos = scene.aos.copy()
    aos[26:, 4, 3] = 1.0                                   # the extra triangles cast shadows (diffuse)
    scene = rt.Scene(aos)
    W, H, aa = 24, 12, int(rng.choice([1, 2]))
    spread = float(rng.choice([0.05, 0.05, 0.2]))
    h = spread / 2.0
    cam = [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-3.3, -2.6))]
    rot = rt.rotation_matrix(float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.15, 0.15)))
    focal = 1100.0 * min(W, H) / 1024.0 * aa                # the reference's field of view along y; along x the 2:1 frame sees the whole room
    spheres = ()
    for _ in range(40):
        px, py = int(rng.integers(2, W - 2)), int(rng.integers(1, H - 1))
        hit, X = primary_hit(aos, rot, cam, focal, W, H, aa, px, py)
        if hit < 0 or aos[hit, 4, 3] <= 0.0:
            continue
        L0 = rng.uniform(-0.8, 0.8, 3)
        axis = int(rng.integers(0, 3))
        if kind.startswith("S"):
            R = float(rng.choice([0.004, 0.01, 0.02]))
            col = (0.5, 0.5, 0.5, float(rng.choice([1.0, 0.0])))        # a shadow-casting sphere (diffuse or mirror)
            dirn = (L0 - X) / np.linalg.norm(L0 - X)
            if kind == "S":
                # sphere somewhere along the shadow ray, the ray passing its centre at about one radius: then move the light
                ctr = X + rng.uniform(0.2, 0.7) * (L0 - X) + np.cross(dirn, rng.uniform(-1, 1, 3)) * R * 0.3
                spheres = ((tuple(float(np.float32(x)) for x in ctr), float(np.float32(R * R)), col),)
                c32, r32 = np.array(spheres[0][0], np.float64), float(np.float32(R * R))
                f = lambda s: g_sphere(X, L0 + s * np.eye(3)[axis], c32, r32, h)
                s = find_root(f, -0.6, 0.6)
                if s is None:
                    continue
                Lr = L0 + s * np.eye(3)[axis]
            else:
                ratio = 40.0 if kind == "S40" else 73.0
                # the START point of the shadow ray at ratio * R from the centre, the ray grazing the sphere
                off = np.cross(dirn, rng.uniform(-1, 1, 3))
                off /= np.linalg.norm(off)
                along = np.sqrt(max((ratio * R) ** 2 - (1.05 * R) ** 2, 0.0))
                ctr = X + 1e-4 * (L0 - X) + along * dirn + 1.05 * R * off
                if np.abs(ctr).max() > 0.98 or along > np.linalg.norm(L0 - X):
                    continue
                # the switch compares |start - centre| with 40 R: the light barely moves `start`, so here the RADIUS is stepped
                # (k ulp of radius_sq, and wider steps: the model's X is not the kernel's FP32 X) — one context per step
                c32 = tuple(float(np.float32(x)) for x in ctr)
                L32 = [float(np.float32(x)) for x in L0]
                variants = [(L32, ((c32, ulp_step(R * R, k), col),)) for k in OFFSETS + [-4096, -1024, -256, 256, 1024, 4096]]
                kw = dict(width=W, height=H, aa_x=aa, aa_y=aa, shadow_samples=64, light_spread=spread, max_bounces=2)
                return dict(scene=scene, kw=kw, rot=rot, cam=cam, focal=focal, variants=variants, kind=kind,
                            info="%s pixel (%d,%d) sphere R %g" % (kind, px, py, R))
            Lr32 = [float(np.float32(x)) for x in Lr]
            lights = [Lr32[:axis] + [ulp_step(Lr32[axis], k)] + Lr32[axis + 1:] for k in OFFSETS]
            info = "%s pixel (%d,%d) sphere R %g axis %d" % (kind, px, py, R, axis)
        else:
            cand = [i for i in range(len(aos)) if i != hit and aos[i, 4, 3] != -1.0]
            ti = int(rng.choice(cand))
            tri = aos[ti, :3, :3].astype(np.float64)
            f = lambda s: g_triangle(kind, X, L0 + s * np.eye(3)[axis], tri, h)
            s = find_root(f, -0.7, 0.7)
            if s is None:
                continue
            Lr = L0 + s * np.eye(3)[axis]
            if np.abs(Lr).max() > 0.97:
                continue
            Lr32 = [float(np.float32(x)) for x in Lr]
            lights = [Lr32[:axis] + [ulp_step(Lr32[axis], k)] + Lr32[axis + 1:] for k in OFFSETS]
            info = "%s pixel (%d,%d) surface %d caster %d axis %d" % (kind, px, py, hit, ti, axis)
        kw = dict(width=W, height=H, aa_x=aa, aa_y=aa, shadow_samples=64, light_spread=spread, max_bounces=2)
        return dict(scene=scene, kw=kw, rot=rot, cam=cam, focal=focal, variants=[(li, spheres) for li in lights], info=info, kind=kind)
    return None


def case(seed, oracle=None, verbose=True):
    """True when every light position renders identically on all device paths (and, with `oracle`, equals the CPU oracle
    at the threshold itself).  A draw without a threshold counts as passed (returns None)."""
    c = build_case(seed)
    if c is None:
        return None
    flags = [0, abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL]
    ok = True
    trs, made_for = [], None
    for vi, (light, spheres) in enumerate(c["variants"]):
        if spheres != made_for:                      # the sphere table is part of the context
            for tr in trs:
                tr.close()
            trs = [rt.RayTracer(abi.make_config(flags=fl, spheres=spheres, **c["kw"]), c["scene"]) for fl in flags]
            made_for = spheres
        ref = None
        for fl, tr in zip(flags, trs):
            a, f = tr.render(c["rot"], c["cam"], light, c["focal"], want_rgb=True)
            if ref is None:
                ref = (a, f)
            elif not (np.array_equal(a, ref[0]) and np.array_equal(f.view(np.uint32), ref[1].view(np.uint32))):
                bad = np.argwhere(a != ref[0])
                if verbose:
                    print("MISMATCH threshold seed %d (%s) variant %d flags %d: %d pixels, first %s" % (
                        seed, c["info"], vi, fl, len(bad), bad[:1].tolist()), flush=True)
                ok = False
        if oracle is not None and vi in (0, len(c["variants"]) // 2, len(c["variants"]) - 1):
            v, n, cc = c["scene"].packed()
            o_argb, _ = oracle.render(abi.make_config(spheres=spheres, **c["kw"]), v, n, cc, c["rot"], c["cam"], light, c["focal"], nthreads=4)
            if not np.array_equal(o_argb, ref[0].ravel()):
                if verbose:
                    print("MISMATCH threshold seed %d (%s) variant %d: device differs from the CPU oracle" % (seed, c["info"], vi), flush=True)
                ok = False
    for tr in trs:
        tr.close()
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ok = aimed = 0
    per_kind = {}
    for s in range(first, first + cases):
        r = case(s)
        if r is None:
            continue
        aimed += 1
        ok += bool(r)
        k = KINDS[s % len(KINDS)]
        per_kind[k] = per_kind.get(k, 0) + 1
        if aimed % 100 == 0:
            print("... %d aimed cases, %d ok" % (aimed, ok), flush=True)
    print("threshold fuzz: %d draws, %d aimed at a threshold (%s), %d identical on all paths at all %d+ offsets" % (
        cases, aimed, ", ".join("%s %d" % kv for kv in sorted(per_kind.items())), ok, len(OFFSETS)))
    return 0 if ok == aimed else 1


if __name__ == "__main__":
    sys.exit(main())
