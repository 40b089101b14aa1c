"""Shared by the tests that compare a strict FP32 image (CPU oracle or HIP product) with frames THE REFERENCE ITSELF
rendered on the MI355X's OpenCL device (tests/golden/ref_gfx950.npz, made by tests/golden/make_ref_gpu_golden.py).

The tolerance, stated once (BASELINE.json north star: "per-pixel match to reference within 1e-4"):
  * the reference's only output is 8-bit ARGB; two colours within 1e-4 of each other quantise (kernels.cl:37-40,
    truncation of 255*c) to channels that differ by at most 1 — so "within tolerance" = every channel within 1 LSB;
  * the reference on this device and the strict oracle do not use the same FP32 roundings (v_rcp_f32 for native_recip,
    fused multiply-adds, AMD's normalize/dot — SURVEY.md appendix A), and a ray-triangle test has discontinuities: a
    primary or shadow ray within a rounding error of a triangle's edge hits under one rounding and misses under the
    other, which moves a pixel by up to 0.7.  Such pixels exist only ON discontinuities, so the statement is:
      (a) at least MIN_FRACTION of the pixels are within tolerance (0.99: the default, axis-aligned view puts whole pixel
          rows and columns exactly on the box's silhouette; a general view gives > 0.9999),
      (b) EVERY pixel beyond tolerance lies within 1 pixel of a discontinuity (an 8-neighbour differing by more than
          EDGE_LSB in some channel) of the strict image or of the reference's own image.
"""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")
MIN_FRACTION = 0.99
MIN_FRACTION_GENERAL_VIEW = 0.9999
EDGE_LSB = 4


def load():
    meta = json.load(open(os.path.join(G, "ref_gfx950.json")))
    return meta, np.load(os.path.join(G, "ref_gfx950.npz"))


def load_mesh():
    """Box + OBJ-mesh frames (make_ref_gpu_golden.py --mesh): the shape of the reference's main(), skeleton.cpp:102-103."""
    meta = json.load(open(os.path.join(G, "ref_gfx950_mesh.json")))
    return meta, np.load(os.path.join(G, "ref_gfx950_mesh.npz"))


def mesh_scene(lon, lat, outward=True):
    """AoS [n,5,4]: the reference's Cornell Box + uob_raytracer_amd/meshgen.py's mesh read by the product's Loader.cpp
    counterpart (host code; tests/test_scene.py pins that loader against the reference's own)."""
    import tempfile
    from uob_raytracer_amd import meshgen, runtime as rt
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.obj")
        meshgen.write_sphere_obj(path, lon, lat, outward=bool(outward))
        return (rt.Scene(np.load(os.path.join(G, "scene_cornell_aos.npy"))) + rt.Scene.load_obj(path)).aos


def channels(argb):
    a = np.asarray(argb).astype(np.uint32)
    return np.stack([((a >> s) & 255).astype(np.int32) for s in (16, 8, 0)], -1)


def discontinuities(img_ch, lsb=EDGE_LSB):
    """Pixels with an 8-neighbour that differs by more than `lsb` in some channel ([H,W,3] int -> [H,W] bool)."""
    H, W = img_ch.shape[:2]
    p = np.pad(img_ch, ((1, 1), (1, 1), (0, 0)), mode="edge")
    e = np.zeros((H, W), bool)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dx != 1 or dy != 1:
                e |= np.abs(p[dy:dy + H, dx:dx + W] - img_ch).max(-1) > lsb
    return e


def compare(strict_argb, ref_argb):
    """Statistics of a strict image against the reference's image of the same frame (2-D uint32 arrays)."""
    s, r = channels(strict_argb), channels(ref_argb)
    d = np.abs(s - r).max(-1)
    beyond = d > 1
    edges = discontinuities(s) | discontinuities(r)
    return {"pixels": int(d.size), "identical": int((d == 0).sum()), "within_tolerance": int((d <= 1).sum()),
            "beyond_tolerance": int(beyond.sum()), "beyond_off_discontinuity": int((beyond & ~edges).sum()),
            "fraction_within": float((d <= 1).mean()), "max_channel_diff": int(d.max()),
            "alpha_ok": bool(((np.asarray(ref_argb) >> 24) == 255).all())}


def assert_within_tolerance(strict_argb, ref_argb, what, general_view=False):
    st = compare(strict_argb, ref_argb)
    need = MIN_FRACTION_GENERAL_VIEW if general_view else MIN_FRACTION
    assert st["alpha_ok"], what
    assert st["fraction_within"] >= need, "%s: only %.5f of the pixels within 1 LSB of the reference's (%r)" % (
        what, st["fraction_within"], st)
    assert st["beyond_off_discontinuity"] == 0, "%s: %d pixels beyond tolerance away from any discontinuity (%r)" % (
        what, st["beyond_off_discontinuity"], st)
    return st


def focal_for(kw):
    return 1100.0 * min(kw["width"], kw["height"]) / 1024.0 * kw.get("aa_x", 2)


def config_kwargs(kw):
    kw = dict(kw)
    if "spheres" in kw:
        kw["spheres"] = tuple(kw["spheres"])
    return kw


def scenes():
    """name -> Cornell Box AoS [26,5,4] (the reference's own LoadTestModel output) with the named change."""
    aos = np.load(os.path.join(G, "scene_cornell_aos.npy"))
    mw = aos.copy()
    mw[[8, 9], 4, :] = (1.0, 1.0, 1.0, 0.0)          # back wall -> mirror (TestModelH.h:58)
    return {"box": aos, "mirrorwall": mw}
