"""-m gpu: the HIP path through the C ABI against golden vectors produced by the REAL reference kernel
(tests/golden/, made by make_golden.py from Source/kernels.cl compiled for x86-64).  Tolerance: 0 ULP on
the float tap and identical ARGB words (the north star allows 1e-4)."""
import json
import os

import numpy as np
import pytest

from oracle import pyref
from uob_raytracer_amd import abi, runtime as rt

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(G, "golden.json")))
POSES = META["poses"]


def focal_for(kw):
    return 1100.0 * min(kw["width"], kw["height"]) / 1024.0 * kw.get("aa_x", 2)


def cfg_of(kw, **extra):
    kw = dict(kw)
    if "spheres" in kw:
        kw["spheres"] = tuple(kw["spheres"])
    kw.update(extra)
    return abi.make_config(**kw)


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(G, "frames_small.npz"))


@pytest.fixture(scope="module")
def subsets():
    return np.load(os.path.join(G, "frames_subsets.npz"))


@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_GENERIC_KERNEL])
@pytest.mark.parametrize("name", sorted(META["frames"]))
def test_small_frames(name, flags, scene, small):
    kw = META["frames"][name]
    tr = rt.RayTracer(cfg_of(kw, flags=flags), scene)
    for pi, (yaw, pitch, cam, light) in enumerate(POSES):
        argb, rgb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, focal_for(kw), want_rgb=True)
        assert np.array_equal(argb, small["%s_p%d_argb" % (name, pi)])
        _, tap = pyref.quantise(rgb[..., :3])
        assert np.array_equal(tap.view(np.uint32), small["%s_p%d_tap" % (name, pi)].view(np.uint32))
    tr.close()


def test_mirror_wall_scene(scene, small):
    s = scene.with_color([8, 9], (1.0, 1.0, 1.0, 0.0))
    kw = META["frames"]["cfg3_480"]
    yaw, pitch, cam, light = POSES[1]
    tr = rt.RayTracer(cfg_of(kw), s)
    argb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, focal_for(kw))
    tr.close()
    assert np.array_equal(argb, small["cfg3_480_mirrorwall_argb"])


@pytest.mark.parametrize("name", sorted(set(META["big"]) - {"default_fast"}))
def test_large_frames_hash_and_subset(name, scene, subsets):
    """Whole-frame FNV-1a-64 known answers of the reference (e.g. e9a893e34410ff28 for the shipped
    1024^2 configuration) plus exact values on a seeded pixel subset."""
    kw = META["big"][name]["config"]
    yaw, pitch, cam, light = POSES[0]
    tr = rt.RayTracer(cfg_of(kw), scene)
    argb, rgb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, focal_for(kw), want_rgb=True)
    tr.close()
    assert "%016x" % pyref.fnv1a64_words(argb) == META["big"][name]["fnv_words"]
    assert int((argb == 0xFF000000).sum()) == META["big"][name]["black_pixels"]
    pix = subsets[name + "_pix"]
    assert np.array_equal(argb.ravel()[pix], subsets[name + "_argb"])
    _, tap = pyref.quantise(rgb.reshape(-1, 4)[pix, :3])
    assert np.array_equal(tap.view(np.uint32), subsets[name + "_tap"].view(np.uint32))


@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL])
def test_full_size_4096_s64_against_reference_values(flags, scene, subsets):
    """4096x4096, 64 shadow samples (the headline size and sample count; 2x2 AA because the reference
    cannot express 4x2): 20000 seeded pixels rendered by the reference kernel itself."""
    kw = META["subset"]["s64_4096"]
    yaw, pitch, cam, light = POSES[0]
    tr = rt.RayTracer(cfg_of(kw, flags=flags), scene)
    argb, rgb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, focal_for(kw), want_rgb=True)
    tr.close()
    pix = subsets["s64_4096_pix"]
    assert np.array_equal(argb.ravel()[pix], subsets["s64_4096_argb"])
    _, tap = pyref.quantise(rgb.reshape(-1, 4)[pix, :3])
    assert np.array_equal(tap.view(np.uint32), subsets["s64_4096_tap"].view(np.uint32))


def test_fast_math_reference_build_within_tolerance(scene, subsets):
    """Against the reference's OWN build options (-cl-fast-relaxed-math): |colour diff| < 1e-4."""
    kw = META["big"]["default_fast"]["config"]
    yaw, pitch, cam, light = POSES[0]
    tr = rt.RayTracer(cfg_of(kw), scene)
    _, rgb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, focal_for(kw), want_rgb=True)
    tr.close()
    _, tap = pyref.quantise(rgb.reshape(-1, 4)[subsets["default_fast_pix"], :3])
    assert np.nanmax(np.abs(tap - subsets["default_fast_tap"])) / 255.0 < 1e-4


def test_headline_frame_properties_and_oracle_subset(scene, oracle):
    """BASELINE.json's headline frame (4096^2, 4x2 AA, 64 samples): a grid the reference cannot express,
    so the checker is the CPU oracle on 3000 seeded pixels, plus size-independent properties:
    determinism, band-partition invariance (3 ragged bands == whole frame) and the black border."""
    kw = dict(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64)
    yaw, pitch, cam, light = POSES[0]
    rot = rt.rotation_matrix(yaw, pitch)
    cfg = abi.make_config(**kw)
    focal = focal_for(kw)
    tr = rt.RayTracer(cfg, scene)
    a0 = tr.render(rot, cam, light, focal)
    a1 = tr.render(rot, cam, light, focal)
    tr.close()
    assert np.array_equal(a0, a1)                                    # idempotent / deterministic
    assert (a0[0] == 0xFF000000).all() and (a0[:, 0] == 0xFF000000).all()
    pix = np.sort(np.random.default_rng(7).choice(4096 * 4096, 3000, replace=False)).astype(np.int32)
    v, n, c = scene.packed()
    o_argb, _ = oracle.render(cfg, v, n, c, rot, cam, light, focal, pix=pix, nthreads=16)
    assert np.array_equal(a0.ravel()[pix], o_argb)
    rebuilt = np.zeros_like(a0)
    for r in range(3):                                               # 4096 = 42*96 + 64: ragged last band
        cfgb = abi.make_config(band_rows=96, band_index=r, band_count=3, **kw)
        trb = rt.RayTracer(cfgb, scene)
        rows = [y for y in range(4096) if (y // 96) % 3 == r]
        rebuilt[rows] = trb.render(rot, cam, light, focal)
        trb.close()
    assert np.array_equal(rebuilt, a0)


def test_exact_reciprocal_selftest():
    """v_rcp_f32 + one Newton step == IEEE 1/x for every FP32 x with 2^-100 <= |x| <= 2^100 (all 2^32
    patterns swept on the device); the documented exceptions are denormal x and |x| >= 2^126 only."""
    r = rt.selftest_rcp()
    assert r["safe_mismatch_1step"] == 0 and r["safe_mismatch_2step"] == 0
    assert r["edge_mismatch_1step"] <= 2 * (1 << 23) + 4 * (1 << 23)   # denormals + two top binades, both signs


def test_normalize_building_blocks_selftest():
    """normalize3 (rt_math.h): the refined v_rsq_f32 == sqrtf for every pattern in [2^-60, 2^60], and the quotient from the
    shared exact reciprocal == a / b for every significand of a against every 97th significand of b (8.6e4 x 8.4e6 pairs
    here; all 2^46 pairs with tools/div_check.hip / rt_selftest_normalize(out, 1): no mismatch, profiles/README.md)."""
    r = rt.selftest_normalize(97)
    assert r["sqrt_mismatches"] == 0, hex(r["sqrt_example"])
    assert r["div_mismatches"] == 0, hex(r["div_example"])
    assert r["div_pairs"] >= (1 << 23) * ((1 << 23) // 97)


def test_normalize_zero_components_take_the_ieee_path(scene, oracle):
    """An odd frame width at the identity rotation: the rays of the centre column have dir.x == 0 exactly, the centre row's
    dir.y == 0 — components outside normalize3's checked range, which take the IEEE square root and divisions behind its
    wave-uniform branch.  Identical to the oracle either way (the wide-domain cases of test_gpu_thresholds.py cover the
    scales at which whole vectors leave the range)."""
    kw = dict(width=65, height=33, aa_x=1, aa_y=1, shadow_samples=3)
    cfg = abi.make_config(**kw)
    v, n, c = scene.packed()
    rot = rt.rotation_matrix(0.0, 0.0)
    cam, light = [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]
    for flags in (0, abi.RT_FLAG_GENERIC_KERNEL):
        tr = rt.RayTracer(abi.make_config(flags=flags, **kw), scene)
        argb, rgb = tr.render(rot, cam, light, focal_for(kw), want_rgb=True)
        tr.close()
        o_argb, o_rgb = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(kw))
        assert np.array_equal(argb.ravel(), o_argb)
        assert np.array_equal(rgb[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))


def test_work_counters_match_oracle(scene, oracle):
    kw = dict(width=96, height=64, shadow_samples=5)
    yaw, pitch, cam, light = POSES[1]
    cfg = abi.make_config(**kw)
    rot = rt.rotation_matrix(yaw, pitch)
    tr = rt.RayTracer(cfg, scene)
    got = tr.count_work(rot, cam, light, focal_for(kw))
    tr.close()
    v, n, c = scene.packed()
    _, _, want = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(kw), want_work=True)
    assert got == want
