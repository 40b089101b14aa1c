"""-m gpu: FUNCTION-level parity.  The device forms of in_shadow (kernels.cl:243-311) and
single_ray_intersections (:168-241), run on caller rays through the C ABI (rt_debug_trace_rays), against the
20 000-ray golden vectors produced by the REAL reference functions (tests/golden/function_vectors.npz, made by
tests/golden/make_golden.py from Source/kernels.cl compiled for x86-64).  Tolerance: 0."""
import os

import numpy as np
import pytest

from uob_raytracer_amd import abi, meshgen, runtime as rt

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(G, "function_vectors.npz"))


def test_in_shadow_against_reference_vectors(scene, vectors):
    tr = rt.RayTracer(abi.make_config(width=64, height=64), scene)       # reference sphere table
    got = tr.trace_in_shadow(vectors["rays"], vectors["radius_sq"])
    tr.close()
    assert np.array_equal(got, vectors["in_shadow"])
    assert 0.05 < got.mean() < 0.95


def test_closest_hit_against_reference_vectors(scene, vectors):
    tr = rt.RayTracer(abi.make_config(width=64, height=64), scene)
    tri, out = tr.trace_closest_hit(vectors["rays_unit"])
    tr.close()
    assert np.array_equal(tri, vectors["hit_tri"])
    hit = tri != -1
    assert np.array_equal(out[hit].view(np.uint32), vectors["hit_out"][hit].view(np.uint32))
    assert set(np.unique(tri)) >= {-2, 0}


@pytest.mark.parametrize("n_lon,n_lat", [(10, 8), (40, 30)])       # one LDS stage / HBM-resident records
def test_functions_on_a_mesh_against_the_oracle(n_lon, n_lat, scene, oracle, vectors, tmp_path):
    """The same entry on box + OBJ mesh (beyond what the reference's local memory holds): oracle as the checker."""
    path = str(tmp_path / "m.obj")
    meshgen.write_sphere_obj(path, n_lon, n_lat)
    both = scene + rt.Scene.load_obj(path)
    cfg = abi.make_config(width=64, height=64)
    v, n, c = both.packed()
    tr = rt.RayTracer(cfg, both)
    rays, r2 = vectors["rays"][:4000], vectors["radius_sq"][:4000]
    assert np.array_equal(tr.trace_in_shadow(rays, r2), oracle.in_shadow(cfg, v, c, rays, r2))
    tri, out = tr.trace_closest_hit(vectors["rays_unit"][:4000])
    o_tri, o_out = oracle.closest_hit(cfg, v, n, c, vectors["rays_unit"][:4000])
    tr.close()
    assert np.array_equal(tri, o_tri)
    hit = tri != -1
    assert np.array_equal(out[hit].view(np.uint32), o_out[hit].view(np.uint32))
    assert (tri >= 26).any()                                           # some rays hit the mesh


def test_wave_timeline_diagnostic(scene, monkeypatch):
    """UOB_RT_TIMELINE (read once in rt_init): the shipped wave kernel leaves one (start, end, jobs) record per wave; the
    frame is the same with and without, and a context created without the knob refuses the query."""
    cfg = abi.make_config(width=512, height=256, aa_x=2, aa_y=2, shadow_samples=16)
    rot, cam, light = rt.rotation_matrix(0.1, 0.0), [0, 0, -3.2], [0, -0.5, -0.7]
    plain = rt.RayTracer(cfg, scene)
    ref = plain.render(rot, cam, light, 1100.0)
    with pytest.raises(rt.RtError):
        plain.wave_timeline()
    plain.close()
    monkeypatch.setenv("UOB_RT_TIMELINE", "1")
    tr = rt.RayTracer(cfg, scene)
    monkeypatch.delenv("UOB_RT_TIMELINE")
    for _ in range(3):
        got = tr.render(rot, cam, light, 1100.0)
        t = tr.wave_timeline()
        assert np.array_equal(got, ref)
        assert t["waves"] > 0 and t["span_us"] > 0 and 0 <= t["mean_idle_tail_us"] <= t["span_us"]
        assert t["jobs"] >= 512 * 256 // 64 and t["max_jobs_per_wave"] >= 1      # every job taken by exactly one wave
    tr.close()


def test_registered_output_is_the_same_frame(scene):
    """rt_register_output: the kernel writes straight into the caller's (pinned, mapped) host framebuffer; same pixels as the
    device-buffer + copy path, also for a sub-range of the registered memory, and again after unregistering."""
    cfg = abi.make_config(width=300, height=200, aa_x=2, aa_y=2, shadow_samples=16)
    rot, cam, light = rt.rotation_matrix(0.2, -0.1), [0.1, 0, -3.0], [0.2, -0.5, -0.7]
    tr = rt.RayTracer(cfg, scene)
    ref = tr.render(rot, cam, light, 600.0)
    big = np.zeros((3, 200, 300), np.uint32)
    tr.register_output(big)
    for k in (1, 0, 2):
        got = tr.render(rot, cam, light, 600.0, out=big[k])
        assert np.array_equal(got, ref)
    other = np.zeros((200, 300), np.uint32)                    # not registered: the copy path
    assert np.array_equal(tr.render(rot, cam, light, 600.0, out=other), ref)
    tr.unregister_output()
    big[:] = 0
    assert np.array_equal(tr.render(rot, cam, light, 600.0, out=big[1]), ref)
    tr.close()
    # several device entries: each writes its own bands into the registered frame
    for devs, h in (((0, 0), 200), ((0, 0, 0), 173)):
        cfgm = abi.make_config(width=300, height=h, aa_x=2, aa_y=2, shadow_samples=16, devices=devs, device_band_rows=16)
        one = rt.RayTracer(abi.make_config(width=300, height=h, aa_x=2, aa_y=2, shadow_samples=16), scene)
        want = one.render(rot, cam, light, 600.0)
        one.close()
        multi = rt.RayTracer(cfgm, scene)
        frame = np.zeros((h, 300), np.uint32)
        multi.register_output(frame)
        for _ in range(2):
            frame[:] = 0
            assert np.array_equal(multi.render(rot, cam, light, 600.0, out=frame), want)
        multi.unregister_output()
        frame[:] = 0
        assert np.array_equal(multi.render(rot, cam, light, 600.0, out=frame), want)
        multi.close()


def test_listed_jobs_task_by_task_is_the_same_frame(scene, monkeypatch):
    """UOB_RT_SPLIT_LISTED=1 (read once in rt_init): last frame's expensive jobs are handed out one task at a time; pixels
    do not depend on it — three frames of one context (the list exists from the second on), whole frame and a band rank."""
    rot, cam, light = rt.rotation_matrix(-0.1, 0.05), [0, 0, -3.2], [0.1, -0.5, -0.7]
    for extra in ({}, {"band_rows": 32, "band_index": 1, "band_count": 3}):
        cfg = abi.make_config(width=1024, height=768, aa_x=4, aa_y=2, shadow_samples=64, **extra)
        plain = rt.RayTracer(cfg, scene)
        ref = plain.render(rot, cam, light, 4400.0)
        plain.close()
        monkeypatch.setenv("UOB_RT_SPLIT_LISTED", "1")
        tr = rt.RayTracer(cfg, scene)
        monkeypatch.delenv("UOB_RT_SPLIT_LISTED")
        for _ in range(3):
            assert np.array_equal(tr.render(rot, cam, light, 4400.0), ref)
        tr.close()
