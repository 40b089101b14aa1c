"""-m gpu: Loader.cpp-style OBJ meshes added to the Cornell Box (skeleton.cpp:102-103) — triangle counts
beyond the 64-triangle wave kernel and beyond one LDS stage (512), against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import focal_for
from uob_raytracer_amd import abi, meshgen, runtime as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_lon,n_lat,kw", [
    (10, 8, dict(width=96, height=64, aa_x=1, aa_y=1, shadow_samples=2)),              # 166 triangles: LDS stage
    (10, 8, dict(width=64, height=48, aa_x=2, aa_y=2, shadow_samples=3, spheres=())),
    (40, 30, dict(width=96, height=64, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0)),   # 2346: HBM records
    (40, 30, dict(width=48, height=32, aa_x=2, aa_y=1, shadow_samples=2)),
    (24, 16, dict(width=80, height=60, aa_x=2, aa_y=2, shadow_samples=64, light_spread=0.3)),     # wide penumbrae
    (12, 9, dict(width=70, height=41, aa_x=4, aa_y=2, shadow_samples=10, spheres=())),            # ragged frame
    # AA grids that do not divide 64 (tasks are runs of the block's Z curve, idle lanes) and more than 64 shadow samples
    # (passes of 64 sample lanes): on the tiled kernel too, no fall-back to the thread-per-pixel kernel
    (12, 9, dict(width=60, height=44, aa_x=3, aa_y=3, shadow_samples=5)),
    (10, 8, dict(width=48, height=36, aa_x=5, aa_y=1, shadow_samples=100, light_spread=0.2)),
    (24, 16, dict(width=40, height=30, aa_x=3, aa_y=2, shadow_samples=129, max_bounces=4)),
    (12, 9, dict(width=36, height=28, aa_x=7, aa_y=9, shadow_samples=3, spheres=())),             # 63 samples: one pixel per task
    (12, 9, dict(width=36, height=28, aa_x=8, aa_y=8, shadow_samples=70, band_rows=5, band_index=1, band_count=2)),
])
# tiled wave kernel with per-frame candidate-tile masks / the same visiting every tile / thread-per-pixel kernel
@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_TILE_BINS, abi.RT_FLAG_GENERIC_KERNEL])
def test_box_plus_mesh_vs_oracle(n_lon, n_lat, kw, flags, scene, oracle, tmp_path):
    path = str(tmp_path / "mesh.obj")
    nf = meshgen.write_sphere_obj(path, n_lon, n_lat, outward=(n_lon + kw["width"]) % 2 == 0)     # either winding
    both = scene + rt.Scene.load_obj(path)
    assert len(both) == 26 + nf
    cfg = abi.make_config(flags=flags, **kw)
    v, n, c = both.packed()
    tr = rt.RayTracer(cfg, both)
    for yaw, pitch, cam, light in [(0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]), (0.3, 0.1, [0.3, 0.2, -2.6], [0.2, -0.6, -0.4])]:
        rot = rt.rotation_matrix(yaw, pitch)
        argb, rgb = tr.render(rot, cam, light, focal_for(cfg), want_rgb=True)
        o_argb, o_rgb = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(cfg))
        assert np.array_equal(argb.ravel(), o_argb)
        assert np.array_equal(rgb[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))
        assert (argb != 0xFF000000).mean() > 0.5
    work = tr.count_work(rot, cam, light, focal_for(cfg))
    assert work["closest_tri_tests"] == (26 + nf) * (work["primary_rays"] + work["bounce_rays"])
    tr.close()


def test_mesh_casts_a_shadow_and_is_visible(scene, tmp_path):
    """Sanity of the placement: the mesh changes the frame (it is inside the view, on the floor)."""
    path = str(tmp_path / "mesh.obj")
    meshgen.write_sphere_obj(path, 16, 12)
    cfg = abi.make_config(width=128, height=128, aa_x=1, aa_y=1, shadow_samples=4)
    rot, cam, light = rt.rotation_matrix(0, 0), [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]
    a0 = rt.RayTracer(cfg, scene).render(rot, cam, light, focal_for(cfg))
    a1 = rt.RayTracer(cfg, scene + rt.Scene.load_obj(path)).render(rot, cam, light, focal_for(cfg))
    assert 0.01 < (a0 != a1).mean() < 0.5


@pytest.mark.parametrize("seed", range(6))
def test_random_soups_tiled_kernel_equals_generic(seed):
    """Random triangle soups of 70..330 triangles (slivers, degenerate, mirrors, glass) in the room: the tiled
    wave kernel (default for n > 64) and the thread-per-pixel kernel must agree bit for bit."""
    from test_gpu_cull import _random_scene, _render
    rng = np.random.default_rng(7000 + seed)
    scene = _random_scene(rng, int(rng.integers(70, 330)), box=bool(seed % 2))
    kw = dict(width=96, height=72, aa_x=[1, 2, 4][seed % 3], aa_y=[1, 2, 2][seed % 3],
              shadow_samples=[64, 8, 21][seed % 3], light_spread=[0.05, 0.25][seed % 2], max_bounces=[4, 10][seed % 2],
              band_rows=[72, 8][seed % 2], band_index=[0, 1][seed % 2], band_count=[1, 3][seed % 2])
    light = rng.uniform(-0.9, 0.9, 3).tolist()
    cam = [float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), -3.0]
    rot = rt.rotation_matrix(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)))
    a0, f0 = _render(kw, 0, scene, rot, cam, light)
    a1, f1 = _render(kw, abi.RT_FLAG_NO_TILE_BINS, scene, rot, cam, light)
    a2, f2 = _render(kw, abi.RT_FLAG_GENERIC_KERNEL, scene, rot, cam, light)
    bad = np.argwhere(a0 != a2)
    assert bad.size == 0, "tiled kernel differs in %d pixels, first at %s" % (len(bad), bad[0])
    assert np.array_equal(f0.view(np.uint32), f2.view(np.uint32))
    assert np.array_equal(a1, a2) and np.array_equal(f1.view(np.uint32), f2.view(np.uint32))


@pytest.mark.parametrize("case", range(4))
def test_tile_masks_never_change_a_pixel(case, scene, tmp_path):
    """Bigger meshes (tens of tiles), lights inside / far outside the scene box, a camera inside the mesh's
    silhouette, bands: candidate-tile masks on == off, bit for bit."""
    from test_gpu_cull import _render
    path = str(tmp_path / "mesh.obj")
    meshgen.write_sphere_obj(path, [60, 90, 48, 72][case], [40, 60, 36, 50][case])
    both = scene + rt.Scene.load_obj(path)
    kw = [dict(width=160, height=128, aa_x=1, aa_y=1, shadow_samples=1),
          dict(width=128, height=96, aa_x=2, aa_y=2, shadow_samples=8, light_spread=0.2),
          dict(width=200, height=130, aa_x=2, aa_y=1, shadow_samples=16, band_rows=10, band_index=2, band_count=3),
          dict(width=144, height=144, aa_x=1, aa_y=1, shadow_samples=64, light_spread=0.6, spheres=())][case]
    light = [[0.0, -0.5, -0.7], [-0.45, 0.8, -0.55], [3.0, -4.0, -6.0], [-0.4, 0.2, -0.5]][case]
    cam = [[0.0, 0.0, -3.2], [0.2, 0.3, -2.0], [-0.3, 0.6, -1.4], [0.0, 0.0, -3.2]][case]
    rot = rt.rotation_matrix([0.0, 0.2, -0.3, 0.0][case], [0.0, -0.1, 0.2, 0.0][case])
    a0, f0 = _render(kw, 0, both, rot, cam, light)
    a1, f1 = _render(kw, abi.RT_FLAG_NO_TILE_BINS, both, rot, cam, light)
    bad = np.argwhere(a0 != a1)
    assert bad.size == 0, "tile masks changed %d pixels, first at %s" % (len(bad), bad[0])
    assert np.array_equal(f0.view(np.uint32), f1.view(np.uint32))


@pytest.mark.parametrize("coop_all", [False, True])
@pytest.mark.parametrize("case", range(5))
def test_frames_in_sequence_equal_the_first(case, coop_all, scene, oracle, tmp_path, monkeypatch):
    """The mesh kernel carries scheduling state from frame to frame: last frame's expensive 16x16-pixel blocks go first,
    the dearest ones as four COOPERATIVE sub-block jobs (the four waves share the tiles and merge per lane through LDS).
    No pixel may depend on it: five frames of one context — with UOB_RT_MASK_DEBUG=8 every block of frames 2.. is
    cooperative — against the CPU oracle, bit for bit."""
    if coop_all:
        monkeypatch.setenv("UOB_RT_MASK_DEBUG", "8")          # read once, in rt_init
    path = str(tmp_path / "mesh.obj")
    meshgen.write_sphere_obj(path, [24, 40, 12, 30, 60][case], [16, 30, 9, 20, 40][case])
    both = scene + rt.Scene.load_obj(path)
    kw = [dict(width=96, height=80, aa_x=1, aa_y=1, shadow_samples=1, spheres=()),
          dict(width=80, height=64, aa_x=2, aa_y=2, shadow_samples=16, light_spread=0.2),
          dict(width=64, height=48, aa_x=3, aa_y=3, shadow_samples=5),
          dict(width=72, height=56, aa_x=2, aa_y=1, shadow_samples=100, max_bounces=4),
          dict(width=128, height=112, aa_x=1, aa_y=1, shadow_samples=2, band_rows=16, band_index=1, band_count=2)][case]
    cfg = abi.make_config(**kw)
    v, n, c = both.packed()
    tr = rt.RayTracer(cfg, both)
    poses = [(0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]), (0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]),
             (0.2, 0.1, [0.2, 0.1, -2.8], [0.2, -0.6, -0.4]), (0.2, 0.1, [0.2, 0.1, -2.8], [0.2, -0.6, -0.4]),
             (0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7])]
    wants = {}
    for k, (yaw, pitch, cam, light) in enumerate(poses):
        rot = rt.rotation_matrix(yaw, pitch)
        argb, rgb = tr.render(rot, cam, light, focal_for(cfg), want_rgb=True)
        key = (yaw, pitch)
        if key not in wants:
            wants[key] = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(cfg))
        o_argb, o_rgb = wants[key]
        bad = np.argwhere(argb.ravel() != o_argb)
        assert bad.size == 0, "frame %d: %d pixels differ, first %s" % (k, len(bad), bad[0])
        assert np.array_equal(rgb[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))
    costs = tr.block_costs()
    assert costs.shape == ((tr.rows + 15) // 16, (cfg.width + 15) // 16) and (costs > 0).all()
    ntiles = (len(both) + 63) // 64
    if len(both) > 16 * 64:                # diagnostic: the shadow-ray tile masks of the last frame (built from 17 tiles on)
        masks = tr.world_masks()
        assert masks.ndim == 4 and masks.shape[0] == masks.shape[1] == masks.shape[2] and masks.shape[3] == (ntiles + 63) // 64
        assert masks.any()
        if ntiles % 64:
            assert not (masks[..., -1] >> np.uint64(ntiles % 64)).any()       # no tile beyond the last
    else:
        with pytest.raises(rt.RtError) as e:
            tr.world_masks()
        assert e.value.code == abi.RT_E_UNSUPPORTED
    tr.close()
