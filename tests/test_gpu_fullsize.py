"""-m gpu: BASELINE.json's configurations at FULL size (the 4096^2 headline has its own tests in test_gpu_cull.py /
test_gpu_headline.py):

  configs[4]  Cornell Box + Loader.cpp OBJ mesh of 100 026 triangles (1 563 LDS tiles, 25 mask words), 2048 x 2048,
              1 spp, 1 shadow ray — and the same with the reference's two spheres and 16 shadow rays, so that the
              bounce / penumbra branches run at that tile count;
  configs[2]  Cornell Box + glass sphere + MIRROR WALL, 1920 x 1080, 4xAA, recursion depth 5.

The CPU oracle cannot render these frames whole in test time (configs[4]: ~200 s on 16 threads), so: oracle parity
on seeded pixel subsets, plus the size-independent properties — determinism, tile masks on == off, a 3-band
partition == the whole frame."""
import numpy as np
import pytest

from conftest import DEFAULT_CAM, DEFAULT_LIGHT, focal_for
from uob_raytracer_amd import abi, meshgen, runtime as rt

pytestmark = pytest.mark.gpu

ROT0 = (0.0, 0.0)


@pytest.fixture(scope="module")
def box_and_big_mesh(scene, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("cfg5") / "mesh_100k.obj")
    nf = meshgen.write_sphere_obj(path, 250, 201)
    assert nf == 100000
    both = scene + rt.Scene.load_obj(path)
    assert len(both) == 100026
    return both


def _render(cfg, scene, want_rgb=False):
    tr = rt.RayTracer(cfg, scene)
    out = tr.render(rt.rotation_matrix(*ROT0), DEFAULT_CAM, DEFAULT_LIGHT, focal_for(cfg), want_rgb=want_rgb)
    ms = tr.last_kernel_ms()
    tr.close()
    return out, ms


CFG5 = {
    "cfg5_1spp_1shadow": dict(width=2048, height=2048, aa_x=1, aa_y=1, shadow_samples=1, spheres=()),
    "cfg5_spheres_s16": dict(width=2048, height=2048, aa_x=1, aa_y=1, shadow_samples=16),
}


@pytest.mark.parametrize("name", list(CFG5))
def test_configs4_full_size(name, box_and_big_mesh, oracle):
    kw = CFG5[name]
    both = box_and_big_mesh
    cfg = abi.make_config(**kw)
    (argb, rgb), ms = _render(cfg, both, want_rgb=True)
    print("%s: %.1f ms" % (name, ms))
    assert (argb != 0xFF000000).mean() > 0.8
    # determinism
    (again, _), _ = _render(cfg, both, want_rgb=True)
    assert np.array_equal(argb, again)
    # tile masks on == off (every one of the 1 563 tiles visited)
    plain, ms_plain = _render(abi.make_config(flags=abi.RT_FLAG_NO_TILE_BINS, **kw), both)
    print("%s without tile masks: %.1f ms" % (name, ms_plain))
    bad = np.argwhere(plain != argb)
    assert bad.size == 0, "tile masks changed %d pixels, first at %s" % (len(bad), bad[0])
    # a 3-band partition (bands of 48 rows: 2048 = 42 * 48 + 32, ragged) == the whole frame
    rebuilt = np.zeros_like(argb)
    for r in range(3):
        part, _ = _render(abi.make_config(band_rows=48, band_index=r, band_count=3, **kw), both)
        rows = [y for y in range(2048) if (y // 48) % 3 == r]
        rebuilt[rows] = part
    assert np.array_equal(rebuilt, argb)
    # CPU oracle on 2 400 seeded pixels: 1 200 anywhere, 1 200 inside the mesh's screen box (its silhouette, its shadow)
    rng = np.random.default_rng(424242)
    ys, xs = np.nonzero(argb != plain_box_pixels(cfg))
    box = (ys.min(), ys.max(), xs.min(), xs.max())
    inside = rng.integers(box[0], box[1] + 1, 1200) * 2048 + rng.integers(box[2], box[3] + 1, 1200)
    pix = np.unique(np.concatenate([rng.choice(2048 * 2048, 1200, replace=False), inside])).astype(np.int32)
    v, n, c = both.packed()
    o_argb, o_rgb = oracle.render(cfg, v, n, c, rt.rotation_matrix(*ROT0), DEFAULT_CAM, DEFAULT_LIGHT, focal_for(cfg),
                                  pix=pix, nthreads=16)
    assert len(pix) >= 2000
    assert np.array_equal(argb.ravel()[pix], o_argb)
    assert np.array_equal(rgb.reshape(-1, 4)[pix, :3].view(np.uint32), o_rgb.view(np.uint32))


_PLAIN = {}


def plain_box_pixels(cfg):
    """The same frame without the mesh (26 triangles): where it differs is the mesh and its shadow."""
    key = (cfg.shadow_samples, cfg.num_spheres)
    if key not in _PLAIN:
        tr = rt.RayTracer(cfg, rt.Scene.cornell_box())
        _PLAIN[key] = tr.render(rt.rotation_matrix(*ROT0), DEFAULT_CAM, DEFAULT_LIGHT, focal_for(cfg))
        tr.close()
    return _PLAIN[key]


def test_configs4_masks_stay_sparse(box_and_big_mesh):
    """How much the bounds can certify shows in the per-frame shadow-ray masks (rt_debug_world_masks): of the 1 563 tiles an
    occupied world cell names 34 on average (49 before the |u| > 1 clause, 52 before the tile-level certificate).  A guard
    against a bound silently losing a clause: frames would stay identical, only slower."""
    tr = rt.RayTracer(abi.make_config(**CFG5["cfg5_1spp_1shadow"]), box_and_big_mesh)
    tr.render(rt.rotation_matrix(*ROT0), DEFAULT_CAM, DEFAULT_LIGHT, focal_for(tr.cfg))
    masks = tr.world_masks()
    tr.close()
    per_cell = np.unpackbits(masks.view(np.uint8), axis=-1).sum(-1)
    occupied = per_cell[per_cell > 0]
    assert 5000 < occupied.size < 12000
    assert occupied.mean() < 40.0, occupied.mean()


@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_CULL])
def test_configs2_mirror_wall_1080p(flags, scene, oracle):
    """configs[2] as BASELINE.json words it: glass sphere + mirror wall, 1920x1080, 4xAA, depth 5 (back wall -> mirror,
    TestModelH.h:58; the reference's own glass and mirror spheres; 10 shadow rays)."""
    s = scene.with_color([8, 9], (1.0, 1.0, 1.0, 0.0))
    kw = dict(width=1920, height=1080, max_bounces=5)
    cfg = abi.make_config(flags=flags, **kw)
    tr = rt.RayTracer(cfg, s)
    poses = [(0.0, 0.0, DEFAULT_CAM, DEFAULT_LIGHT), (0.35, -0.1, [0.3, 0.1, -2.7], [-0.3, -0.5, -0.6])]
    v, n, c = s.packed()
    for yaw, pitch, cam, light in poses:
        rot = rt.rotation_matrix(yaw, pitch)
        argb, rgb = tr.render(rot, cam, light, focal_for(cfg), want_rgb=True)
        rng = np.random.default_rng(77)
        pix = rng.choice(1920 * 1080, 40000, replace=False).astype(np.int32)
        o_argb, o_rgb = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(cfg), pix=pix, nthreads=16)
        assert np.array_equal(argb.ravel()[pix], o_argb)
        assert np.array_equal(rgb.reshape(-1, 4)[pix, :3].view(np.uint32), o_rgb.view(np.uint32))
        # the mirror wall is in view and reflects lit surfaces: a good part of the wall's pixels is not black
        assert (argb != 0xFF000000).mean() > 0.4       # the box fills the central 1080 x 1080 of the frame
    tr.close()
