"""-m gpu: per-mesh material and placement of OBJ meshes (the reference hard-codes blue, scale 1.5
and one translation, Loader.cpp:20,42,48-52): a MIRROR mesh and a GLASS mesh in the box, every device path against
the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import focal_for
from uob_raytracer_amd import abi, meshgen, runtime as rt

pytestmark = pytest.mark.gpu


def _scene(scene, tmp_path, lon, lat):
    p1, p2 = str(tmp_path / "a.obj"), str(tmp_path / "b.obj")
    meshgen.write_sphere_obj(p1, lon, lat, radius=0.16, bumps=0.05)
    meshgen.write_sphere_obj(p2, lon, lat, radius=0.12, bumps=0.0)
    mirror = rt.Scene.load_obj(p1, color=(0.9, 0.9, 0.9, 0.0), scale=1.2, translate=(-0.45, 0.95, -0.4))
    glass = rt.Scene.load_obj(p2, color=(0.0, 0.0, 0.0, -1.0), scale=1.7, translate=(0.35, 1.1, -0.6))
    assert (mirror.aos[:, 4, 3] == 0.0).all() and (glass.aos[:, 4, 3] == -1.0).all()
    return scene + mirror + glass


@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_TILE_BINS, abi.RT_FLAG_GENERIC_KERNEL])
@pytest.mark.parametrize("lon,lat,kw", [
    (4, 3, dict(width=160, height=120, shadow_samples=64, aa_x=2, aa_y=2)),                 # 26 + 2*16 = 58: wave kernel
    (12, 8, dict(width=128, height=96, shadow_samples=8, aa_x=2, aa_y=1, max_bounces=6)),    # 26 + 2*168: tiled kernel
    (30, 20, dict(width=96, height=72, shadow_samples=3, aa_x=1, aa_y=1, spheres=())),       # 26 + 2*1140: HBM records + masks
])
def test_mirror_and_glass_meshes_vs_oracle(lon, lat, kw, flags, scene, oracle, tmp_path):
    s = _scene(scene, tmp_path, lon, lat)
    if len(s) <= 64 and flags == abi.RT_FLAG_NO_TILE_BINS:
        flags = abi.RT_FLAG_NO_CULL
    cfg = abi.make_config(flags=flags, **kw)
    v, n, c = s.packed()
    tr = rt.RayTracer(cfg, s)
    for yaw, pitch, cam, light in [(0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]), (-0.25, 0.1, [-0.2, 0.1, -2.8], [0.3, -0.6, -0.3])]:
        rot = rt.rotation_matrix(yaw, pitch)
        argb, rgb = tr.render(rot, cam, light, focal_for(cfg), want_rgb=True)
        o_argb, o_rgb = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(cfg))
        bad = np.argwhere(argb.ravel() != o_argb)
        assert bad.size == 0, "%d pixels differ, first %s" % (len(bad), bad[0])
        assert np.array_equal(rgb[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))
    tr.close()
