"""CPU, container only (-m ref is implied by availability): the oracle restatement against the reference
kernel itself (oracle/_ref, built from /root/reference by oracle/build_ref.py) on inputs beyond the
committed fixtures — different views, lights and a modified scene."""
import numpy as np
import pytest

from oracle import pyref
from uob_raytracer_amd import abi

pytestmark = pytest.mark.skipif(not pyref.have_ref("default256"), reason="oracle/_ref not built (needs /root/reference)")

VIEWS = [(0.0, 0.0, [0.0, 0.0, -3.2], [0.4, -0.5, -0.7]), (-0.5, 0.3, [-0.4, -0.2, -2.7], [0.0, 0.3, -0.2]),
         (1.2, 0.0, [1.5, 0.0, -1.5], [0.0, -0.9, 0.0])]
CASES = {"default256": dict(width=256, height=256),
         "cfg2_256": dict(width=256, height=256, shadow_samples=16, spheres=()),
         "cfg3_480": dict(width=480, height=270, max_bounces=5),
         "aa3_256": dict(width=256, height=256, aa_x=3, aa_y=3),
         "cfg1": dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=())}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_reference_kernel(name, scene, oracle):
    kw = CASES[name]
    cfg = abi.make_config(**kw)
    k = pyref.RefKernel(name)
    aos = scene.aos.copy()
    aos[[2, 3], 4, :] = (1.0, 1.0, 1.0, 0.0)      # left wall -> mirror
    aos[[20, 21], 4, :] = (0.0, 0.0, 0.0, -1.0)   # one tall-block face -> glass
    focal = 1100.0 * min(cfg.width, cfg.height) / 1024.0 * cfg.aa_x
    for scene_aos in (scene.aos, aos):
        v, n, c = pyref.pack_scene(scene_aos)
        for yaw, pitch, cam, light in VIEWS:
            rot = pyref.rot_matrix(yaw, pitch)
            a0, t0 = k.render(v, n, c, rot, cam, light, focal)
            a1, r1 = oracle.render(cfg, v, n, c, rot, cam, light, focal)
            assert np.array_equal(a0, a1)
            _, tap = pyref.quantise(r1)
            assert np.array_equal(tap.view(np.uint32), t0.view(np.uint32))


def test_product_scene_equals_reference_scene(scene):
    assert np.array_equal(scene.aos.view(np.uint32), pyref.ref_load_test_model().view(np.uint32))
