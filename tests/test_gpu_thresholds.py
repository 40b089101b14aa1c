"""-m gpu: (1) lights placed ON the decision thresholds of the exact culls (tools/fuzz_thresholds.py): for a chosen
(surface point, shadow caster) pair the light is moved until one certificate of rt_wave_common.h — det(A1) / det(A2)
centre against hh |p|_1, u+v <= 1 (W0 - hh |w|_1 against slackW), |t d|^2 against radius_sq, a sphere's miss
certificate, the |L|/R = 40 switch (and 73) — sits within a few ulp of its decision value, then stepped by -8..8 (and
+-16, 32, 64) ulp: interval cull on == cull off == generic kernel at every step, and == the CPU ORACLE at the first, the
middle and the last step.  (2) the WIDE domain of rt_init (scene scales 2^-10 and 2^14, translations of 3e4, cameras
1000x farther, lights on planes / vertices / surfaces / far away, spreads beyond the scene) against the CPU ORACLE, not
only against another device path.  The reference comparisons being certified: kernels.cl:266, :272, :302-306."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

from uob_raytracer_amd import abi, runtime as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(36))          # four of each of the nine kinds of threshold
def test_light_on_a_certificate_threshold(seed, oracle):
    from fuzz_thresholds import case
    r = case(seed, oracle=oracle)
    assert r is not False
    if r is None:
        pytest.skip("this draw offered no threshold of its kind")


# seeds of tools/fuzz_paths.py wide_case chosen to cover: scale 2^-10 (0, 2, 23), 2^14 (38, 56, 72), translations of
# 1e4..3e4 (3, 6, 7), camera 1000x farther (10, 16), every kind of light placement (5, 11, 8, 1, 17, 43), a spread 4x the scene (38)
WIDE = [0, 2, 23, 38, 56, 72, 3, 6, 7, 10, 16, 5, 11, 8, 1, 17, 43]


@pytest.mark.parametrize("seed", WIDE)
def test_wide_domain_against_the_oracle(seed, oracle):
    from fuzz_paths import wide_case
    scene, kw, rot, cam, light, focal, info = wide_case(seed)
    f = 64.0 / kw["width"]                                # the same view on a 64x48 frame
    focal = focal * min(64, 48) / min(kw["width"], kw["height"])
    kw = dict(kw, width=64, height=48)
    v, n, c = scene.packed()
    want, want_rgb = oracle.render(abi.make_config(**kw), v, n, c, rot, cam, light, focal, nthreads=8)
    for fl in (0, abi.RT_FLAG_NO_CULL if len(scene) <= 64 else abi.RT_FLAG_NO_TILE_BINS, abi.RT_FLAG_GENERIC_KERNEL):
        tr = rt.RayTracer(abi.make_config(flags=fl, **kw), scene)
        a, tap = tr.render(rot, cam, light, focal, want_rgb=True)
        tr.close()
        bad = np.argwhere(a.ravel() != want)
        assert bad.size == 0, "wide seed %d (%s) flags %d: %d pixels differ from the oracle" % (seed, info, fl, len(bad))
        assert np.array_equal(tap.reshape(-1, 4)[:, :3].view(np.uint32), want_rgb.view(np.uint32)), "float tap differs (%s)" % info
