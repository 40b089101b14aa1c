"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from conftest import DEFAULT_CAM, DEFAULT_LIGHT, focal_for
from uob_raytracer_amd import abi, runtime as rt

pytestmark = pytest.mark.gpu

POSES = [
    (0.0, 0.0, DEFAULT_CAM, DEFAULT_LIGHT),
    (0.3, -0.2, [0.2, 0.1, -2.9], [-0.3, -0.5, -0.7]),
]

CASES = {
    "reference_constants_256": dict(width=256, height=256),
    "cfg1_hard_shadow_1spp": dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()),
    "cfg2_diffuse_s16": dict(width=256, height=256, shadow_samples=16, spheres=()),
    "cfg3_bounce5_480x270": dict(width=480, height=270, max_bounces=5),
    "s64_aa4x2_192": dict(width=192, height=192, aa_x=4, aa_y=2, shadow_samples=64),
    "aa3x3": dict(width=128, height=128, aa_x=3, aa_y=3),
    "ragged_100x37": dict(width=100, height=37, shadow_samples=4),
    # 64 shadow samples: the wave-per-surface-point kernel (rt_kernel_wave.hip)
    "wave_s64_aa2x2_160": dict(width=160, height=160, shadow_samples=64),
    "wave_s64_aa1_128_nospheres": dict(width=128, height=128, aa_x=1, aa_y=1, shadow_samples=64, spheres=()),
    "wave_s64_aa8x8_48": dict(width=48, height=48, aa_x=8, aa_y=8, shadow_samples=64),
    "wave_s64_aa4x2_ragged_100x37": dict(width=100, height=37, aa_x=4, aa_y=2, shadow_samples=64),
    "wave_s64_hard_shadow": dict(width=128, height=128, aa_x=2, aa_y=1, shadow_samples=64, light_spread=0.0),
    # more than 64 shadow samples: passes of 64 sample lanes
    "wave_s100_aa2x2": dict(width=96, height=64, shadow_samples=100),
    "wave_s129_aa3x1_spread": dict(width=64, height=48, aa_x=3, aa_y=1, shadow_samples=129, light_spread=0.3),
    "wave_s64_bands": dict(width=128, height=120, aa_x=4, aa_y=2, shadow_samples=64, band_rows=8, band_index=1, band_count=3),
}


# every configuration is rendered by all three device paths: the hierarchical wave kernel (default; falls
# back to the generic kernel by itself where it does not apply, e.g. 3x3 AA), the same kernel testing
# every triangle (RT_FLAG_NO_CULL) and the one-thread-per-pixel kernel (RT_FLAG_GENERIC_KERNEL)
@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_CULL, abi.RT_FLAG_GENERIC_KERNEL])
@pytest.mark.parametrize("name", list(CASES))
def test_frame_bit_exact_vs_oracle(name, flags, scene, oracle):
    cfg = abi.make_config(flags=flags, **CASES[name])
    v, n, c = scene.packed()
    tracer = rt.RayTracer(cfg, scene)
    for yaw, pitch, cam, light in POSES:
        rot = rt.rotation_matrix(yaw, pitch)
        argb, rgb = tracer.render(rot, cam, light, focal_for(cfg), want_rgb=True)
        o_argb, o_rgb = oracle.render(cfg, v, n, c, rot, cam, light, focal_for(cfg))
        o_argb = o_argb.reshape(argb.shape)
        bad = np.argwhere(argb != o_argb)
        assert bad.size == 0, "%d pixels differ, first %s: %08x vs %08x" % (
            len(bad), bad[0], argb[tuple(bad[0])], o_argb[tuple(bad[0])])
        # float tap: bit-exact (tolerance 0; the north star allows 1e-4)
        assert np.array_equal(rgb[..., :3].reshape(-1, 3).view(np.uint32), o_rgb.view(np.uint32))
    tracer.close()
