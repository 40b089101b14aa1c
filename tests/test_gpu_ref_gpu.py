"""-m gpu: the HIP path through the C ABI against THE REFERENCE ITSELF on this GPU's OpenCL device.

(1) committed frames (tests/golden/ref_gfx950.npz, rendered by Source/kernels.cl built for gfx950 against AMD's own
    OpenCL builtins with the reference's own options; tests/golden/make_ref_gpu_golden.py);
(2) live: where the box's OpenCL runtime exposes the GPU and oracle/_ref/*.co travelled, the reference kernel is run
    again beside the product on views that are NOT in the fixtures (light positions of update()'s animation).
Tolerance: tests/refgpu_check.py (1 LSB of the 8-bit output = 1e-4 in colour; outliers only on discontinuities)."""
import numpy as np
import pytest

import refgpu_check as RC
from oracle import ref_gpu
from uob_raytracer_amd import abi, runtime as rt

pytestmark = pytest.mark.gpu

META, FRAMES = RC.load()
CASES = [(name, pi) for name in sorted(META["frames"]) for pi in META["frames"][name]["poses"]]


@pytest.fixture(scope="module")
def scenes():
    return {k: rt.Scene(a) for k, a in RC.scenes().items()}


@pytest.mark.parametrize("name,pi", CASES)
def test_hip_matches_the_reference_on_its_own_device(name, pi, scenes):
    m = META["frames"][name]
    kw = RC.config_kwargs(m["config"])
    yaw, pitch, cam, light = META["poses"][pi]
    tr = rt.RayTracer(abi.make_config(**kw), scenes[m["scene"]])
    argb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, RC.focal_for(kw))
    tr.close()
    RC.assert_within_tolerance(argb, FRAMES["%s_p%d" % (name, pi)], "%s pose %d" % (name, pi), general_view=(pi == 1))


def test_hip_matches_the_reference_4096_64_samples(scenes):
    m = META["crops_4096"]
    kw = RC.config_kwargs(m["config"])
    yaw, pitch, cam, light = META["poses"][0]
    tr = rt.RayTracer(abi.make_config(**kw), scenes["box"])
    argb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, RC.focal_for(kw))
    tr.close()
    assert int((argb == 0xFF000000).sum()) - m["run"]["black_pixels"] in range(-64, 65)      # background pixels of the whole frame
    for i, (x0, y0, w, h) in enumerate(m["crops"]):
        RC.assert_within_tolerance(argb[y0:y0 + h, x0:x0 + w], FRAMES["s64_4096_crop%d" % i], "4096^2 window %d" % i)


MMETA, MFRAMES = RC.load_mesh()
MCASES = [(name, pi) for name in sorted(MMETA["frames"]) for pi in MMETA["frames"][name]["poses"]]


@pytest.mark.parametrize("flags", [0, abi.RT_FLAG_NO_TILE_BINS])
@pytest.mark.parametrize("name,pi", MCASES)
def test_hip_mesh_kernel_matches_the_reference_on_box_plus_mesh(name, pi, flags):
    """More than 64 triangles: the LDS-tiled mesh kernel (with and without its tile masks) against the reference's frames of
    the box + OBJ mesh — the scene shape of its main(), skeleton.cpp:102-103."""
    m = MMETA["frames"][name]
    kw = RC.config_kwargs(m["config"])
    sc = rt.Scene(RC.mesh_scene(*m["mesh"]))
    assert len(sc) == m["triangles"]
    yaw, pitch, cam, light = MMETA["poses"][pi]
    tr = rt.RayTracer(abi.make_config(flags=flags, **kw), sc)
    argb = tr.render(rt.rotation_matrix(yaw, pitch), cam, light, RC.focal_for(kw))
    tr.close()
    RC.assert_within_tolerance(argb, MFRAMES["%s_p%d" % (name, pi)], "%s pose %d" % (name, pi), general_view=(pi == 1))


def test_live_reference_kernel_on_box_plus_mesh():
    """The reference kernel run beside the product on a box + mesh scene and views that are in no fixture."""
    if not ref_gpu.have("default256") or not ref_gpu.have("cfg1"):
        pytest.skip("oracle/_ref/*.co or oracle/ref_cl_host not present on this machine")
    if not ref_gpu.gpu_available():
        pytest.skip("the OpenCL runtime of this machine exposes no GPU device")
    for variant, kw, mesh in (("default256", dict(width=256, height=256), (20, 10)),
                              ("cfg1", dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()), (26, 14))):
        sc = rt.Scene(RC.mesh_scene(*mesh))
        v, n, c = sc.packed()
        tr = rt.RayTracer(abi.make_config(**kw), sc)
        for yaw, pitch, cam, light in ((0.1, 0.05, [0.1, 0.0, -3.1], [-0.1512, -0.5, -0.7]), (-0.25, 0.1, [-0.3, 0.2, -3.0], [-0.398, -0.5, -0.7])):
            rot = rt.rotation_matrix(yaw, pitch)
            ref, info = ref_gpu.run(variant, kw["width"], kw["height"], v, n, c, rot, cam, light, RC.focal_for(kw))
            assert info["n"] == len(sc)
            argb = tr.render(rot, cam, light, RC.focal_for(kw))
            RC.assert_within_tolerance(argb, ref.reshape(argb.shape), "%s + mesh live yaw %.2f" % (variant, yaw), general_view=True)
        tr.close()


LIVE = [("default", dict(width=1024, height=1024)), ("cfg2", dict(width=1024, height=1024, shadow_samples=16, spheres=())),
        ("s64_512", dict(width=512, height=512, shadow_samples=64))]


@pytest.mark.parametrize("variant,kw", LIVE)
def test_live_reference_kernel_beside_the_product(variant, kw, scenes):
    if not ref_gpu.have(variant):
        pytest.skip("oracle/_ref/ref_%s_gfx950.co or oracle/ref_cl_host not present on this machine" % variant)
    if not ref_gpu.gpu_available():
        pytest.skip("the OpenCL runtime of this machine exposes no GPU device")
    sc = scenes["box"]
    v, n, c = sc.packed()
    tr = rt.RayTracer(abi.make_config(**kw), sc)
    # the light where update() puts it after 7 and 31 frames of its oscillation, seen from a turned, moved camera
    for yaw, pitch, cam, light in ((0.1, 0.05, [0.1, 0.0, -3.1], [-0.1512, -0.5, -0.7]), (-0.25, 0.1, [-0.3, 0.2, -3.0], [-0.398, -0.5, -0.7])):
        rot = rt.rotation_matrix(yaw, pitch)
        ref, info = ref_gpu.run(variant, kw["width"], kw["height"], v, n, c, rot, cam, light, RC.focal_for(kw))
        argb = tr.render(rot, cam, light, RC.focal_for(kw))
        RC.assert_within_tolerance(argb, ref.reshape(argb.shape), "%s live yaw %.2f" % (variant, yaw), general_view=True)
    tr.close()
