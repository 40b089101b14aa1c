"""CPU: the oracle restatement (oracle/rt_oracle.c) against frames THE REFERENCE ITSELF rendered on the MI355X's OpenCL
device (tests/golden/ref_gfx950.npz: Source/kernels.cl built for gfx950 against AMD's own OpenCL builtin library with
the reference's own options, launched through the OpenCL runtime — no stand-in for any builtin).  This is what pins
the oracle to the reference; the tolerance and why it is not zero are stated in tests/refgpu_check.py."""
import numpy as np
import pytest

import refgpu_check as RC
from oracle import pyref
from uob_raytracer_amd import abi

META, FRAMES = RC.load()
CASES = [(name, pi) for name in sorted(META["frames"]) for pi in META["frames"][name]["poses"]]


@pytest.fixture(scope="module")
def packed():
    return {k: pyref.pack_scene(a) for k, a in RC.scenes().items()}


def test_fixture_provenance():
    """The frames were rendered by an OpenCL GPU device of the gfx950 family, by the kernel named `draw`, as the reference launches it."""
    assert META["probe"]["opencl_gpu_devices"] >= 1
    for name, m in META["frames"].items():
        for run in m["runs"].values():
            assert run["device"].startswith("gfx950") and run["W"] == m["config"]["width"] and run["n"] == 26


@pytest.mark.parametrize("name,pi", CASES)
def test_oracle_matches_the_reference_on_its_own_device(name, pi, packed, oracle):
    m = META["frames"][name]
    kw = RC.config_kwargs(m["config"])
    v, n, c = packed[m["scene"]]
    yaw, pitch, cam, light = META["poses"][pi]
    argb, _ = oracle.render(abi.make_config(**kw), v, n, c, pyref.rot_matrix(yaw, pitch), cam, light, RC.focal_for(kw), nthreads=8)
    st = RC.assert_within_tolerance(argb.reshape(kw["height"], kw["width"]), FRAMES["%s_p%d" % (name, pi)],
                                    "%s pose %d" % (name, pi), general_view=(pi == 1))
    print(name, pi, st)


def test_oracle_matches_the_reference_4096_64_samples(packed, oracle):
    """The headline's size and sample count (2x2 AA: the reference cannot express 4x2), windows of the 4096^2 frame."""
    m = META["crops_4096"]
    kw = RC.config_kwargs(m["config"])
    v, n, c = packed["box"]
    yaw, pitch, cam, light = META["poses"][0]
    for i, (x0, y0, w, h) in enumerate(m["crops"]):
        ys, xs = np.mgrid[y0:y0 + h, x0:x0 + w]
        pix = (ys * kw["width"] + xs).astype(np.int32).ravel()
        argb, _ = oracle.render(abi.make_config(**kw), v, n, c, pyref.rot_matrix(yaw, pitch), cam, light, RC.focal_for(kw),
                                pix=pix, nthreads=8)
        st = RC.assert_within_tolerance(argb.reshape(h, w), FRAMES["s64_4096_crop%d" % i], "4096^2 window %d" % i)
        print(i, st)


# ---- box + OBJ mesh: the scene shape of the reference's main() (skeleton.cpp:102-103), rendered by the reference on the GPU ----
MMETA, MFRAMES = RC.load_mesh()
MCASES = [(name, pi) for name in sorted(MMETA["frames"]) for pi in MMETA["frames"][name]["poses"]]


@pytest.mark.parametrize("name,pi", MCASES)
def test_oracle_matches_the_reference_on_box_plus_mesh(name, pi, oracle):
    m = MMETA["frames"][name]
    kw = RC.config_kwargs(m["config"])
    aos = RC.mesh_scene(*m["mesh"])
    assert aos.shape[0] == m["triangles"]
    assert "%016x" % pyref.fnv1a64_words(np.ascontiguousarray(aos).view(np.uint32).ravel()) == m["scene_fnv"]      # the scene the reference rendered
    assert m["runs"]["p%d" % pi]["device"].startswith("gfx950") and m["runs"]["p%d" % pi]["n"] == m["triangles"]
    v, n, c = pyref.pack_scene(aos)
    yaw, pitch, cam, light = MMETA["poses"][pi]
    argb, _ = oracle.render(abi.make_config(**kw), v, n, c, pyref.rot_matrix(yaw, pitch), cam, light, RC.focal_for(kw), nthreads=8)
    st = RC.assert_within_tolerance(argb.reshape(kw["height"], kw["width"]), MFRAMES["%s_p%d" % (name, pi)],
                                    "%s pose %d" % (name, pi), general_view=(pi == 1))
    print(name, pi, st)
