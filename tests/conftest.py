import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (container only)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are skipped (not failed) on a machine without a HIP device."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device present")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the product library and the oracle are built (no-op when up to date)."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "uob_raytracer_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


@pytest.fixture(scope="session")
def scene():
    from uob_raytracer_amd import runtime as rt
    return rt.Scene.cornell_box()


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyref
    return pyref.Oracle()


# the default view of the reference (skeleton.cpp:61-67)
DEFAULT_CAM = [0.0, 0.0, -3.2]
DEFAULT_LIGHT = [0.0, -0.5, -0.7]


def focal_for(cfg):
    """focal_length 2200 at 1024 wide, 2x2 AA (skeleton.cpp:61), rescaled: units are AA sub-pixels along x."""
    return 1100.0 * min(cfg.width, cfg.height) / 1024.0 * cfg.aa_x
