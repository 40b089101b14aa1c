"""-m gpu: a slice of tools/fuzz_paths.py inside the suite — random scenes, lights, cameras, sphere tables, sampling
and band partitions; the culled paths (interval cull / tile masks) must equal the unculled ones and the generic
kernel bit for bit.  `python tools/fuzz_paths.py 2000` runs the long version, `--wide` the one over the whole
domain rt_init accepts (scene scales 2^-10 .. 2^14, translations to 3e4, far cameras, lights on planes / vertices /
surfaces, spreads up to the scene size, slivers)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(5000, 5032))
def test_fuzz_case(seed):
    from fuzz_paths import one_case
    assert one_case(seed)


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_case_wide_domain(seed):
    from fuzz_paths import one_case_wide
    assert one_case_wide(seed)


@pytest.mark.parametrize("seed", [1519])
def test_fuzz_wide_regressions(seed):
    """1519: a camera 50 000 units from a scene of size 32 looks at a diffuse sphere inside a mesh scene.  The reference's
    discriminant b*b - 4*a*c is then rounded by more than its own value, its "hit points" lie tens of units off the
    sphere — and the world cells they fall into must still carry their tile masks (rt_bin_occupancy's sphere slack)."""
    from fuzz_paths import one_case_wide
    assert one_case_wide(seed)
