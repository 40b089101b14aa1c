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
