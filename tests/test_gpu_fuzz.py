"""-m gpu: a slice of tools/fuzz_paths.py inside the suite — random scenes, lights, cameras, sphere tables, sampling
and band partitions; the culled paths (interval cull / tile masks) must equal the unculled ones and the generic
kernel bit for bit.  `python tools/fuzz_paths.py 2000` runs the long version."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(5000, 5032))
def test_fuzz_case(seed):
    from fuzz_paths import one_case
    assert one_case(seed)
