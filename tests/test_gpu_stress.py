"""The randomised parity stress (tests/stress_lib.py) as part of `pytest -m gpu`: a bounded number of seeds, so that its
evidence is in the driver's record and not only in a tool's printout.  GANGFIT_STRESS_SEEDS widens it; `-m gpu_stress`
selects it alone."""
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.gpu_stress]

SEEDS = int(os.environ.get("GANGFIT_STRESS_SEEDS", "100"))
CHUNK = 20


@pytest.fixture(scope="module")
def stress_ctxs():
    import stress_lib

    ctxs = stress_lib.make_contexts()
    yield ctxs
    for c in ctxs.values():
        c.close()


@pytest.mark.parametrize("first", range(1, SEEDS + 1, CHUNK))
def test_stress_seeds(stress_ctxs, first):
    import stress_lib

    cases = 0
    for seed in range(first, min(first + CHUNK, SEEDS + 1)):
        c, bad = stress_lib.one_seed(stress_ctxs, 10_000 + seed)
        cases += c
        assert bad is None, bad
    assert cases >= CHUNK
