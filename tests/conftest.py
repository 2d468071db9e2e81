import os
import sys

import pytest

# the deployment's setting ("Deployment"), made here because this process IS the deployment of the test run:
# before anything initialises the HIP runtime
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_stress: the randomised parity stress (a subset of -m gpu; GANGFIT_STRESS_SEEDS widens it)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def gf_ctx():
    """One gf_ctx for the whole GPU session.  Fails loudly (no skip, no CPU fallback) when the library is absent."""
    import gangfit

    ctx = gangfit.Context(0)
    yield ctx
    ctx.close()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU would only produce confusing errors: say why instead.
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: GPU tests run on the MI355X box via `pytest -m gpu`")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
