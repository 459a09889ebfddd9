import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flope_amd")):      # flope_amd/ on the path exposes the `sunflower` mirror
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def state_dict():
    from flope_amd.weights import synthetic_state_dict
    return synthetic_state_dict(0)


@pytest.fixture(scope="session")
def golden_cfg1():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "posenet_cfg1.npz")))


@pytest.fixture(scope="session")
def ref_fixtures():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "reference_fixtures.npz")))


@pytest.fixture(scope="session")
def harness():
    import ctypes
    path = os.environ.get("FLOPE_HOST_HARNESS") or os.path.join(ROOT, "tests", "host_harness", "libflope_host_harness.so")   # override: the -fsanitize build (make asan)
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "tests/host_harness/libflope_host_harness.so"])
    return ctypes.CDLL(path)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
