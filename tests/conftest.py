import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest`` run on a machine without a GPU skips the gpu-marked tests instead of failing in them.  With a GPU
    present nothing is skipped: a missing libpdegym_hip.so must fail loudly there (there is no fallback path)."""
    try:
        import torch
        if torch.cuda.is_available():
            return
    except Exception:  # pragma: no cover
        pass
    skip = pytest.mark.skip(reason="gpu test: no HIP device in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Group(dict):
    """One named case of a golden .npz ('<case>/<array>' keys)."""
    __getattr__ = dict.__getitem__


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        case, arr = k.split("/", 1)
        out.setdefault(case, Group())[arr] = z[k]
    return out


@pytest.fixture(scope="session")
def golden_transport():
    return load_golden("transport")


@pytest.fixture(scope="session")
def golden_parabolic():
    return load_golden("parabolic")


@pytest.fixture(scope="session")
def golden_mixed():
    return load_golden("mixed")


@pytest.fixture(scope="session")
def golden_kat():
    return load_golden("kat")


@pytest.fixture(scope="session")
def golden_ns():
    return load_golden("ns2d")


@pytest.fixture(scope="session")
def golden_traffic():
    return load_golden("traffic")


@pytest.fixture(scope="session")
def golden_tumor():
    return load_golden("tumor")
