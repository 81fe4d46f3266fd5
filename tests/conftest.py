import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raptor_amd import _capi  # noqa: E402
from raptor_amd.urdf import load_urdf  # noqa: E402

ROBOTS = os.path.join(ROOT, "tests", "robots")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), bound through the same ctypes class as the product."""
    if not os.path.exists(ORACLE_LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    return _capi.CApi(ORACLE_LIB, "oracle_")


@pytest.fixture(scope="session")
def hip():
    """The product library on cuda:0.  Fails (not skips) when the extension is missing."""
    import raptor_amd

    api = raptor_amd.api()
    api.init(0)
    return api


@pytest.fixture(scope="session")
def kinova():
    m = load_urdf(os.path.join(ROBOTS, "kinova.urdf"))
    m.gravity = np.array([0.0, 0.0, -9.81])  # KinovaConstants.h:29 GRAVITY
    return m


@pytest.fixture(scope="session")
def kinova_friction(kinova):
    """Same arm with the optional joint terms switched on (armature / damping / friction paths)."""
    m = kinova.copy()
    rng = np.random.default_rng(7)
    m.armature = rng.uniform(0.05, 0.3, m.nv)
    m.damping = rng.uniform(0.1, 1.0, m.nv)
    m.friction = rng.uniform(0.1, 1.0, m.nv)
    return m
