import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orb_oracle
    orb_oracle.lib()
    return orb_oracle


@pytest.fixture(scope="session")
def capi():
    # PyTorch-ROCm ships its own HIP/HSA runtime with the same soname as /opt/rocm's.
    # Whichever is loaded first serves the whole process, and torch cannot see the
    # GPU once the system runtime is already in: load torch first.
    import torch  # noqa: F401
    from slam_amd import capi as c
    c.lib()
    return c
