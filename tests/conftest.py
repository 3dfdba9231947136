import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def emulator_backend():
    """Install the C-ABI contract emulator (oracle/cabi_emulator.py) for host-logic tests on CPU."""
    from michigan_amd import _cabi
    from oracle.cabi_emulator import EmulatorBackend
    prev = _cabi.set_backend(EmulatorBackend())
    yield
    _cabi.set_backend(prev)


@pytest.fixture
def hip_backend():
    """Make sure the real libmichigan_hip.so backend is active (GPU tests)."""
    from michigan_amd import _cabi
    prev = _cabi.set_backend(None)
    be = _cabi.backend()
    assert be.name == "hip"
    yield be
    _cabi.set_backend(prev)
