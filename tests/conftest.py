import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def kifs():
    """The product package; importing it loads libkifs_hip.so (no fallback)."""
    import __graft_entry__ as g
    g.build()  # no-op when libkifs_hip.so matches the recorded source hash
    import kifs_raymarching_amd as K
    return K


@pytest.fixture(scope="session")
def gs(kifs):
    """One GraphicState on cuda:0 shared by the GPU tests (single process, one context)."""
    g = kifs.GraphicState(0)
    yield g
    g.close()
