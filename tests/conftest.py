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


@pytest.fixture(scope="session", autouse=True)
def _device_fills_are_ordered_before_renders():
    """torch.zeros(..., device="cuda") is a fill KERNEL on torch's current stream; the library renders on its own
    non-blocking streams (or on a torch.cuda.Stream() a test passes), which nothing orders after that fill.  A render
    that finishes within microseconds -- a frame whose every ray hits in a few steps -- can then be overwritten by
    the fill that was supposed to precede it (seen once on test_wave_kernel_with_every_pixel_live_and_hitting[Box]
    after the round-3 speed-ups).  A real host orders its own writes; the tests do it here, once, for every device
    tensor they create with a fill."""
    try:
        import torch
    except ImportError:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    names = ["zeros", "ones", "full", "zeros_like", "ones_like", "full_like"]
    originals = {n: getattr(torch, n) for n in names}

    def ordered(fn):
        def wrapper(*args, **kwargs):
            t = fn(*args, **kwargs)
            if t.is_cuda:
                torch.cuda.synchronize(t.device)
            return t
        return wrapper
    for n in names:
        setattr(torch, n, ordered(originals[n]))
    yield
    for n in names:
        setattr(torch, n, originals[n])
