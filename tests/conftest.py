import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both shared libraries must exist (the driver's build() makes them; build here if a dev forgot)."""
    from pharmsol_amd import _ffi
    import oracle

    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    oracle.build()
    yield


def rel_err(got, want, floor=1e-12):
    import numpy as np

    return np.abs(got - want) / np.maximum(np.abs(want), floor)
