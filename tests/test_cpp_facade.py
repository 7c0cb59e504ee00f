"""The C++ host facade (include/pharmsol_hip.hpp) over the C ABI: a compiled test program that reads like the
reference's own tests (tests/cpp/facade_test.cpp).  `cpu` mode: data model, label rules, population compiler,
loud failure without a device.  `gpu` mode: analytical_readme values, a two-compartment grid against the
oracle at 1e-6, ODE dose conservation."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _build():
    subprocess.run(["make", "-C", ROOT, "-s", "tests/cpp/facade_test"], check=True)


def test_cpp_facade_host_logic():
    _build()
    r = subprocess.run([BIN, "cpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_facade_on_gpu():
    _build()
    r = subprocess.run([BIN, "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu: 0 failure(s)" in r.stdout
