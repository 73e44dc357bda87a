"""Runs the C++ host-layer test (tests/cpp/test_host_layer.cpp) on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "test_host_layer")


def test_cpp_host_layer_builds():
    """CPU: the header-only C++ layer compiles with plain g++ and links the C ABI."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_layer_matches_oracle():
    assert os.path.exists(EXE), "build it first: make -C tests/cpp (or __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
