"""Runs the C++ tests of the host side (tests/cpp) on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "test_host_layer")
EXE_GRAPH = os.path.join(HERE, "cpp", "test_graph_capture")


def test_cpp_host_layer_builds():
    """CPU: the header-only C++ layer compiles with plain g++ and links the C ABI."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE) and os.path.exists(EXE_GRAPH)


@pytest.mark.gpu
def test_cpp_host_layer_matches_oracle():
    assert os.path.exists(EXE), "build it first: make -C tests/cpp (or __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_graph_capture_two_graphs_before_first_launch():
    """Raw HIP graph API over the C ABI: two graphs instantiated back to back before either is
    launched, replays equal to the eager results; the no-allocation rule under capture."""
    assert os.path.exists(EXE_GRAPH), "build it first: make -C tests/cpp (or __graft_entry__.build())"
    r = subprocess.run([EXE_GRAPH], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
