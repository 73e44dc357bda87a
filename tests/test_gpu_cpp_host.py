"""Runs the C++ tests of the host side (tests/cpp) on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "test_host_layer")
EXE_GRAPH = os.path.join(HERE, "cpp", "test_graph_capture")
EXE_DROPIN = os.path.join(HERE, "cpp", "test_dropin")
EXE_PIPE = os.path.join(HERE, "cpp", "test_pipeline_shard")


def test_cpp_host_layer_builds():
    """CPU: the header-only C++ layer compiles with plain g++ and links the C ABI."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE) and os.path.exists(EXE_GRAPH) and os.path.exists(EXE_DROPIN) and os.path.exists(EXE_PIPE)


def test_dropin_header_parses_against_the_reference_headers():
    """CPU: include/slam_amd_opencv.hpp with the shims of include/dropin/ forced in front of the
    reference's own frame / map headers (syntax only; skipped where the reference tree is absent)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "dropin_reference_headers.ok"], stdout=subprocess.DEVNULL)
    note = open(os.path.join(HERE, "cpp", "dropin_reference_headers.ok")).read()
    assert "checked against" in note or "skipped" in note


def test_opencv_diff_tool_parses():
    """CPU: tests/cpp/opencv_diff.cpp (the tool that compares the library with OpenCV itself, built only where
    OpenCV is installed) parses and type-checks against the declarations of the test-only stand-in."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "opencv_diff_syntax.ok"], stdout=subprocess.DEVNULL)
    assert "parses" in open(os.path.join(HERE, "cpp", "opencv_diff_syntax.ok")).read()


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


@pytest.mark.gpu
def test_cpp_dropin_opencv_header_call_shapes():
    """The drop-in slam::Detector / slam::Matcher (OpenCV-typed header, compiled against a test-only
    stand-in of the OpenCV types) driven with the reference's call shapes, results vs the oracle."""
    assert os.path.exists(EXE_DROPIN), "build it first: make -C tests/cpp (or __graft_entry__.build())"
    r = subprocess.run([EXE_DROPIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_frame_pipeline_and_pair_shard():
    """slam_amd::FramePipeline (three streams, pinned ring) against the sequential oracle, and
    slam_amd::PairShard (the RCCL pair shard) with a world of one rank against the oracle."""
    assert os.path.exists(EXE_PIPE), "build it first: make -C tests/cpp (or __graft_entry__.build())"
    r = subprocess.run([EXE_PIPE], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout + r.stderr
