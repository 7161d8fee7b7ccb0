"""The C++ drop-in shims (include/elimaloc/*.hpp) against the reference's own call lines.

tests/shim_harness/pcm_calls.cpp quotes the lines of pcm_matching.cpp / pcm_matching.hpp that touch Registration,
VoxelHashMap, PointStruct, CovStruct and RegistrationConfig (pcm.cpp:82-105, 122-143, 257-258, 280-282, 308-312, 387,
408-414; pcm.hpp:205-220).  Eigen is absent from the image, so the Eigen-typed form of the shims is compiled against a
test-only <Eigen/Core> stub (tests/fake_eigen) -- with the reference's C++14 -- and the stub-free form (linalg_types.hpp
stand-ins) through examples/pcm_harness.cpp.  The -m gpu tests run both harnesses end to end.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "elimaloc_amd")


def _build(tmp, run, std="c++14"):
    exe = os.path.join(str(tmp), "pcm_calls_run" if run else "pcm_calls")
    cmd = ["g++", "-std=" + std, "-Wall", "-Wextra", "-Werror"] + (["-DELM_RUN"] if run else []) + [
        "-I", os.path.join(ROOT, "tests", "fake_eigen"),
        # the shim directory comes first, so "registration.hpp" / "voxel_hash_map.hpp" resolve to it
        "-I", os.path.join(ROOT, "include", "elimaloc"), "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "shim_harness", "pcm_calls.cpp"), "-L", LIBDIR, "-lelimaloc_hip", "-Wl,-rpath," + LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def _build_plain(tmp, std="c++14"):
    exe = os.path.join(str(tmp), "pcm_harness")
    subprocess.check_call(["g++", "-std=" + std, "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "pcm_harness.cpp"), "-L", LIBDIR, "-lelimaloc_hip", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_reference_call_lines_compile_eigen_typed(tmp_path):
    from elimaloc_amd import _lib
    _lib.lib()  # builds the library on a fresh checkout
    for std in ("c++14", "c++17"):
        assert os.path.exists(_build(tmp_path, run=False, std=std))


def test_shims_compile_without_eigen(tmp_path):
    from elimaloc_amd import _lib
    _lib.lib()
    assert os.path.exists(_build_plain(tmp_path))


@pytest.mark.gpu
def test_reference_call_sequence_runs(tmp_path):
    """Init (map build + covariances + read-backs), CallbackInitialPose (FindGroundHeight -> VoxelDownsample -> RunRegister) and
    CallbackPointCloud (VoxelDownsample -> RunRegister -> in-place TransformPoints) through the Eigen-typed shims on the GPU."""
    r = subprocess.run([_build(tmp_path, run=True)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_plain_shim_harness_runs(tmp_path):
    r = subprocess.run([_build_plain(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_call_sequence_runs_on_a_device_group(tmp_path):
    """the same harness with ELM_DEVICES=0,0: the shims' process-wide context is a device GROUP (two ranks on this GPU, host-memory
    exchange) -- map replicated, every RunRegister sharded -- and pcm_matching.cpp's call lines do not change"""
    exe = _build(tmp_path, run=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ELM_DEVICES="0,0"))
    assert r.returncode == 0, r.stdout + r.stderr
