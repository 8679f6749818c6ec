"""The CMake target `himg` (CMakeLists.txt at the repo root; the reference defines a
target of that name with a PUBLIC include directory, src/lib/CMakeLists.txt:13-26, and
links its three executables against it, src/CMakeLists.txt:27-34): a consumer project
that does what the reference's src/CMakeLists.txt does -- add_subdirectory, then
target_link_libraries(<exe> himg) -- configures, builds tests/cpp/api_roundtrip.cpp and
the executable resolves against the engine's library."""
import os
import shutil
import subprocess

import pytest

import himg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONSUMER = """cmake_minimum_required(VERSION 3.16)
project(consumer CXX)
set(CMAKE_CXX_STANDARD 11)
add_subdirectory({root} himg_amd)
add_executable(api_roundtrip {root}/tests/cpp/api_roundtrip.cpp)
target_link_libraries(api_roundtrip himg)
"""


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_target_himg_builds_a_reference_style_caller(tmp_path):
    himg_amd.lib()   # the in-tree library exists (build.py); the kernels are not compiled twice here
    src, bld = tmp_path / "src", tmp_path / "build"
    src.mkdir()
    (src / "CMakeLists.txt").write_text(CONSUMER.format(root=ROOT))
    lib = os.path.join(ROOT, "himg_amd", "lib", "libhimg_hip.so")
    r = subprocess.run(["cmake", "-S", str(src), "-B", str(bld), "-DHIMG_PREBUILT_LIBRARY=" + lib],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", str(bld)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = bld / "api_roundtrip"
    assert exe.exists()
    # Every himg:: symbol resolves against the library the target names.
    r = subprocess.run(["ldd", str(exe)], capture_output=True, text=True)
    assert "libhimg_hip.so" in r.stdout and "not found" not in r.stdout, r.stdout


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_configures_the_kernel_build(tmp_path):
    """Without a prebuilt library the project configures the hipcc build of every source
    (configure only: compiling the kernels is build.py's / __graft_entry__.build()'s job here)."""
    bld = tmp_path / "build"
    r = subprocess.run(["cmake", "-S", ROOT, "-B", str(bld)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", str(bld), "--target", "help"], capture_output=True, text=True)
    assert r.returncode == 0
    for t in ("himg_library", "chimg", "dhimg", "benchmark"):
        assert t in r.stdout, r.stdout
