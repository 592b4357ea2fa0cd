"""The build hook of the drop-in boundary: the reference's
top-level CMakeLists does add_subdirectory(modules/OpticalFlow) and links the target
`OpticalFlow` (/root/reference/CMakeLists.txt:10,17), whose PUBLIC include dirs must provide
<flow_opencv.hpp>.  This test builds a stand-in top-level project of the same shape (same two
CMake lines, same compile flags, C++11) around the facade, with the replay harness as the
executable because mainloop.cpp itself needs OpenCV/mavlink headers the image lacks."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACADE = os.path.join(ROOT, "aero-optical-flow_amd", "facade")

TOP = """cmake_minimum_required(VERSION 3.1)
project(aero-optical-flow)
find_package(Threads REQUIRED)
add_subdirectory(modules/OpticalFlow)
add_executable(aero-optical-flow src/replay_mainloop.cpp)
set_property(TARGET aero-optical-flow PROPERTY CXX_STANDARD 11)
set_property(TARGET aero-optical-flow PROPERTY COMPILE_FLAGS "-Wall -Wextra -Wshadow -Wundef -Wformat=2 -Wlogical-op -Wsign-compare -Wformat-security -Wpointer-arith -Winit-self -Wstrict-aliasing=2 -Wuninitialized")
target_link_libraries(aero-optical-flow OpticalFlow)
target_link_libraries(aero-optical-flow pthread)
"""


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_facade_is_a_cmake_target_named_opticalflow(tmp_path, aof):
    proj = tmp_path / "proj"
    (proj / "src").mkdir(parents=True)
    (proj / "modules").mkdir()
    os.symlink(FACADE, proj / "modules" / "OpticalFlow")
    shutil.copy(os.path.join(FACADE, "replay", "replay_mainloop.cpp"), proj / "src" / "replay_mainloop.cpp")
    (proj / "CMakeLists.txt").write_text(TOP)
    build = tmp_path / "build"
    # the reference asks for CMake 3.1, which CMake >= 4 only accepts with this policy floor
    r = subprocess.run(["cmake", "-DCMAKE_POLICY_VERSION_MINIMUM=3.5", "-S", str(proj), "-B", str(build)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", str(build), "-j4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = build / "aero-optical-flow"
    assert exe.exists()
    # the executable resolves its engine libraries without LD_LIBRARY_PATH
    ldd = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libOpticalFlow" in ldd and "not found" not in ldd, ldd
    # without a GPU it must still start, report the failed engine and never publish
    usage = subprocess.run([str(exe)], capture_output=True, text=True)
    assert usage.returncode == 2 and "usage" in usage.stderr
