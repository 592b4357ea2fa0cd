"""ASan + UBSan over the product's host-only logic (parameter validation, grids, reduction chunks,
workspace layout, MAVLink packer), 20 000 random parameter sets including invalid ones."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_is_clean_under_asan_ubsan(tmp_path):
    pkg = os.path.join(ROOT, "aero-optical-flow_amd")
    exe = tmp_path / "host_selftest"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"),
           "-I" + os.path.join(pkg, "facade", "include"),
           os.path.join(ROOT, "tests", "native", "host_selftest.cpp"), os.path.join(pkg, "csrc", "aof_params.cpp"),
           os.path.join(pkg, "facade", "src", "optical_flow_rad.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stdout}{r.stderr}"
    assert "valid parameter sets" in r.stdout
