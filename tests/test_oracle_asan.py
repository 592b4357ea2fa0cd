"""AddressSanitizer + UBSan run of the CPU oracle over random geometries and option mixes
(GPU sanitizers are not available on the pool; the host-side checker is where an
out-of-bounds restatement would otherwise go unnoticed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", d, "-s", "asan"], check=True, timeout=300)
    r = subprocess.run([os.path.join(d, "selftest_asan")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
