"""Host-side helpers (no GPU): the limb-packing conversion of the window combine and the verifier's persistent thread pool, built from
the library's own headers with g++ and run as a native check (tests/native/hostmath_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_from_limbs26_and_host_pool(tmp_path):
    exe = str(tmp_path / "hostmath_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "bulletproofspp_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "hostmath_check.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
