"""rational_reduce_scalar (csrc/hostmath.hpp) takes Euclid's steps with word-sized quotient estimates; the plain multi-limb
restatement of rationalReduceScalar (src/Commitment.hs:242-255) is kept beside it.  Both must return the same (r, s) with the same
signs on every input: 200 000 scalars (uniform, short, near n, powers of two), compiled for the host with g++.  The plain version
itself is pinned against the oracle and the golden vectors in test_abi.py / test_golden.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_rational_reduce_equals_plain(tmp_path):
    exe = str(tmp_path / "rr_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bulletproofspp_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "rr_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("bad 0;"), out
