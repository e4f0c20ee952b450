"""The host-side SHA-256 (csrc/sha256.hip.h: the CPU's SHA extensions when present, the portable rounds otherwise) against the
FIPS 180-4 "abc" vector and against the portable compression function on 2000 messages of lengths 0 .. 20 000 streamed in random
pieces.  The host hashes the transcripts of small batches (the oracle of <= 8 proofs runs on the host)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sha256_paths_agree(tmp_path):
    exe = str(tmp_path / "sha_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bulletproofspp_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "sha_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("bad 0;"), out
