"""The C ABI from plain C99: examples/c_client/rp_roundtrip.c is compiled with gcc against include/bppp.h and libbppp_hip.so (no
Python, no HIP headers on the client side) and run on the GPU: an MSM identity, then setup -> prove -> verify -> tamper -> reject with
the culprit identified, for both argument flavours, through host buffers only — what a binding in the reference's own language
 would do."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_round_trip(tmp_path, gpu):
    lib = os.path.join(ROOT, "bulletproofspp_amd", "lib")
    exe = str(tmp_path / "rp_roundtrip")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_client", "rp_roundtrip.c"),
                    "-L", lib, "-lbppp_hip", "-Wl,-rpath," + lib, "-o", exe], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "c client ok" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
    assert p.stdout.count("tampering identified") == 2
