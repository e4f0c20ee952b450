"""Build recipe for the in-tree native libraries (hipcc cross-compiles gfx950 without a GPU).

  libbppp_hip.so        the product: HIP kernels + C ABI (include/bppp.h)
  libbppp_hip_test.so   test-only hooks (include/bppp_test.h): device field / group primitives for the parity tests and the
                        multiply-rate microbenchmark bench.py quotes; NOT linked into the product library
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SOURCES = ["msm.hip", "basis.hip", "comb.hip", "fold.hip", "rounds.hip", "nl.hip", "nlb.hip", "nlbatch.hip", "ip.hip", "trrp.hip", "rp.hip", "rpprove.hip", "rpprove_dev.hip", "rpp_transcript.hip", "ipb.hip", "ipb_host.hip", "brpprove.hip", "brpprove_dev.hip", "glv.hip", "capi.hip"]
TEST_SOURCES = ["testhooks.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"] + os.environ.get("BPPP_EXTRA_FLAGS", "").split()


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.join(LIB, "obj"), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    headers.append(os.path.join(HERE, "..", "include", "bppp.h"))
    jobs = []
    for src in HIP_SOURCES + TEST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB, "obj", src.replace(".hip", ".o"))
        if force or _newer(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    out = os.path.join(LIB, "libbppp_hip.so")
    objs = [os.path.join(LIB, "obj", s.replace(".hip", ".o")) for s in HIP_SOURCES]
    if force or jobs or _newer(out, objs + [os.path.join(CSRC, "exports.map")]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    tout = os.path.join(LIB, "libbppp_hip_test.so")
    tobjs = [os.path.join(LIB, "obj", s.replace(".hip", ".o")) for s in TEST_SOURCES]
    if force or jobs or _newer(tout, tobjs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", tout] + tobjs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
