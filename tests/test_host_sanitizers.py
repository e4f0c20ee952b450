"""The host C++ of the library that handles untrusted input or runs threads, under sanitizers (CPU build only):
  * csrc/rpsetup.hpp (range descriptions -> setup, digits, round counts), csrc/hostmath.hpp (256-bit arithmetic, rationalReduceScalar's fast path),
    csrc/sha256.hip.h's host paths: AddressSanitizer + UndefinedBehaviorSanitizer, -fno-sanitize-recover (tests/native/*.cpp);
  * csrc/hostpool.hpp (the verifier's persistent worker pool): ThreadSanitizer.
GPU AddressSanitizer is not available on this pool; the device code is covered by the parity tests."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
INC = os.path.join(ROOT, "bulletproofspp_amd", "csrc")
ASAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
TSAN = ["-fsanitize=thread", "-g", "-O1"]
SHAPES = {"32bit": (11, 6, 3), "64bit": (16, 6, 3), "32by64": (384, 70, 7), "64by64": (512, 261, 8), "96by64": (768, 261, 8), "128by64": (1024, 261, 9), "bin_test": (192, 2, 6),
          "rec_test": (62, 24, 5)}     # (nrmLen, linLen, rounds) of SURVEY.md App. B


def _build(tmp_path, src, flags, name):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-pthread", "-I", INC] + flags + ["-o", exe, os.path.join(NATIVE, src)], check=True)
    return exe


def _schema_lines():
    """the reference's example schemas with the CLI's defaults applied (app/Parse.hs:100-186: count = 1, min = 0, max = 2^64, base = approxLogW)"""
    from bulletproofspp_amd.rangeproof import approx_log_w
    out = []
    ex = os.path.join(ROOT, "tests", "golden", "examples")
    for name in sorted(os.listdir(ex)):
        p = os.path.join(ex, name, "schema.json")
        if not os.path.exists(p):
            continue
        s = json.load(open(p))
        nl, ll, k = SHAPES.get(name, (0, 0, 0))
        flav = 0 if str(s.get("argument", "IP")).lower() in ("nl", "normlinear") else 1
        typed = bool(s.get("typed", False)) or (bool(s.get("conserved", False)) and not s.get("binary", False))
        rs = []
        for r in s["ranges"]:
            lo, hi = int(r.get("min", 0)), int(r.get("max", 2**64))
            base = int(r["base"]) if "base" in r else (2 if s.get("binary", False) else approx_log_w(hi - lo))
            rs += [[str(base), str(lo), str(hi), str(int(bool(r.get("isShared", False)))), str(int(bool(r.get("isOutput", False)))),
                    str(int(bool(r.get("isAssumed", False))))]] * int(r.get("count", 1))
        out.append(" ".join([name, str(int(typed)), str(flav), str(nl), str(ll), str(k), str(len(rs))] + [x for r in rs for x in r]))
    return "\n".join(out) + "\n"


def test_setup_parsing_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "rpsetup_check.cpp", ASAN, "rpsetup_check")
    p = subprocess.run([exe], input=_schema_lines(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok") and "schemas 8," in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("src,expect", [("hostmath_check.cpp", "ok"), ("sha_check.cpp", None), ("rr_check.cpp", None)])
def test_host_math_under_asan_ubsan(tmp_path, src, expect):
    exe = _build(tmp_path, src, ASAN, src[:-4] + "_asan")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stdout[-2000:] + p.stderr[-3000:]
    if expect:
        assert p.stdout.strip().endswith(expect)
    else:
        assert p.stdout.startswith("bad 0;")


def test_host_pool_under_tsan(tmp_path):
    exe = _build(tmp_path, "hostmath_check.cpp", TSAN, "hostmath_tsan")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=1500, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr and p.stdout.strip().endswith("ok"), p.stdout[-2000:] + p.stderr[-3000:]
