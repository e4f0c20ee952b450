"""Second opinion on the oracle's group law: OpenSSL libcrypto secp256k1."""
import os
import random
import subprocess

import pytest

import pyoracle as O

EXE = os.path.join(os.path.dirname(O.oracle_lib_path()), "openssl_check")


def _openssl(sgs):
    inp = f"{len(sgs)}\n" + "".join(f"{s:x} {(p or (0, 0))[0]:x} {(p or (0, 0))[1]:x}\n" for s, p in sgs)
    out = subprocess.run([EXE], input=inp, capture_output=True, text=True, check=True).stdout.split()
    x, y = int(out[0], 16), int(out[1], 16)
    return None if x == 0 and y == 0 else (x, y)


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (33, 3), (300, 4)])
def test_inner_product_matches_openssl(oracle_lib, n, seed):
    if not os.path.exists(EXE):
        pytest.skip("openssl_check not built (libcrypto headers missing)")
    rnd = random.Random(seed)
    pts = O.hash_points(b"ossl%d" % seed, n)
    sgs = [(rnd.randrange(O.N), p) for p in pts]
    if n > 4:
        sgs[1] = (0, pts[1])
        sgs[2] = (5, None)
        sgs[3] = (O.N - 1, pts[3])
    assert _openssl(sgs) == oracle_lib.inner_product(sgs)


def test_pair_ip_matches_openssl(oracle_lib):
    if not os.path.exists(EXE):
        pytest.skip("openssl_check not built")
    rnd = random.Random(9)
    g0, g1 = O.hash_points(b"pair", 2)
    for _ in range(5):
        a, b = O.rational_reduce_scalar(rnd.randrange(O.N))
        assert oracle_lib.pair_ip(b, g0, a, g1) == _openssl([(b % O.N, g0), (a % O.N, g1)])
