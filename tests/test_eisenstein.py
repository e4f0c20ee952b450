"""The Eisenstein (FastPrime) variant of rationalReduceScalar and the 65-row pair fold.

CPU part: the oracle's restatement (oracle/pyoracle.py: rational_reduce_scalar_eis, pair_ip_eis) is pinned by what can be pinned
offline — the reference's constants (charEis of Fr has norm n; conj charEis recomposes to 0 mod n, which is why reducedChar
conjugates, src/Commitment.hs:296-297), the defining invariant x = a / b in Fr with (normEis a)^2 <= 2n and components under
rationalReducedScalarLength = 65 bits (:304), and the group identity of the fold against plain scalar multiplications — and the
library's host entry point bppp_rational_reduce_eis must return the same (a, b).  GPU part: bppp_fold_points_eis_device equals the
oracle's fold bit for bit on edge and random scalars."""
import ctypes as C
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd import capi

EDGE = [0, 1, 2, O.N - 1, O.N - 2, (O.N + 1) // 2, (O.N - 1) // 2, 2**128, 2**128 - 1, 2**64, 2**255 % O.N, O.LAMBDA, O.N - O.LAMBDA, O.LAMBDA + 1,
        (O.LAMBDA * O.LAMBDA) % O.N, 3**160 % O.N, 0xFFFFFFFF, 1238349833]


def _cases(n, seed):
    rnd = random.Random(seed)
    return EDGE + [rnd.randrange(O.N) for _ in range(n)]


def test_eisenstein_constants():
    c = O.CHAR_EIS_FR
    assert O.eis_norm(c) == O.N                                        # FastSECP256K1.hs:56
    assert O.eis_recompose(O.eis_conj(c)) == 0                         # conj charEis is 0 mod n (Commitment.hs:296-297) ...
    assert O.eis_recompose(c) != 0                                     # ... the un-conjugated factor is not
    assert pow(O.LAMBDA, 3, O.N) == 1 and O.LAMBDA != 1
    assert O.eis_mul((2, 3), (5, -7)) == (2 * 5 - 3 * -7, 2 * -7 + 3 * 5 - 3 * -7)      # (a0 + b0 w)(a1 + b1 w), w^2 = -1 - w
    # the rounding rule of Eis.hs:80-82: nearest integer, an exact tie stays at the floor (m - |r| < |r| is strict)
    assert O.eis_quot((7, 0), (2, 0)) == (3, 0) and O.eis_quot((-7, 0), (2, 0)) == (-4, 0)
    assert O.eis_quot((9, 0), (4, 0)) == (2, 0) and O.eis_quot((11, 0), (4, 0)) == (3, 0) and O.eis_quot((-11, 0), (4, 0)) == (-3, 0)


def test_rational_reduce_eis_invariants_and_host_entry_point():
    lib = capi.load_library()
    for x in _cases(400, 7):
        a, b = O.rational_reduce_scalar_eis(x)
        assert O.eis_norm(a) ** 2 <= 2 * O.N
        assert O.eis_recompose(a) == O.eis_recompose(b) * x % O.N      # x = a / b
        assert all(abs(v) < 2**65 for v in a + b)                      # rationalReducedScalarLength = 65
        if x:
            assert O.eis_recompose(b) != 0
        am, bm = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        an, bn = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
        xs = capi.int_to_limbs(x)
        assert lib.bppp_rational_reduce_eis(xs.ctypes.data, am.ctypes.data, an.ctypes.data, bm.ctypes.data, bn.ctypes.data) == 0
        comp = lambda m, s, k: (-1 if s[k] else 1) * (int(m[2 * k]) | (int(m[2 * k + 1]) << 64))
        assert ((comp(am, an, 0), comp(am, an, 1)), (comp(bm, bn, 0), comp(bm, bn, 1))) == (a, b), hex(x)
    bad = capi.int_to_limbs(O.N)
    am = np.zeros(4, dtype=np.uint64); an = np.zeros(2, dtype=np.int32)
    assert lib.bppp_rational_reduce_eis(bad.ctypes.data, am.ctypes.data, an.ctypes.data, am.ctypes.data, an.ctypes.data) == -1


def test_pair_ip_eis_is_the_group_element(oracle_lib):
    ec = oracle_lib
    pts = O.hash_points(b"eis fold", 6)
    for i, x in enumerate(_cases(12, 8)):
        a, b = O.rational_reduce_scalar_eis(x)
        gl, gr = pts[i % 5], pts[(i % 5) + 1]
        want = ec.add(ec.mul(O.eis_recompose(b), gl), ec.mul(O.eis_recompose(a), gr))
        assert O.pair_ip_eis(b, gl, a, gr, ec) == want
    assert O.pair_ip_eis((3, -2), None, (5, 7), pts[0], ec) == ec.mul((5 + 7 * O.LAMBDA) % O.N, pts[0])


@pytest.mark.gpu
def test_gpu_fold_points_eis_matches_oracle(gpu, oracle_lib):
    from bulletproofspp_amd.capi import array_to_point, points_to_array
    ec = oracle_lib
    pts = O.hash_points(b"eis gpu", 21)
    pts[4] = None
    for x in _cases(6, 9):
        a, b = gpu.rational_reduce_eis(x)
        assert (a, b) == O.rational_reduce_scalar_eis(x)
        out = gpu.fold_points_eis(b, a, points_to_array(pts))
        for j in range(11):
            gl, gr = pts[2 * j], pts[2 * j + 1] if 2 * j + 1 < len(pts) else None
            assert array_to_point(out[j]) == O.pair_ip_eis(b, gl, a, gr, ec), (hex(x), j)
    # degenerate pairs: GR = +-GL, GR = lambda GL (the two halves then collide inside the final addition)
    g = pts[0]
    special = [g, g, g, ec.neg(g), g, O.cm_mul(g), g, ec.neg(O.cm_mul(g))]
    a, b = gpu.rational_reduce_eis(EDGE[5])
    out = gpu.fold_points_eis(b, a, points_to_array(special))
    for j in range(4):
        assert array_to_point(out[j]) == O.pair_ip_eis(b, special[2 * j], a, special[2 * j + 1], ec)
    out = gpu.fold_points_eis((1, -1), (-1, 1), points_to_array([g, g]))      # b' g + a' g = 0
    assert array_to_point(out[0]) is None
