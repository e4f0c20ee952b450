"""Device field / group arithmetic vs the oracle (bit-exact: integer work)."""
import ctypes as C
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd.capi import points_to_array, scalars_to_array, array_to_scalars, array_to_point, load_test_library

pytestmark = pytest.mark.gpu

OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "inv": 4, "neg": 5, "mag8mul": 6, "inv_vartime": 7, "inv_safegcd": 8, "inv_fermat": 9,
       "fr_mag8sqr": 10, "fr_mag16norm": 11, "fr_reduced_chain": 12, "fr_weak16": 13, "fr_is_zero16": 14}
# modulus selector of bppp_test_fe_op: 0 = Fq via the production 10x26 limbs, 1 = Fr (8x32), 2 = Fq via the 8x32 code path,
# 3 = Fr via the production 10x26 limbs (fr26.hip.h)


def _fe_op(gpu, op, mod, a, b):
    A, B = scalars_to_array(a), scalars_to_array(b)
    out = np.zeros_like(A)
    lib = load_test_library()            # the hooks live in libbppp_hip_test.so, not in the product library
    rc = lib.bppp_test_fe_op(gpu.h, OPS[op], mod, A.ctypes.data, B.ctypes.data, len(a), out.ctypes.data)
    assert rc == 0, gpu.lib.bppp_last_error(gpu.h)
    return array_to_scalars(out)


def _edge_values(m):
    return [0, 1, 2, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, 2**255 % m, 2**128, 2**128 - 1, 2**64, 2**32 + 977,
            0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, m - 2**32, m - 977, 3**160 % m, 2**224 - 1]


@pytest.mark.parametrize("mod,m", [(0, O.P), (1, O.N), (2, O.P)])
def test_field_ops_match_python(gpu, mod, m):
    rnd = random.Random(11 + mod)
    edge = _edge_values(m)
    a = edge + [rnd.randrange(m) for _ in range(2000)] + [e for e in edge for _ in edge]
    b = edge[::-1] + [rnd.randrange(m) for _ in range(2000)] + [f for _ in edge for f in edge]
    assert _fe_op(gpu, "add", mod, a, b) == [(x + y) % m for x, y in zip(a, b)]
    assert _fe_op(gpu, "sub", mod, a, b) == [(x - y) % m for x, y in zip(a, b)]
    assert _fe_op(gpu, "mul", mod, a, b) == [(x * y) % m for x, y in zip(a, b)]
    assert _fe_op(gpu, "sqr", mod, a, b) == [(x * x) % m for x in a]
    assert _fe_op(gpu, "neg", mod, a, b) == [(-x) % m for x in a]
    small = a[:300]
    assert _fe_op(gpu, "inv", mod, small, small) == [O.inv_mod(x, m) for x in small]
    if mod:          # the binary-Euclid and the division-step inverses (8 x 32 code path): every edge value and 1000 random ones
        assert _fe_op(gpu, "inv_vartime", mod, a[:1000], a[:1000]) == [O.inv_mod(x, m) for x in a[:1000]]
        assert _fe_op(gpu, "inv_safegcd", mod, a[:1500], a[:1500]) == [O.inv_mod(x, m) for x in a[:1500]]
    else:            # Fq production path: fq_inv is the division-step inverse; the addition chain must agree with it
        assert _fe_op(gpu, "inv", mod, a[:1500], a[:1500]) == [O.inv_mod(x, m) for x in a[:1500]]
        assert _fe_op(gpu, "inv_fermat", mod, small, small) == [O.inv_mod(x, m) for x in small]


def test_fq26_worst_case_magnitudes(gpu):
    """lazy reduction: (8a) * (-7b + 8p') with every limb near its magnitude-8 bound must still be exact"""
    rnd = random.Random(5)
    edge = _edge_values(O.P) + [O.P - 1] * 4 + [2**256 - 2**32 - 978, 2**255, (1 << 256) - 1 - 2**32 - 977 - 1]
    edge = [e % O.P for e in edge]
    a = edge + [rnd.randrange(O.P) for _ in range(3000)] + [e for e in edge for _ in edge]
    b = edge[::-1] + [rnd.randrange(O.P) for _ in range(3000)] + [f for _ in edge for f in edge]
    assert _fe_op(gpu, "mag8mul", 0, a, b) == [(-56 * x * y) % O.P for x, y in zip(a, b)]


def test_reference_constant_3_pow_160(gpu):
    # "3^160" test value of FastPrime/Internal.hs:108-116 (verified in SURVEY.md App. C): square-and-multiply on device
    acc = [1]
    for _ in range(160):
        acc = _fe_op(gpu, "mul", 1, acc, [3])
    assert acc[0] == 3**160 % O.N


def _pt_op(gpu, op, ps, qs):
    A, B = points_to_array(ps), points_to_array(qs)
    out = np.zeros_like(A)
    lib = load_test_library()
    rc = lib.bppp_test_point_op(gpu.h, op, A.ctypes.data, B.ctypes.data, len(ps), out.ctypes.data)
    assert rc == 0
    return [array_to_point(out[i]) for i in range(len(ps))]


def test_group_law_complete(gpu, oracle_lib):
    py = O.PyEC()
    pts = O.hash_points(b"grp", 40)
    G = (O.GX, O.GY)
    ps = pts[:20] + [G, G, None, G, None, pts[0]]
    qs = pts[20:40] + [G, py.neg(G), G, None, None, py.neg(pts[0])]
    want = [py.add(p, q) for p, q in zip(ps, qs)]
    assert _pt_op(gpu, 0, ps, qs) == want          # XYZZ += affine, incl. P=Q, P=-Q, infinities
    assert _pt_op(gpu, 1, ps, qs) == want          # XYZZ += XYZZ
    assert _pt_op(gpu, 2, ps, qs) == [py.add(py.add(p, p), py.add(p, p)) for p in ps]
    # lambda*G = (beta*Gx, Gy): the endomorphism constants of FastSECP256K1.hs:39,53
    assert oracle_lib.mul(O.LAMBDA, G) == (O.BETA * O.GX % O.P, O.GY)


def _edge_fr():
    n = O.N
    e = _edge_values(n) + [n - 1] * 3 + [(n - 1) // 2, (n + 1) // 2, 2**255, 2**256 - 1, 2**256 - n, 2**256 - n - 1, n - (2**256 - n),
                                         2**252 - 1, 2**234 - 1, (1 << 26) - 1, ((1 << 256) - 1) // 3]
    # every 26-bit limb all-ones below the top, and single saturated limbs
    e += [((1 << 26) - 1) << (26 * i) for i in range(9)] + [((1 << 22) - 1) << 234]
    return [x % n for x in e]


def test_fr26_ops_match_python(gpu):
    """Fr in 10 x 26-bit lazy limbs (csrc/fr26.hip.h) — the representation the verifier's scalar kernels (k_trrp_*, k_vb_shared4,
    k_ipvb_proof, k_brp_public) and the prover rounds compute in — against Python integers, edge scalars included."""
    n = O.N
    rnd = random.Random(26)
    edge = _edge_fr()
    a = edge + [rnd.randrange(n) for _ in range(3000)] + [e for e in edge for _ in edge]
    b = edge[::-1] + [rnd.randrange(n) for _ in range(3000)] + [f for _ in edge for f in edge]
    assert _fe_op(gpu, "add", 3, a, b) == [(x + y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "sub", 3, a, b) == [(x - y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "mul", 3, a, b) == [(x * y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "sqr", 3, a, b) == [(x * x) % n for x in a]
    assert _fe_op(gpu, "neg", 3, a, b) == [(-x) % n for x in a]


def test_fr26_worst_case_magnitudes(gpu):
    """every limb at its magnitude-8 (mul, sqr) / magnitude-16 (normalize, weak, is_zero) bound must still be exact"""
    n = O.N
    rnd = random.Random(27)
    edge = _edge_fr()
    a = edge + [rnd.randrange(n) for _ in range(3000)] + [e for e in edge for _ in edge]
    b = edge[::-1] + [rnd.randrange(n) for _ in range(3000)] + [f for _ in edge for f in edge]
    assert _fe_op(gpu, "mag8mul", 3, a, b) == [(-56 * x * y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "fr_mag8sqr", 3, a, b) == [(64 * x * x) % n for x in a]
    assert _fe_op(gpu, "fr_mag16norm", 3, a, b) == [(8 * x - 7 * y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "fr_reduced_chain", 3, a, b) == [(2 * x - y) % n for x, y in zip(a, b)]
    assert _fe_op(gpu, "fr_weak16", 3, a, b) == [(16 * x) % n for x in a]
    # is_zero at magnitude 16: 8a == 7b exactly when b = 8a/7
    inv7 = O.inv_mod(7, n)
    b2 = [(8 * x * inv7) % n if i % 2 == 0 else y for i, (x, y) in enumerate(zip(a, b))]
    assert _fe_op(gpu, "fr_is_zero16", 3, a, b2) == [int((8 * x - 7 * y) % n == 0) for x, y in zip(a, b2)]
