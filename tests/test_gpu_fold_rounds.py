"""Round kernels (point fold, scalar folds, round sums, tensor) vs the oracle's restatement of
NormArgument.hs / Bulletproof.hs."""
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd.capi import points_to_array, scalars_to_array, array_to_point, array_to_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 8, 65, 387])
def test_fold_points_matches_pair_ip(gpu, oracle_lib, n):
    rnd = random.Random(n)
    pts = O.hash_points(b"fold%d" % n, n)
    if n > 4:
        pts[3] = None
    e = rnd.randrange(O.N)
    a1, b1 = O.rational_reduce_scalar(e)
    assert gpu.rational_reduce(e) == (a1, b1)
    got = gpu.fold_points(b1, a1, points_to_array(pts))
    want = [oracle_lib.pair_ip(b1, pts[2 * j], a1, pts[2 * j + 1] if 2 * j + 1 < n else None) for j in range((n + 1) // 2)]
    assert [array_to_point(got[j]) for j in range(len(want))] == want


@pytest.mark.parametrize("b,a", [(1, 1), (1, -1), (0, 5), (7, 0), (0, 0), (-3, 2**128 + 12345), (2**129 - 1, -(2**129 - 1)),
                                 (0x155555555555555555555555555555555, 0xAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA), (-(2**128), 2**64 - 1)])
def test_fold_points_edge_pairs(gpu, oracle_lib, b, a):
    """Pairs the joint-sparse-form table must get right by the group law: GR = GL (sum entry is a doubling, difference is
    infinity), GR = -GL (the reverse), either or both members infinity, plus ordinary pairs; scalar edge cases."""
    ec = O.PyEC()
    h = O.hash_points(b"edge", 6)
    neg = lambda p: (p[0], O.P - p[1])
    pts = [h[0], h[0], h[1], neg(h[1]), None, h[2], h[3], None, None, None, h[4], h[5], h[5], h[5]]
    got = gpu.fold_points(b, a, points_to_array(pts))
    want = [oracle_lib.pair_ip(b, pts[2 * j], a, pts[2 * j + 1]) for j in range(len(pts) // 2)]
    assert [array_to_point(got[j]) for j in range(len(want))] == want


def test_rational_reduce_edges(gpu):
    rnd = random.Random(3)
    for x in [0, 1, 2, O.N - 1, O.N - 2, 2**128, 2**129, (O.N - 1) // 2, (O.N + 1) // 2] + [rnd.randrange(O.N) for _ in range(300)]:
        assert gpu.rational_reduce(x) == O.rational_reduce_scalar(x)


@pytest.mark.parametrize("n", [1, 2, 5, 512, 261, 1000])
def test_round_scalar_kernels(gpu, n):
    rnd = random.Random(100 + n)
    x = [rnd.randrange(O.N) for _ in range(n)]
    c = [rnd.randrange(O.N) for _ in range(n)]
    q = rnd.randrange(1, O.N)
    qi = O.inv_mod(q, O.N)
    dx, dc = gpu.to_device(scalars_to_array(x)), gpu.to_device(scalars_to_array(c))
    npair = (n + 1) // 2
    dxw, drw, dout = gpu.alloc(2 * npair * 32), gpu.alloc(npair * 32), gpu.alloc(npair * 32)
    try:
        # Norm.makeScalarsComs (NormArgument.hs:113-118)
        st = O.Norm(q, qi, 1, [(v, None) for v in x])
        sX, xw, sR, rw = st.make_scalars_coms()
        q4 = pow(q, 4, O.N)
        sx, sr = gpu.norm_round_sums(dx, n, q4)
        assert 2 * pow(q, 3, O.N) * sx % O.N == sX and q4 * sr % O.N == sR
        gpu.norm_round_openings(dx, n, q, qi, dxw, drw)
        assert array_to_scalars(gpu.download(dxw, (2 * npair, 4))) == [v for v, _ in xw.body]
        assert array_to_scalars(gpu.download(drw, (npair, 4))) == [v for v, _ in rw.body]
        # Linear.makeScalarsComs (NormArgument.hs:56-59)
        lt = O.Linear(1, [(cc, v, None) for cc, v in zip(c, x)])
        lX, lxw, lR, lrw = lt.make_scalars_coms()
        assert gpu.lin_round_sums(dc, dx, n) == (lX, lR)
        gpu.lin_round_openings(dx, n, dxw, drw)
        assert array_to_scalars(gpu.download(dxw, (2 * npair, 4))) == [v for _, v, _ in lxw.body]
        assert array_to_scalars(gpu.download(drw, (npair, 4))) == [v for _, v, _ in lrw.body]
        # collapse scalar parts (NormArgument.hs:129, :71)
        u, v = rnd.randrange(O.N), rnd.randrange(O.N)
        gpu.fold_scalars(u, v, dx, n, dout)
        want = [(u * x[2 * j] + (v * x[2 * j + 1] if 2 * j + 1 < n else 0)) % O.N for j in range(npair)]
        assert array_to_scalars(gpu.download(dout, (npair, 4))) == want
    finally:
        for p in (dx, dc, dxw, drw, dout):
            gpu.free(p)


@pytest.mark.parametrize("nb,k", [(1, 0), (2, 1), (3, 4), (2, 8), (1, 9)])
def test_tensor_matches_list_instance(gpu, nb, k):
    rnd = random.Random(nb * 31 + k)
    bs = [rnd.randrange(O.N) for _ in range(nb)]
    es = [rnd.randrange(O.N) for _ in range(k)]       # last round first
    q = rnd.randrange(1, O.N)
    qs = [pow(q, 2**r, O.N) for r in range(k)]        # iterate (^2) q
    want = O.tensor(bs, es, lambda r: qs[r])
    dout = gpu.alloc(max(1, nb << k) * 32)
    try:
        gpu.tensor(bs, es, qs, dout)
        assert array_to_scalars(gpu.download(dout, (nb << k, 4))) == want
    finally:
        gpu.free(dout)


@pytest.mark.parametrize("nb,k", [(2, 8), (3, 9), (1, 6), (4, 2)])
def test_tensor_is_the_vector_instance(gpu, nb, k):
    """The device kernel indexes bits exactly as the Data.Vector instance of tensor' does (src/Bulletproof.hs:114-122:
    V.generate, multIndex n = bs ! (n div 2^k) * product [testBit n j ? e : q]) — one output element per lane, no list recursion —
    so the V.Vector semantics of SURVEY.md row a15 / f4 are what runs on the GPU; the list instance gives the same vector."""
    rnd = random.Random(nb * 131 + k)
    bs = [rnd.randrange(O.N) for _ in range(nb)]
    es = [rnd.randrange(O.N) for _ in range(k)]
    qs = [rnd.randrange(O.N) for _ in range(k)]
    want = O.tensor_vector(bs, es, qs)
    assert want == O.tensor(bs, es, lambda r: qs[r])
    dout = gpu.alloc(max(1, nb << k) * 32)
    try:
        gpu.tensor(bs, es, qs, dout)
        assert array_to_scalars(gpu.download(dout, (nb << k, 4))) == want
    finally:
        gpu.free(dout)


@pytest.mark.parametrize("mod,m", [(0, O.P), (1, O.N)])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000])
def test_batch_inverse_matches_reference_semantics(gpu, mod, m, n):
    """batchInverse (src/Data/Field/BatchInverse.hs:14-24): inverse of every element, zero mapped to zero"""
    rnd = random.Random(n * 3 + mod)
    xs = [rnd.randrange(m) for _ in range(n)]
    for i in range(0, n, 5):
        xs[i] = 0
    if n > 3:
        xs[3] = m - 1
    dx = gpu.to_device(scalars_to_array(xs))
    try:
        gpu.batch_inverse(dx, n, mod, dx)                     # in place
        got = array_to_scalars(gpu.download(dx, (n, 4)))
    finally:
        gpu.free(dx)
    assert got == O.batch_inverse(xs, m)
