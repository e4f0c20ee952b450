"""Registered basis with fixed-base precomputation (bppp_basis_*): the same group elements as the arbitrary-point MSM and as the
oracle's Straus restatement, for every window width, prefixes of the basis, batches, zero scalars and infinity points."""
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd.capi import points_to_array, scalars_to_array

pytestmark = pytest.mark.gpu


def _case(n, seed):
    rnd = random.Random(seed)
    pts = O.hash_points(b"basis%d" % seed, n)
    if n > 5:
        pts[3] = None
    return pts, rnd


@pytest.mark.parametrize("c", [0, 2, 4, 7, 9, 10, 13, 16])
def test_basis_msm_equals_oracle_and_plain_msm(gpu, oracle_lib, c):
    n = 200
    pts, rnd = _case(n, 11)
    bas = gpu.basis(points_to_array(pts), window_bits=c)
    assert bas.n == n and (c == 0 or bas.window_bits == c) and bas.table_bytes == (256 // bas.window_bits + 1) * n * 64
    for n_terms, batch in ((n, 1), (n, 6), (57, 3), (1, 2)):
        sc = [[rnd.randrange(O.N) for _ in range(n_terms)] for _ in range(batch)]
        sc[0][0] = 0
        if n_terms > 2:
            sc[-1][2] = O.N - 1
            sc[-1][1] = (O.N + 1) // 2
        d_s = gpu.to_device(np.concatenate([scalars_to_array(r) for r in sc]))
        d_p = gpu.to_device(points_to_array(pts[:n_terms]))
        try:
            got = bas.msm(d_s, n_terms, batch)
            plain = gpu.msm_batch_device(d_s, d_p, n_terms, batch, shared_points=True)
        finally:
            gpu.free(d_s); gpu.free(d_p)
        assert got == plain
        for b in range(batch):
            assert got[b] == oracle_lib.inner_product(list(zip(sc[b], pts[:n_terms])))
    bas.close()


def test_basis_many_small_instances(gpu, oracle_lib):
    """the prover's shape: thousands of instances over one short basis (grouped bucket reduction, one bucket set per instance)"""
    n, batch = 97, 4500
    pts, rnd = _case(n, 12)
    rng = np.random.default_rng(3)
    sc = rng.integers(0, 2**64, size=(batch * n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(1)
    sc[5] = 0
    d_s = gpu.to_device(sc)
    d_p = gpu.to_device(points_to_array(pts))
    bas = gpu.basis(points_to_array(pts), batch_hint=batch)
    try:
        got = bas.msm(d_s, n, batch)
        plain = gpu.msm_batch_device(d_s, d_p, n, batch, shared_points=True)
    finally:
        gpu.free(d_s); gpu.free(d_p)
    assert got == plain
    from bulletproofspp_amd.capi import array_to_scalars
    for b in (0, 1, 2222, batch - 1):
        assert got[b] == oracle_lib.inner_product(list(zip(array_to_scalars(sc[b * n:(b + 1) * n]), pts)))
    bas.close()


@pytest.mark.parametrize("c", [0, 4, 7, 11, 13])
def test_basis_comb_equals_bucket_route_and_oracle(gpu, oracle_lib, c):
    """bppp_basis_enable_comb (csrc/comb.hip: every multiple of every window stored, one mixed addition per non-zero digit): the same
    group elements as the bucket route and the oracle, for several window widths, prefixes of the basis, an infinity point and a
    repeated point in the basis, zero scalars, the scalars around n / 2 where the sign fold flips, sparse vectors (zeros on a
    power-of-two pattern, as the argument's R scalars) and all-zero instances."""
    n, batch = 150, 70
    pts, rnd = _case(n, 21)
    pts[9] = pts[8]                                    # the same point twice
    bas = gpu.basis(points_to_array(pts), batch_hint=batch)
    cc, tb = bas.enable_comb(window_bits=c, budget_bytes=64 << 20)
    W = -(-257 // cc)
    assert (c == 0 or cc == c) and tb == W * n * (1 << (cc - 1)) * 64 and (c != 0 or tb <= (64 << 20))      # the budget picks the width only when none is forced
    for n_terms in (n, 64, 65, 3):
        sc = [[rnd.randrange(O.N) for _ in range(n_terms)] for _ in range(batch)]
        sc[0] = [0] * n_terms                                                        # an empty sum
        sc[1] = [s_ if (i >> 1) & 1 else 0 for i, s_ in enumerate(sc[1])]            # zeros on every left half of level 1
        sc[2] = [s_ if i & 1 else 0 for i, s_ in enumerate(sc[2])]
        sc[3][0], sc[3][1], sc[3][2] = O.N - 1, (O.N + 1) // 2, (O.N - 1) // 2
        sc[4] = [1] * n_terms
        sc[5] = [rnd.randrange(16) for _ in range(n_terms)]                         # digits of a range proof: one non-zero window
        sc[6] = [O.N - 1 - rnd.randrange(16) for _ in range(n_terms)]
        d_s = gpu.to_device(np.concatenate([scalars_to_array(r) for r in sc]))
        d_p = gpu.to_device(points_to_array(pts[:n_terms]))
        try:
            got = bas.msm(d_s, n_terms, batch)
            plain = gpu.msm_batch_device(d_s, d_p, n_terms, batch, shared_points=True)
        finally:
            gpu.free(d_s); gpu.free(d_p)
        assert got == plain and got[0] is None
        for b in (1, 3, 4, 5, 6, batch - 1):
            assert got[b] == oracle_lib.inner_product(list(zip(sc[b], pts[:n_terms])))
    # a budget nothing fits into is refused
    other = gpu.basis(points_to_array(pts))
    with pytest.raises(Exception):
        other.enable_comb(budget_bytes=1 << 10)
    other.close()
    bas.close()


def test_basis_2_14_and_lifetime(gpu, oracle_lib):
    import ctypes
    n = 1 << 14
    rng = np.random.default_rng(14)
    xs = rng.integers(0, 2**64, size=(3 * n, 4), dtype=np.uint64)
    dx, dp = gpu.to_device(xs), gpu.alloc(3 * n * 64)
    gpu.lift_x(dx, 3 * n, dp)
    pts = gpu.download(dp, (3 * n, 8))
    gpu.free(dx); gpu.free(dp)
    pts = np.ascontiguousarray(pts[(pts != 0).any(axis=1)][:n])
    sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))
    d_pts = gpu.to_device(pts)
    bas = gpu.basis(d_pts, device=True, n=n)
    gpu.free(d_pts)                                   # the handle keeps its own table
    d_s = gpu.to_device(sc)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    want = oracle_lib.inner_product_raw(sc.ctypes.data_as(u64p), pts.ctypes.data_as(u64p), n)
    assert bas.msm(d_s, n, 1) == [want]
    gpu.free(d_s)
    bas.close()


def test_comb_many_short_instances(gpu, oracle_lib):
    """Hundreds of instances of a few dozen terms (the inner-product prover's rows: 1 + 6 + 16 = 23 terms) take k_comb_msm_packed — 8 or 16 lanes per
    instance, several instances per wavefront (csrc/comb.hip, round 4): the same points as the bucket route for 3, 23, 24, 25 and 48 terms (the lane
    counts' boundaries), and 49 terms (one wavefront per instance again); zero rows, sparse rows, sign-boundary scalars, an infinity point in the basis."""
    n, batch = 60, 600
    pts, rnd = _case(n, 31)
    bas = gpu.basis(points_to_array(pts), batch_hint=batch)
    bas.enable_comb(window_bits=7, budget_bytes=64 << 20)
    for n_terms in (3, 23, 24, 25, 48, 49):
        sc = [[rnd.randrange(O.N) for _ in range(n_terms)] for _ in range(batch)]
        sc[0] = [0] * n_terms
        sc[1] = [s_ if i & 1 else 0 for i, s_ in enumerate(sc[1])]
        sc[2][0], sc[2][1], sc[2][2] = O.N - 1, (O.N + 1) // 2, (O.N - 1) // 2
        sc[batch - 1] = [1] * n_terms
        sc[batch - 2] = [0] * (n_terms - 1) + [rnd.randrange(O.N)]               # only the last lane's last term
        d_s = gpu.to_device(np.concatenate([scalars_to_array(r) for r in sc]))
        d_p = gpu.to_device(points_to_array(pts[:n_terms]))
        try:
            got = bas.msm(d_s, n_terms, batch)
            plain = gpu.msm_batch_device(d_s, d_p, n_terms, batch, shared_points=True)
        finally:
            gpu.free(d_s); gpu.free(d_p)
        assert got == plain and got[0] is None, "comb differs from the bucket route at %d terms" % n_terms
        for b in (1, 2, 7, batch - 2, batch - 1):
            assert got[b] == oracle_lib.inner_product(list(zip(sc[b], pts[:n_terms])))
    bas.close()
