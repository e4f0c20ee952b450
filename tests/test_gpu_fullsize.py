"""BASELINE.json full-size checks through size-independent properties (the oracle's Straus loop would need ~80 s at
2^20): agreement of different window decompositions, the sharding identity MSM(A ++ B) = MSM(A) + MSM(B) that the
multi-GPU layout relies on, negation symmetry, and bit-exact agreement with the oracle on a 2^14-term prefix."""
import ctypes

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd.capi import points_to_array, scalars_to_array, array_to_point

pytestmark = pytest.mark.gpu
N20 = 1 << 20


@pytest.fixture(scope="module")
def big(gpu):
    rng = np.random.default_rng(0xB9B9)
    sc = rng.integers(0, 2**64, size=(N20, 4), dtype=np.uint64)
    sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))
    m = int(N20 * 2.3)
    xs = rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64)
    dx = gpu.to_device(xs)
    dp = gpu.alloc(m * 64)
    gpu.lift_x(dx, m, dp)
    pts = gpu.download(dp, (m, 8))
    gpu.free(dx); gpu.free(dp)
    pts = np.ascontiguousarray(pts[(pts != 0).any(axis=1)][:N20])
    assert pts.shape[0] == N20
    sc[1] = 0
    pts[2] = 0
    d_sc, d_pts = gpu.to_device(sc), gpu.to_device(pts)
    yield {"sc": sc, "pts": pts, "d_sc": d_sc, "d_pts": d_pts}
    gpu.free(d_sc); gpu.free(d_pts)


def test_msm_2_20_window_decompositions_agree(gpu, big):
    r16 = gpu.msm_device(big["d_sc"], big["d_pts"], N20, 16)
    r13 = gpu.msm_device(big["d_sc"], big["d_pts"], N20, 13)
    r15 = gpu.msm_device(big["d_sc"], big["d_pts"], N20, 15)    # carry window holds one 2^19-entry bucket (heavy-merge path)
    r10 = gpu.msm_device(big["d_sc"], big["d_pts"], N20, 10)
    assert r16 == r13 == r15 == r10 and r16 is not None
    big["full"] = r16


def test_msm_2_20_sharding_identity(gpu, big, oracle_lib):
    """MSM(all) == sum over 8 contiguous shards (bulletproofspp_amd/dist.py layout), added with the oracle's group law."""
    from bulletproofspp_amd.dist import shard_range
    full = big.get("full") or gpu.msm_device(big["d_sc"], big["d_pts"], N20, 0)
    acc = None
    for r in range(8):
        lo, hi = shard_range(N20, r, 8)
        part = gpu.msm_device(big["d_sc"] + lo * 32, big["d_pts"] + lo * 64, hi - lo, 0)
        acc = oracle_lib.add(acc, part)
    assert acc == full


def test_msm_2_16_distinct_points_matches_oracle(gpu, big, oracle_lib):
    """BASELINE config 2 proper: 2^16 DISTINCT points (the first 2^16 of the lifted set) with full-width scalars, bit-exact
    against the oracle's 256-row Straus restatement (~5 s of CPU); through both entry points and a second window width."""
    n = 1 << 16
    u64p = ctypes.POINTER(ctypes.c_uint64)
    sc, pts = np.ascontiguousarray(big["sc"][:n]), np.ascontiguousarray(big["pts"][:n])
    assert len({bytes(r) for r in pts[:, :4]}) >= n - 1          # distinct x coordinates (one row is the infinity encoding)
    want = oracle_lib.inner_product_raw(sc.ctypes.data_as(u64p), pts.ctypes.data_as(u64p), n)
    assert gpu.msm_device(big["d_sc"], big["d_pts"], n, 0) == want
    assert gpu.msm_device(big["d_sc"], big["d_pts"], n, 11) == want
    assert gpu.msm(sc, pts) == want


def test_msm_2_20_negation_and_prefix_oracle(gpu, big, oracle_lib):
    n = 1 << 14
    u64p = ctypes.POINTER(ctypes.c_uint64)
    sc, pts = np.ascontiguousarray(big["sc"][:n]), np.ascontiguousarray(big["pts"][:n])
    want = oracle_lib.inner_product_raw(sc.ctypes.data_as(u64p), pts.ctypes.data_as(u64p), n)
    assert gpu.msm_device(big["d_sc"], big["d_pts"], n, 0) == want
    # negated points give the negated sum: y -> p - y on the whole 2^20 set
    full = big.get("full") or gpu.msm_device(big["d_sc"], big["d_pts"], N20, 0)
    negp = big["pts"].copy()
    P = O.P
    ys = negp[:, 4:]
    # p - y limb-wise with borrow, vectorised (skip the infinity row)
    pl = np.array([(P >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
    borrow = np.zeros(N20, dtype=np.uint64)
    out = np.zeros_like(ys)
    for k in range(4):
        a = np.full(N20, pl[k], dtype=np.uint64)
        b = ys[:, k]
        d = a - b - borrow
        borrow = ((a < b) | ((a == b) & (borrow == 1))).astype(np.uint64)
        out[:, k] = d
    inf = ~(big["pts"] != 0).any(axis=1)
    out[inf] = 0
    negp[:, 4:] = out
    d_neg = gpu.to_device(negp)
    try:
        got = gpu.msm_device(big["d_sc"], d_neg, N20, 0)
    finally:
        gpu.free(d_neg)
    assert got == O.PyEC.neg(full)
