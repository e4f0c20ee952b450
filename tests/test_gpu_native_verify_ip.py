"""The inner-product flavour (src/Bulletproof/InnerProductArgument.hs — the CLI's default argument, app/Parse.hs:100) at batch scale:
bppp_ip_verify_batch_device and the end-to-end verifier bppp_rp_verify_batch over setups of flavour 1.

The reference's three inner-product examples (examples/32bit, 64bit — the paper's 416-byte proof — and rec_test: typed, three
range kinds) give the shapes; proofs are made by the host protocol code over the per-proof device argument (bppp_ip_*), every one is
also verified by the ORACLE backend (oracle/pyoracle.py's restatement of verifyBPM with makeNorm's basis change done point by point as
InnerProductArgument.hs:194-206 does), and then the batch goes to the library as FILES.  Checked: device-derived challenges equal
RP.verifier_challenges; the combined point of bppp_ip_verify_batch_device equals sum_b rho_b * (the point bppp_ip_verify leaves for
proof b) for valid AND invalid proofs (so the basis change folded into the shared-basis scalars is the per-pair one); accept / reject
/ identify on tampered members."""
import copy
import hashlib
import json
import os
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd import encoding as E
from bulletproofspp_amd import rangeproof as RP
from bulletproofspp_amd.capi import _ptr, array_to_point, int_to_limbs, points_to_array, scalars_to_array
from rp_backends import OracleBackend
from test_rangeproof import EXAMPLES, EXAMPLE_SHAPES

pytestmark = pytest.mark.gpu


def _example(gpu, name, nproofs):
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    assert (st.flavour, st.nrm_len, st.lin_len, st.rounds, st.final_lens) == EXAMPLE_SHAPES[name] and st.flavour == "IP"
    wit_json = json.load(open(os.path.join(EXAMPLES, name, "witness.json")))
    proofs = []
    for j in range(nproofs):                      # the example's amounts under different blindings and prover randomness
        inputs = RP.inputs_from_witness(wit_json, b"ip batch %d" % j)
        proofs.append(RP.prove(st, RP.witness(st, inputs), RP.sha256_oracle(), RP.hash_to_scalar(b"ip rand %d" % j)))
    return schema, st, proofs


def _ip_point(gpu, st, v):
    """the point bppp_ip_verify leaves for one proof (infinity iff it verifies)"""
    out = np.zeros(8, dtype=np.uint64)
    flat = [p for xr in v["responses"] for p in xr]
    rc = gpu.lib.bppp_ip_verify(gpu.h, _ptr(int_to_limbs(v["q"])), _ptr(int_to_limbs(v["sp"])), _ptr(points_to_array([st.g])), _ptr(scalars_to_array(v["pub_norm"])),
                                _ptr(points_to_array(st.gs)), len(st.gs), _ptr(scalars_to_array(v["pub_lin_c"])), _ptr(scalars_to_array(v["pub_lin_x"])),
                                _ptr(points_to_array(st.hs)), len(st.hs), _ptr(scalars_to_array(v["es"])), len(v["es"]), _ptr(scalars_to_array(v["wit_norm"])),
                                len(v["wit_norm"]), _ptr(scalars_to_array(v["wit_lin"])), len(v["wit_lin"]), _ptr(scalars_to_array([s for s, _ in v["init_terms"]])),
                                _ptr(points_to_array([p for _, p in v["init_terms"]])), len(v["init_terms"]), _ptr(points_to_array(flat)), _ptr(out))
    gpu._check(rc, "bppp_ip_verify")
    return array_to_point(out)


def _ip_batch_point(gpu, st, vs, rhos):
    B, k, (fn, fl), ninit = len(vs), st.rounds, st.final_lens, len(vs[0]["init_terms"])
    cat = lambda rows: np.concatenate([scalars_to_array([x % RP.N for x in r]) for r in rows])
    up = gpu.to_device
    d = [up(points_to_array([st.g])), up(points_to_array(st.gs)), up(points_to_array(st.hs)), up(scalars_to_array(rhos)), up(scalars_to_array([v["q"] for v in vs])),
         up(scalars_to_array([v["sp"] for v in vs])), up(cat([v["pub_norm"] for v in vs])), up(cat([v["pub_lin_c"] for v in vs])), up(cat([v["pub_lin_x"] for v in vs])),
         up(cat([v["es"] for v in vs])), up(cat([v["wit_norm"] for v in vs])), up(cat([v["wit_lin"] for v in vs])), up(cat([[s for s, _ in v["init_terms"]] for v in vs])),
         up(np.concatenate([points_to_array([p for _, p in v["init_terms"]]) for v in vs])),
         up(np.concatenate([points_to_array([p for xr in v["responses"] for p in xr]) for v in vs]))]
    out = np.zeros(8, dtype=np.uint64)
    try:
        rc = gpu.lib.bppp_ip_verify_batch_device(gpu.h, B, len(st.gs), len(st.hs), k, fn, fl, ninit, *[_ptr(x) for x in d], _ptr(out))
        gpu._check(rc, "bppp_ip_verify_batch_device")
    finally:
        for x in d:
            gpu.free(x)
    return array_to_point(out)


@pytest.mark.parametrize("name", ["64bit", "32bit", "rec_test"])
def test_ip_batch_point_is_the_weighted_sum_of_the_single_proof_points(gpu, oracle_lib, name):
    _, st, proofs = _example(gpu, name, 5)
    ec = oracle_lib
    vs = [RP.verify_inputs(st, p, RP.sha256_oracle()) for p in proofs]
    ob = OracleBackend(oracle_lib)
    for v in vs[:2]:                              # the reference's verifier on the CPU: transformed basis, one commit
        assert ob.verify_bp("IP", v["q"], v["sp"], st.g, v["pub_norm"], st.gs, v["pub_lin_c"], v["pub_lin_x"], st.hs, v["es"], v["responses"], v["wit_norm"],
                            v["wit_lin"], v["init_terms"])
    rnd = random.Random(name)
    rhos = [rnd.randrange(1, RP.N) for _ in vs]
    assert all(_ip_point(gpu, st, v) is None for v in vs)
    assert _ip_batch_point(gpu, st, vs, rhos) is None
    # invalid members: the combination must be EXACTLY sum rho_b * E_b with E_b the single-proof point (oracle arithmetic for the sum)
    bad = copy.deepcopy(vs)
    bad[1]["wit_norm"][0] = (bad[1]["wit_norm"][0] + 5) % RP.N             # enters through vx, vy and the tensor on g0 / g1
    bad[2]["wit_lin"][0] = (bad[2]["wit_lin"][0] + 7) % RP.N
    bad[3]["es"][0] = (bad[3]["es"][0] + 1) % RP.N                         # a challenge: its inverse, the tensors, the response scalars
    bad[4]["q"] = (bad[4]["q"] + 1) % RP.N                                 # r itself: q = r^4 and the factor on every g0
    singles = [_ip_point(gpu, st, v) for v in bad]
    assert singles[0] is None and all(s is not None for s in singles[1:])
    want = ec.inner_product(list(zip(rhos, singles)))
    assert _ip_batch_point(gpu, st, bad, rhos) == want
    # a zero challenge has no inverse: refused, not mis-verified
    z = copy.deepcopy(vs); z[0]["es"][1] = 0
    with pytest.raises(Exception, match="challenge is zero"):
        _ip_batch_point(gpu, st, z, rhos)


@pytest.mark.parametrize("name", ["64bit", "32bit", "rec_test"])
def test_native_ip_setups_verify_from_files(gpu, name):
    schema, st, proofs = _example(gpu, name, 7)
    nat = RP.NativeRangeProofs(gpu, st)
    assert (nat.shape["rounds"], nat.shape["final_norm"], nat.shape["final_lin"]) == (st.rounds, st.final_lens[0], st.final_lens[1])
    files = [list(E.encode_proof(4, p)) for p in proofs]
    if name == "64bit":
        assert len(files[0][1]) == 418                # the paper's 416-byte proof + 2 sign bytes (README.md:169-172)
    seed = hashlib.sha256(b"ip verifier").digest()
    for host_oracle_max in (2**64 - 1, 0):            # transcript hashing on the host (<= 8 proofs) and on the device
        nat.set_option("host_oracle_max", host_oracle_max)
        ok, status, chs = nat.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
        assert ok and status == [0] * len(files)
        for p, got in zip(proofs, chs):
            assert got == tuple(RP.verifier_challenges(st, p, RP.sha256_oracle()))
    bad = [list(f) for f in files]
    pf = bytearray(bad[2][1]); pf[32 * sum(st.final_lens)] ^= 1; bad[2][1] = bytes(pf)          # a sign bit: another point
    pf = bytearray(bad[5][1]); pf[9] ^= 0x10; bad[5][1] = bytes(pf)                              # a final norm witness scalar
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and status == [0, 0, 1, 0, 0, 1, 0]
    sw = [list(f) for f in files]; sw[1][0] = files[3][0]                                        # another proof's input commitments
    ok, status, _ = nat.verify_batch([c for c, _ in sw], [p for _, p in sw], seed, want_status=True)
    if files[1][0] != files[3][0]:
        assert not ok and status[1] == 1 and sum(status) == 1
    nat.close()


@pytest.mark.parametrize("name", ["64bit", "32bit", "rec_test"])
def test_native_ip_lockstep_prover_equals_host_protocol_bytes(gpu, oracle_lib, name):
    """bppp_rp_prove_batch on an inner-product setup (csrc/rpprove.hip ip_argument_lockstep: every commitment an MSM over the ORIGINAL
    basis — makeNorm's basis change and every point fold carried in the scalars) against rangeproof.prove over the per-proof device
    argument (bppp_ip_*: basis change by scalar multiplications, folds by the reference's rationalReduceScalar pairs) and over the
    ORACLE backend: the same files byte for byte."""
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    st_o = RP.setup_from_schema(OracleBackend(oracle_lib), schema)
    wit_json = json.load(open(os.path.join(EXAMPLES, name, "witness.json")))
    B = 9
    inputs = [RP.inputs_from_witness(wit_json, b"ip lockstep %d" % j) for j in range(B)]
    prefixes = [b"ip lockstep rand %02d" % j for j in range(B)]
    nat = RP.NativeRangeProofs(gpu, st)
    got = nat.prove_batch(inputs, prefixes)
    for b in range(B):
        proof = RP.prove(st, RP.witness(st, inputs[b]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[b]))
        assert got[b] == E.encode_proof(4, proof), "proof %d differs from the host protocol code's" % b
    p_o = RP.prove(st_o, RP.witness(st_o, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0]))
    assert got[0] == E.encode_proof(4, p_o) and RP.verify(st_o, p_o, RP.sha256_oracle())
    assert nat.verify_batch([c for c, _ in got], [p for _, p in got])
    # the device-resident route (csrc/rpprove_dev.hip phases + csrc/ipb.hip: one stream of kernels over a comb table of the setup's basis),
    # transcript hashed on the device and on the host cores: the same bytes again
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 8)
    for host_oracle_max in (0, 2**64 - 1):
        nat.set_option("host_oracle_max", host_oracle_max)
        assert nat.prove_batch(inputs, prefixes) == got, "device-resident inner-product prover differs (host_oracle_max %d)" % host_oracle_max
    nat.close()


def test_native_ip_lockstep_prover_on_a_long_vector(gpu):
    """the 64by64 schema under the inner-product argument (nrmLen 512 -> 256 pairs, linLen 261, 8 rounds, odd lengths on the way down):
    lockstep prover == per-proof device argument, and the batch verifies"""
    schema = dict(json.load(open(os.path.join(EXAMPLES, "64by64", "schema.json"))), argument="IP")
    st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    assert st.flavour == "IP" and st.nrm_len == 512
    rnd = random.Random(64)
    B = 3
    inputs = [[(rnd.randrange(2**64), 0, rnd.randrange(RP.N)) for _ in range(64)] for _ in range(B)]
    prefixes = [b"ip long %d" % j for j in range(B)]
    nat = RP.NativeRangeProofs(gpu, st)
    got = nat.prove_batch(inputs, prefixes)
    proof = RP.prove(st, RP.witness(st, inputs[1]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[1]))
    assert got[1] == E.encode_proof(4, proof)
    assert nat.verify_batch([c for c, _ in got], [p for _, p in got])
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 8); nat.set_option("host_oracle_max", 0)       # device-resident: odd lengths on the way down
    assert nat.prove_batch(inputs, prefixes) == got
    nat.close()


def test_native_ip_verify_larger_batch_64bit(gpu):
    """2^10 files of the 64bit shape (64 distinct proofs tiled: the weights differ by position) through the device oracle, one
    corrupted member found"""
    _, st, proofs = _example(gpu, "64bit", 64)
    nat = RP.NativeRangeProofs(gpu, st)
    files = [E.encode_proof(4, p) for p in proofs] * 16
    assert nat.verify_batch([c for c, _ in files], [p for _, p in files])
    bad = list(files)
    pf = bytearray(bad[777][1]); pf[70] ^= 4; bad[777] = (bad[777][0], bytes(pf))
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], want_status=True)
    assert not ok and [i for i, s_ in enumerate(status) if s_] == [777]
    nat.close()
