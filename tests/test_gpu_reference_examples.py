"""The CLI's `test` mode (app/Main.hs:169-205: setup from schema.json, prove witness.json, verify = True) on EVERY example the reference
ships (examples/*, data fixtures under tests/golden/examples), through the library's own end-to-end entry points: the schema's basis
(getPoints over its basisSeed), its argument flavour (IP unless the schema says NL), typed-reciprocal or binary as the schema says;
bppp_rp_prove_batch writes the two files, bppp_rp_verify_batch accepts them, the host protocol code accepts the decoded proof too and
derives the same challenges, and a flipped bit is rejected.  Three proofs per example (the example's amounts under three blindings)."""
import hashlib
import json
import os

import pytest

from bulletproofspp_amd import encoding as E
from bulletproofspp_amd import rangeproof as RP
from bulletproofspp_amd import rangeproof_binary as BRP
from test_rangeproof import EXAMPLES

pytestmark = pytest.mark.gpu

ALL = ["32bit", "64bit", "rec_test", "32by64", "64by64", "96by64", "128by64", "bin_test"]


@pytest.mark.parametrize("name", ALL)
def test_reference_example_through_the_native_layer(gpu, name):
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    wit_json = json.load(open(os.path.join(EXAMPLES, name, "witness.json")))
    binary = bool(schema.get("binary", False))
    if binary:
        st = BRP.setup_from_schema(RP.GpuBackend(gpu), schema)
        nat = BRP.NativeBinaryRangeProofs(gpu, st)
        ncom, verify_host, chal_host = 2, BRP.verify, BRP.verifier_challenges
    else:
        st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
        nat = RP.NativeRangeProofs(gpu, st)
        ncom, verify_host, chal_host = 4, RP.verify, RP.verifier_challenges
    assert st.flavour == {"ip": "IP", "nl": "NL"}[str(schema.get("argument", "IP")).lower()]
    B = 3
    inputs = []
    for j in range(B):
        rows = RP.inputs_from_witness(wit_json, b"examples %d" % j) if j else RP.inputs_from_witness(wit_json)
        inputs.append([(v, bl) for v, _, bl in rows] if binary else rows)
    files = nat.prove_batch(inputs, [b"default random seed", b"another random seed", b"a third random seed"])
    seed = hashlib.sha256(name.encode()).digest()
    ok, status, chs = nat.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
    assert ok and status == [0] * B
    lift = E.gpu_lift_x(gpu)
    for (cf, pf), ch in zip(files, chs):
        coms = E.decode_commitments(len(st.rds), cf, lift)[0]
        proof = E.decode_proof(ncom, st.rounds, st.final_lens, coms, pf, lift)
        assert proof is not None and verify_host(st, proof, RP.sha256_oracle())
        want = chal_host(st, proof, RP.sha256_oracle())
        assert tuple(ch) == tuple(want)
    bad = [list(f) for f in files]
    pf = bytearray(bad[1][1]); pf[5] ^= 2; bad[1][1] = bytes(pf)                     # a bit of the first final-witness scalar
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and status == [0, 1, 0]
    nat.close()
