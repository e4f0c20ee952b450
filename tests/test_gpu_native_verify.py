"""The native end-to-end batch verifier (csrc/rp.hip: bppp_rp_verify_batch) against the host protocol code.

Proofs are made by bulletproofspp_amd.rangeproof.prove (GPU backend) with the CLI's shaOracle restated in Python
(RP.sha256_oracle), written to the reference's file format (bulletproofspp_amd.encoding) and handed to the library as BYTES: the
library decodes them, hashes every transcript on the device and decides the whole batch with one MSM.  Checks: the device-derived
challenges equal RP.verifier_challenges bit for bit; honest batches are accepted; a tampered / malformed member is rejected and
identified; the full-size configurations of BASELINE.json (configs 4 and 5) go through the same path."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd import encoding as E
from bulletproofspp_amd import rangeproof as RP

pytestmark = pytest.mark.gpu


def _setup(gpu, typed):
    pts = O.hash_points(b"native verify", 120)
    rds = [RP.make_range_data(4, 0, 256, True, True, False), RP.make_range_data(4, 10, 266, True, True, False), RP.make_range_data(16, 0, 2**64, False, True, False),
           RP.make_range_data(3, 0, 100, False, True, False)]
    pub = [(False, 7, 500)] if typed else []
    return RP.setup(RP.GpuBackend(gpu), pts, typed, pub, rds, "NL")


def _proofs(st, n, typed, seed=1):
    rnd = random.Random(seed)
    out = []
    for j in range(n):
        if typed:
            vals = [200, 20, 250, 30]          # outputs of type 7 balancing the public input of 500
            inputs = [(v, 7, rnd.randrange(O.N)) for v in vals]
        else:
            inputs = [(rnd.randrange(256), 0, rnd.randrange(O.N)), (10 + rnd.randrange(256), 0, rnd.randrange(O.N)), (rnd.randrange(2**64), 0, rnd.randrange(O.N)),
                      (rnd.randrange(100), 0, rnd.randrange(O.N))]
        out.append(RP.prove(st, RP.witness(st, inputs), RP.sha256_oracle(), RP.hash_to_scalar(b"nv%d-%d" % (seed, j))))
    return out


@pytest.mark.parametrize("typed", [False, True])
def test_native_challenges_and_accept(gpu, typed):
    st = _setup(gpu, typed)
    proofs = _proofs(st, 5, typed)
    nat = RP.NativeRangeProofs(gpu, st)
    files = [E.encode_proof(4, p) for p in proofs]
    seed = hashlib.sha256(b"verifier randomness").digest()
    ok, status, chs = nat.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
    for p, (ch, es) in zip(proofs, chs):
        want_ch, want_es = RP.verifier_challenges(st, p, RP.sha256_oracle())
        assert ch == want_ch and es == want_es          # every SHA-256 transcript hash and its Binary (Prime p) decode, on the device
        assert RP.verify(st, p, RP.sha256_oracle())
    assert ok and status == [0] * 5
    # a different oracle tag gives different challenges, and the proofs (made for the untagged oracle) no longer verify
    nat2 = RP.NativeRangeProofs(gpu, st, oracle_tag=b"other")
    ok2, _, chs2 = nat2.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
    assert not ok2 and chs2[0][0] == RP.verifier_challenges(st, proofs[0], RP.sha256_oracle(b"other"))[0]
    nat.close(); nat2.close()


def test_native_reject_and_identify(gpu):
    st = _setup(gpu, False)
    proofs = _proofs(st, 9, False, seed=2)
    nat = RP.NativeRangeProofs(gpu, st)
    files = [list(E.encode_proof(4, p)) for p in proofs]
    seed = bytes(range(32))
    assert nat.verify_batch([c for c, _ in files], [p for _, p in files], seed)
    # (a) flip a sign bit of proof 3: a well-formed, different proof
    bad = [list(f) for f in files]
    pf = bytearray(bad[3][1]); pf[32 * sum(st.final_lens)] ^= 1; bad[3][1] = bytes(pf)
    # (b) a final-witness scalar of proof 6 changed
    pf = bytearray(bad[6][1]); pf[7] ^= 0x40; bad[6][1] = bytes(pf)
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and status == [0, 0, 0, 1, 0, 0, 1, 0, 0]
    # (c) an input commitment of proof 0 replaced by an x with no point on the curve: malformed (decodeCommitments = Nothing)
    x_bad = next(x for x in range(2, 100) if pow((x**3 + 7) % O.P, (O.P - 1) // 2, O.P) != 1)
    cf = bytearray(files[0][0]); cf[1:33] = E.put_field(x_bad); mal = [list(f) for f in files]; mal[0][0] = bytes(cf)
    ok, status, _ = nat.verify_batch([c for c, _ in mal], [p for _, p in mal], seed, want_status=True)
    assert not ok and status == [2] + [0] * 8
    # (d) the commitment of another value: verifies as a proof, but not for these inputs
    sw = [list(f) for f in files]; sw[2][0] = files[4][0]
    ok, status, _ = nat.verify_batch([c for c, _ in sw], [p for _, p in sw], seed, want_status=True)
    assert not ok and status[2] == 1 and sum(status) == 1
    # wrong file length
    assert nat.verify_batch([c for c, _ in files], [files[0][1][:-1]] + [p for _, p in files[1:]], seed) is False
    nat.close()
