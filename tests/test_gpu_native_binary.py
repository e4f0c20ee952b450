"""RangeProof.Binary natively (bppp_rp_create_binary + the bppp_rp_* entry points; csrc/rp.hip k_brp_public, csrc/rpprove.hip
prove_batch_binary): the library's lockstep prover writes the SAME BYTES as the host protocol code (bulletproofspp_amd/rangeproof_binary.py:
proveBRPM, src/RangeProof/Binary.hs:169-204) — which itself gives the oracle backend's transcript bit for bit — and the end-to-end
verifier derives the same challenges, accepts honest batches and identifies tampered members.  Cases: the reference's
examples/bin_test (conserved, three ranges, nrmLen 192, 6 rounds) and the shapes of tests/test_rangeproof.py (odd widths, an
assumed range, top-digit boundaries, a negative minimum)."""
import hashlib
import json
import os
import random

import pytest

import pyoracle as O
from bulletproofspp_amd import encoding as E
from bulletproofspp_amd import rangeproof as RP
from bulletproofspp_amd import rangeproof_binary as BRP
from rp_backends import OracleBackend
from test_rangeproof import BIN_CASES, EXAMPLES

pytestmark = pytest.mark.gpu

CASES = dict(BIN_CASES)
CASES["negative_minimum"] = ([(-50, 75, True, False), (-8, 8, False, False)], 40, [37, -3])


@pytest.mark.parametrize("name", list(CASES))
def test_native_binary_prover_and_verifier(gpu, oracle_lib, name):
    ranges, net, vals = CASES[name]
    rds = [BRP.make_range_data(*r) for r in ranges]
    pts = O.hash_points(b"native binary", 4 + sum(len(rd.base_coeffs) for rd in rds))
    st = BRP.setup(RP.GpuBackend(gpu), pts, True, rds, net, "NL")
    st_o = BRP.setup(OracleBackend(oracle_lib), pts, True, rds, net, "NL")
    rnd = random.Random(name)
    B = 5
    inputs = [[(v, rnd.randrange(RP.N)) for v in vals] for _ in range(B)]
    prefixes = [b"bin %s %02d" % (name.encode(), b) for b in range(B)]
    nat = BRP.NativeBinaryRangeProofs(gpu, st)
    assert nat.shape["lin_len"] == 2 and nat.shape["challenges_per_proof"] == 4 + st.rounds
    got = nat.prove_batch(inputs, prefixes)
    proofs = []
    for b in range(B):
        proof = BRP.prove(st, BRP.witness(st, inputs[b]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[b]))
        proofs.append(proof)
        want = E.encode_proof(2, proof)
        assert got[b][0] == want[0], "commitments file differs (proof %d)" % b
        assert got[b][1] == want[1], "proof file differs (proof %d)" % b
        assert BRP.verify(st, proof, RP.sha256_oracle())
    # the device-resident route (csrc/brpprove_dev.hip: field algebra, randomness, transcript and the fixed-basis argument as one stream of
    # kernels over a comb table of the setup's basis): the same bytes, with the transcript hashed on the device and on the host cores;
    # then the host-algebra cross-check (prove_batch_binary) once more
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 7)
    for host_oracle_max in (0, 2**64 - 1):
        nat.set_option("host_oracle_max", host_oracle_max)
        assert nat.prove_batch(inputs, prefixes) == got, "device-resident binary prover differs (host_oracle_max %d)" % host_oracle_max
    nat.set_option("host_algebra", 1)
    assert nat.prove_batch(inputs, prefixes) == got
    nat.set_option("host_algebra", 0)
    # proof 0 again with every group operation done by the oracle (Straus commits, the reference's fold): the same bytes
    p_o = BRP.prove(st_o, BRP.witness(st_o, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0]))
    assert E.encode_proof(2, p_o) == got[0] and BRP.verify(st_o, p_o, RP.sha256_oracle())
    seed = hashlib.sha256(b"binary verifier").digest()
    for host_oracle_max in (2**64 - 1, 0):                                 # transcript hashing on the host (<= 8 proofs) and on the device
        nat.set_option("host_oracle_max", host_oracle_max)
        ok, status, chs = nat.verify_batch([c for c, _ in got], [p for _, p in got], seed, want_status=True, want_challenges=True)
        assert ok and status == [0] * B
        for p, (lead, es) in zip(proofs, chs):
            want_lead, want_es = BRP.verifier_challenges(st, p, RP.sha256_oracle())
            assert lead == want_lead and es == want_es
    bad = [list(f) for f in got]
    pf = bytearray(bad[1][1]); pf[32 * sum(st.final_lens)] ^= 1; bad[1][1] = bytes(pf)           # a sign bit of blCom
    pf = bytearray(bad[3][1]); pf[11] ^= 0x20; bad[3][1] = bytes(pf)                             # a final witness scalar
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and status == [0, 1, 0, 1, 0]
    x_bad = next(x for x in range(2, 100) if pow((x**3 + 7) % O.P, (O.P - 1) // 2, O.P) != 1)
    cf = bytearray(got[2][0]); cf[1:33] = E.put_field(x_bad); mal = [list(f) for f in got]; mal[2][0] = bytes(cf)
    ok, status, _ = nat.verify_batch([c for c, _ in mal], [p for _, p in mal], seed, want_status=True)
    assert not ok and status == [0, 0, 2, 0, 0]
    # a value outside its range, and amounts that do not balance, are refused
    out_of_range = [list(r) for r in inputs[:2]]
    out_of_range[1][0] = (ranges[0][1], out_of_range[1][0][1])
    with pytest.raises(Exception, match="proof 1"):
        nat.prove_batch(out_of_range, prefixes[:2])
    off = [list(r) for r in inputs[:1]]
    lo0, hi0 = ranges[0][0], ranges[0][1]
    off[0][0] = (vals[0] + 1 if vals[0] + 1 < hi0 else vals[0] - 1, off[0][0][1])
    with pytest.raises(Exception, match="balance"):
        nat.prove_batch(off, prefixes[:1])
    nat.close()


def test_native_binary_on_the_reference_example(gpu):
    schema = json.load(open(os.path.join(EXAMPLES, "bin_test", "schema.json")))
    st = BRP.setup_from_schema(RP.GpuBackend(gpu), schema)
    assert (st.flavour, st.nrm_len, st.rounds, st.final_lens) == ("NL", 192, 6, (3, 1))
    inputs = RP.inputs_from_witness(json.load(open(os.path.join(EXAMPLES, "bin_test", "witness.json"))))
    row = [(v, bl) for v, _, bl in inputs]
    nat = BRP.NativeBinaryRangeProofs(gpu, st)
    assert nat.shape["proof_bytes"] == 32 * 4 + 2 + 32 * 14          # 14 points + 4 scalars
    B = 70                                                           # the same amounts under 70 different prover randomness streams
    files = nat.prove_batch([row] * B, [b"bin_test %03d" % b for b in range(B)])
    proof = BRP.prove(st, BRP.witness(st, row), RP.sha256_oracle(), RP.hash_to_scalar(b"bin_test 000"))
    assert files[0] == E.encode_proof(2, proof) and len({p for _, p in files}) == B
    assert nat.verify_batch([c for c, _ in files], [p for _, p in files])
    bad = list(files); pf = bytearray(bad[41][1]); pf[200] ^= 8; bad[41] = (bad[41][0], bytes(pf))
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], want_status=True)
    assert not ok and [i for i, s_ in enumerate(status) if s_] == [41]
    # the inner-product flavour of the same schema: the library verifies what the host protocol code proves
    st_ip = BRP.setup_from_schema(RP.GpuBackend(gpu), dict(schema, argument="IP"))
    p_ip = BRP.prove(st_ip, BRP.witness(st_ip, row), RP.sha256_oracle(), RP.hash_to_scalar(b"bin ip"))
    assert BRP.verify(st_ip, p_ip, RP.sha256_oracle())
    nat_ip = BRP.NativeBinaryRangeProofs(gpu, st_ip)
    c_ip, f_ip = E.encode_proof(2, p_ip)
    assert nat_ip.verify_batch([c_ip], [f_ip])
    assert nat_ip.prove_batch([row], [b"bin ip"])[0] == (c_ip, f_ip)                 # the binary prover over the lockstep inner-product argument
    # ... and device-resident (csrc/brpprove_dev.hip + csrc/ipb.hip over a comb table), 70 randomness streams, both hashing routes
    ip_files = nat_ip.prove_batch([row] * B, [b"bin ip %03d" % b for b in range(B)])
    nat_ip.set_option("comb_min", 1); nat_ip.set_option("comb_bits", 7)
    for host_oracle_max in (0, 2**64 - 1):
        nat_ip.set_option("host_oracle_max", host_oracle_max)
        assert nat_ip.prove_batch([row] * B, [b"bin ip %03d" % b for b in range(B)]) == ip_files
    assert nat_ip.prove_batch([row], [b"bin ip"])[0] == (c_ip, f_ip) and nat_ip.verify_batch([c for c, _ in ip_files], [p for _, p in ip_files])
    # the norm-linear setup the same way: the device route writes what the host-algebra route wrote above
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 7); nat.set_option("host_oracle_max", 0)
    assert nat.prove_batch([row] * B, [b"bin_test %03d" % b for b in range(B)]) == files
    t_ip = bytearray(f_ip); t_ip[40] ^= 1
    assert not nat_ip.verify_batch([c_ip], [bytes(t_ip)])
    nat_ip.close(); nat.close()


def test_native_binary_at_the_64_by_64_bit_shape(gpu, monkeypatch):
    """BASELINE config 3 read literally — a 64 x 64-bit aggregated BINARY range proof: 64 outputs in [0, 2^64), nrmLen 4096, 10 rounds,
    conserved against one public input.  The lockstep prover's files verify, the end-to-end verifier derives the challenges the host
    protocol code derives from the same files (both transcript-hashing routes), and a tampered member is identified.  (Byte equality
    with the host prover is asserted on the smaller shapes above; at 4096 positions the Python prover takes minutes.)"""
    count, amount = 64, 10000
    rds = [BRP.make_range_data(0, 2**64, True, False)] * count
    pts = O.hash_points(b"binary 64by64", 4 + 64 * count)
    st = BRP.setup(RP.GpuBackend(gpu), pts, True, rds, amount * count, "NL")
    assert (st.nrm_len, st.rounds) == (4096, 10)
    nat = BRP.NativeBinaryRangeProofs(gpu, st)
    assert nat.shape["proof_bytes"] == 867 and nat.shape["challenges_per_proof"] == 14
    rnd = random.Random(64)
    B = 3
    inputs = []
    for _ in range(B):
        d = [rnd.randrange(-5000, 5000) for _ in range(count // 2)]
        inputs.append([(amount + x, rnd.randrange(RP.N)) for x in d] + [(amount - x, rnd.randrange(RP.N)) for x in d])
    files = nat.prove_batch(inputs, [b"bin64 %d" % b for b in range(B)])
    # the device-resident route over a (narrow, 1.3-GB) comb table of the 4099 points: byte-identical to the host-algebra route above
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 9); nat.set_option("host_oracle_max", 0)
    assert nat.prove_batch(inputs, [b"bin64 %d" % b for b in range(B)]) == files
    # ... and with the argument re-based after 3 / 4 folds (the default for batches of 384 proofs or more: csrc/nlb.hip): the same bytes
    for level in ("3", "4"):
        monkeypatch.setenv("BPPP_NLB_REBASE", level)
        assert nat.prove_batch(inputs, [b"bin64 %d" % b for b in range(B)]) == files
    monkeypatch.delenv("BPPP_NLB_REBASE")
    nat.set_option("host_oracle_max", 2**64 - 1)
    seed = hashlib.sha256(b"binary 64by64").digest()
    lift = E.gpu_lift_x(gpu)
    for host_oracle_max in (2**64 - 1, 0):
        nat.set_option("host_oracle_max", host_oracle_max)
        ok, status, chs = nat.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
        assert ok and status == [0] * B
        for (cf, pf), (lead, es) in zip(files, chs):
            coms = E.decode_commitments(count, cf, lift)[0]
            proof = E.decode_proof(2, st.rounds, st.final_lens, coms, pf, lift)
            want_lead, want_es = BRP.verifier_challenges(st, proof, RP.sha256_oracle())
            assert lead == want_lead and es == want_es
    bad = [list(f) for f in files]
    pf = bytearray(bad[2][1]); pf[9] ^= 4; bad[2][1] = bytes(pf)
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and status == [0, 0, 1]
    nat.close()
