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

_OPT_DEFAULTS = {"comb_min": 1024, "comb_bits": 0, "split_min": 4096, "host_oracle_max": 2**64 - 1, "fold_points": 0, "host_algebra": 0}


class _options:
    """bppp_rp_set_option for the duration of a block (the handle's defaults afterwards)"""

    def __init__(self, nat, **kw):
        self.nat, self.kw = nat, kw

    def __enter__(self):
        for k, v in self.kw.items():
            self.nat.set_option(k, v)

    def __exit__(self, *exc):
        for k in self.kw:
            self.nat.set_option(k, _OPT_DEFAULTS[k])


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


def test_sharded_weights_are_global_and_bound_to_the_proofs(gpu):
    """The job sharded proof-per-GPU (bppp_rp_verify_shard_device): the weights rho are indexed by the position in the
    JOB and bound to the proof bytes.  Two copies of one honest proof whose first linear witness scalar is shifted by +d and -d have
    error terms that cancel under EQUAL weights (the final witness scalars enter no transcript hash, so both copies have the same
    challenges): placed on two ranks, each at its local slot 0, they must still be rejected by the sum of the rank points.  And the
    sum of the rank points equals the combined point one rank forms over the whole job."""
    from bulletproofspp_amd.capi import points_to_array
    st = _setup(gpu, False)
    proofs = _proofs(st, 3, False, seed=5)
    nat = RP.NativeRangeProofs(gpu, st)
    files = [E.encode_proof(4, p) for p in proofs]
    fn = st.final_lens[0]
    d = 0x1234567

    def shifted(pf, delta):
        b = bytearray(pf)
        v = E.get_field(bytes(b[32 * fn:32 * fn + 32]), O.N)
        b[32 * fn:32 * fn + 32] = E.put_field((v + delta) % O.N)
        return bytes(b)
    plus, minus = shifted(files[0][1], d), shifted(files[0][1], -d)
    seed = hashlib.sha256(b"one seed for every rank").digest()
    def up(bs):                                   # device buffers through the library (this process has no torch GPU context)
        raw = b"".join(bs)
        return gpu.to_device(np.frombuffer(raw + b"\0" * (-len(raw) % 8), dtype=np.uint64))

    def shard(coms, prfs, offset):
        dc, dp = up(coms), up(prfs)
        try:
            return nat.verify_batch_device_point(len(prfs), dc, dp, seed, index_offset=offset)
        finally:
            gpu.free(dc); gpu.free(dp)
    c0 = files[0][0]
    ok_a, pt_a = shard([c0], [plus], 0)
    ok_b, pt_b = shard([c0], [minus], 1)
    assert not ok_a and not ok_b and pt_a is not None and pt_b is not None
    assert gpu.sum_points(points_to_array([pt_a, pt_b])) is not None, "cancelling pair split across two ranks was accepted"
    # the same pair inside ONE batch: rejected too (no weight is fixed, every weight depends on the proof's own bytes)
    ok_ab, pt_ab = shard([c0, c0], [plus, minus], 0)
    assert not ok_ab and pt_ab == gpu.sum_points(points_to_array([pt_a, pt_b]))
    # with the offsets dropped (both shards at job position 0) the two error terms differ only by the bytes of the final witness:
    # still no cancellation, because rho hashes those bytes
    ok_c, pt_c = shard([c0], [minus], 0)
    assert gpu.sum_points(points_to_array([pt_a, pt_c])) is not None
    # honest job [p0, p1 | p2] on two ranks: every rank accepts, the points are the identity; a job with one bad member equals the
    # single-rank combination point for point
    ok0, q0 = shard([files[0][0], files[1][0]], [files[0][1], files[1][1]], 0)
    ok1, q1 = shard([files[2][0]], [files[2][1]], 2)
    assert ok0 and ok1 and q0 is None and q1 is None
    ok0, q0 = shard([files[0][0], files[1][0]], [plus, files[1][1]], 0)
    okw, qw = shard([f[0] for f in files], [plus, files[1][1], files[2][1]], 0)
    assert not ok0 and not okw and qw == gpu.sum_points(points_to_array([q0, q1]))
    nat.close()


def test_eight_shards_of_one_job(gpu):
    """BASELINE config 5's layout, shard by shard in one process (a one-GPU box admits at most 6 processes on its card; 2, 4 and 5 real ranks
    run in tests/test_gpu_two_ranks.py): ONE job of 44 proofs cut into the 8 contiguous shards dist.shard_range gives 8 ranks (6 + 6 + 6 + 6 +
    5 + 5 + 5 + 5), each verified by bppp_rp_verify_shard_device with the job-wide seed and its own offset.  Honest job: every shard's point is
    the identity.  One corrupted proof in shard 5: only that shard rejects, the sum of the 8 points is not the identity and equals, point
    for point, what one rank forms over the whole job."""
    from bulletproofspp_amd.capi import points_to_array
    from bulletproofspp_amd.dist import shard_range
    st = _setup(gpu, False)
    nat = RP.NativeRangeProofs(gpu, st)
    rnd = random.Random(85)
    J, W = 44, 8
    inputs = [[(rnd.randrange(256), 0, rnd.randrange(O.N)), (10 + rnd.randrange(256), 0, rnd.randrange(O.N)), (rnd.randrange(2**64), 0, rnd.randrange(O.N)),
               (rnd.randrange(100), 0, rnd.randrange(O.N))] for _ in range(J)]
    files = nat.prove_batch(inputs, [b"eight shards %02d" % b for b in range(J)])
    seed = hashlib.sha256(b"job seed, broadcast by rank 0").digest()

    def up(bs):
        raw = b"".join(bs)
        return gpu.to_device(np.frombuffer(raw + b"\0" * (-len(raw) % 8), dtype=np.uint64))

    def shard(fs, offset):
        dc, dp = up([c for c, _ in fs]), up([p for _, p in fs])
        try:
            return nat.verify_batch_device_point(len(fs), dc, dp, seed, index_offset=offset)
        finally:
            gpu.free(dc); gpu.free(dp)
    ranges = [shard_range(J, r, W) for r in range(W)]
    assert [hi - lo for lo, hi in ranges] == [6, 6, 6, 6, 5, 5, 5, 5]
    res = [shard(files[lo:hi], lo) for lo, hi in ranges]
    assert all(ok and pt is None for ok, pt in res)
    bad = list(files)
    lo5, hi5 = ranges[5]
    pf = bytearray(bad[lo5 + 2][1]); pf[9] ^= 4; bad[lo5 + 2] = (bad[lo5 + 2][0], bytes(pf))
    res = [shard(bad[lo:hi], lo) for lo, hi in ranges]
    assert [ok for ok, _ in res] == [r != 5 for r in range(W)]
    total = gpu.sum_points(points_to_array([pt for _, pt in res]))
    ok_all, pt_all = shard(bad, 0)
    assert total is not None and not ok_all and pt_all == total
    nat.close()


@pytest.mark.parametrize("typed", [False, True])
def test_native_prover_equals_host_protocol_bytes(gpu, typed):
    """bppp_rp_prove_batch against rangeproof.prove: same inputs, same hashToScalar randomness, same oracle => the same
    commitments file and proof file byte for byte (every commitment, response and final witness scalar)."""
    st = _setup(gpu, typed)
    nat = RP.NativeRangeProofs(gpu, st)
    rnd = random.Random(77)
    B = 6
    if typed:
        inputs = [[(v, 7, rnd.randrange(O.N)) for v in (200, 20, 250, 30)] for _ in range(B)]
    else:
        inputs = [[(rnd.randrange(256), 0, rnd.randrange(O.N)), (10 + rnd.randrange(256), 0, rnd.randrange(O.N)), (rnd.randrange(2**64), 0, rnd.randrange(O.N)),
                   (rnd.randrange(100), 0, rnd.randrange(O.N))] for _ in range(B)]
    inputs[0][0] = (0, inputs[0][0][1], inputs[0][0][2]) if not typed else inputs[0][0]            # range minimum
    if not typed:
        inputs[1] = [(255, 0, 1), (265, 0, 2), (2**64 - 1, 0, 3), (99, 0, 4)]                      # every range at its maximum
    prefixes = [b"native prover %02d" % b for b in range(B)]
    got = nat.prove_batch(inputs, prefixes)
    for b in range(B):
        proof = RP.prove(st, RP.witness(st, inputs[b]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[b]))
        want = E.encode_proof(4, proof)
        assert got[b][0] == want[0], "commitments file differs (proof %d)" % b
        assert got[b][1] == want[1], "proof file differs (proof %d)" % b
    assert nat.verify_batch([c for c, _ in got], [p for _, p in got], b"\x07" * 32)
    # the host-algebra path of the library (per-proof field work and hashing in C++ on the host cores; the default does both on the
    # device, csrc/rpprove_dev.hip) writes the same bytes
    with _options(nat, host_algebra=1):
        assert nat.prove_batch(inputs, prefixes) == got
    # the fixed-basis route (large batches take it by default: a comb table over the setup's [g | H | G], csrc/comb.hip — the
    # range-proof commitments and EVERY round commitment of the argument are comb MSMs over the original points with the fold
    # coefficients multiplied into the scalars; no point is folded, no half-GCD is taken, csrc/nlb.hip) writes the same bytes as well
    with _options(nat, comb_min=1):
        assert nat.prove_batch(inputs, prefixes) == got
        assert nat.prove_batch(inputs[:3], prefixes[:3]) == got[:3]      # the table is kept by the handle across batches
        # (up to 64 proofs the oracle of that route runs on the host; with it on the device, as for larger batches: the same bytes)
        with _options(nat, host_oracle_max=0):
            assert nat.prove_batch(inputs, prefixes) == got
    # ... and with the table in place the point-folding route of the argument is still there (the general bppp_nlb_* entry points use it)
    with _options(nat, fold_points=1):
        assert nat.prove_batch(inputs, prefixes) == got
    # two half-batches in flight on two contexts (the default for large batches, csrc/rpprove.hip): the same bytes, and a refused
    # input in the second half is reported under its index in the whole batch
    with _options(nat, split_min=2):
        assert nat.prove_batch(inputs, prefixes) == got
        assert nat.prove_batch(inputs[:5], prefixes[:5]) == got[:5]
        worse = [list(r) for r in inputs]
        worse[4][3] = (100 if not typed else 1000, worse[4][3][1], worse[4][3][2])
        with pytest.raises(Exception, match="proof 4"):
            nat.prove_batch(worse, prefixes)
    # a value outside its range is refused
    bad = [list(r) for r in inputs]
    bad[2][3] = (100 if not typed else 1000, bad[2][3][1], bad[2][3][2])
    with pytest.raises(Exception):
        nat.prove_batch(bad, prefixes)
    nat.close()


def test_native_prover_routes_agree_over_batch_sizes(gpu):
    """The two routes of the library's prover — point-folding argument + bucket MSMs, and fixed-basis mode (comb table, stream of
    kernels, several wavefronts per instance when the launch is small, host oracle up to 8 proofs, two half-batches in flight) — on
    batch sizes around every internal boundary (1, the host-oracle limit, 64-lane groups, odd splits): byte-identical files."""
    st = _setup(gpu, False)
    rnd = random.Random(99)
    sizes = [1, 2, 3, 8, 9, 31, 64, 65, 130]
    nmax = max(sizes)
    inputs = [[(rnd.randrange(256), 0, rnd.randrange(O.N)), (10 + rnd.randrange(256), 0, rnd.randrange(O.N)), (rnd.randrange(2**64), 0, rnd.randrange(O.N)),
               (rnd.randrange(100), 0, rnd.randrange(O.N))] for _ in range(nmax)]
    prefixes = [b"routes %04d" % b for b in range(nmax)]
    fold = RP.NativeRangeProofs(gpu, st)
    want = fold.prove_batch(inputs, prefixes)                       # below the comb threshold: the point-folding route
    assert fold.verify_batch([c for c, _ in want], [p for _, p in want], b"\x21" * 32)
    fold.close()
    nat = RP.NativeRangeProofs(gpu, st)
    with _options(nat, comb_min=1, comb_bits=7):
        for n in sizes:
            assert nat.prove_batch(inputs[:n], prefixes[:n]) == want[:n], n
        with _options(nat, split_min=2):
            for n in (2, 9, 65, 130):
                assert nat.prove_batch(inputs[:n], prefixes[:n]) == want[:n], n
    nat.close()


def test_comb_table_allocation_failure_falls_back_to_the_bucket_route(gpu):
    """A comb table that cannot be allocated (a forced 18-bit window over 2662 points = 335 GB, more than the card has) must not fail
    the batch it was meant to speed up: the handle notes the failure, clears the runtime's sticky error and proves on the bucket /
    point-folding route — byte-identical to a handle that never builds a table; later batches work too."""
    pts = O.hash_points(b"comb alloc failure", 2 + 261 + 2400)
    rds = [RP.make_range_data(256, 0, 2**64, True, True, False)] * 300
    st = RP.setup(RP.GpuBackend(gpu), pts, False, [], rds, "NL")
    assert (st.nrm_len, st.lin_len) == (2400, 261)
    rnd = random.Random(4)
    inputs = [[(rnd.randrange(2**64), 0, rnd.randrange(O.N)) for _ in range(300)] for _ in range(2)]
    prefixes = [b"alloc %d" % b for b in range(2)]
    plain = RP.NativeRangeProofs(gpu, st)
    plain.set_option("comb_budget", 0)
    want = plain.prove_batch(inputs, prefixes)
    assert plain.verify_batch([c for c, _ in want], [p for _, p in want])
    plain.close()
    nat = RP.NativeRangeProofs(gpu, st)
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 18)
    assert nat.prove_batch(inputs, prefixes) == want
    assert nat.prove_batch(inputs[:1], prefixes[:1]) == want[:1]
    assert nat.verify_batch([c for c, _ in want], [p for _, p in want])
    nat.close()


def _example_setup(gpu, name, typed_override=None):
    from test_rangeproof import EXAMPLES
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    if typed_override:
        schema = dict(schema, typed=True)
    count = sum(int(r.get("count", 1)) for r in schema["ranges"])
    return RP.setup_from_schema(RP.GpuBackend(gpu), schema), count


def test_config5_batch_of_4096_proofs_64by64(gpu):
    """BASELINE config 5 on one GPU: 2^12 DISTINCT aggregated 64 x 64-bit proofs (examples/64by64: nrmLen 512, linLen 261, 8 rounds)
    made by the lockstep prover, verified end to end from their bytes with one combined MSM of 344 838 terms; then one of them
    corrupted => rejected and identified."""
    st, count = _example_setup(gpu, "64by64")
    assert (st.nrm_len, st.lin_len, st.rounds, st.final_lens) == (512, 261, 8, (2, 2)) and count == 64
    nat = RP.NativeRangeProofs(gpu, st)
    B = 4096
    rng = np.random.default_rng(5)
    amounts = rng.integers(0, 2**63, size=(B, count), dtype=np.uint64)
    blinds = rng.integers(1, 2**63, size=(B, count), dtype=np.uint64)
    inputs = [[(int(a) * 2 + (i & 1), 0, int(bl)) for i, (a, bl) in enumerate(zip(amounts[b], blinds[b]))] for b in range(B)]
    files = nat.prove_batch(inputs, [b"cfg5 %06d" % b for b in range(B)])
    assert len({p for _, p in files}) == B
    seed = hashlib.sha256(b"cfg5").digest()
    ok, status, chs = nat.verify_batch([c for c, _ in files], [p for _, p in files], seed, want_status=True, want_challenges=True)
    assert ok and status == [0] * B
    # one proof of the batch checked against the host protocol code: same challenges, and it verifies there too
    proof = E.decode_proof(4, st.rounds, st.final_lens, E.decode_commitments(count, files[17][0], E.gpu_lift_x(gpu))[0], files[17][1], E.gpu_lift_x(gpu))
    assert chs[17] == tuple(RP.verifier_challenges(st, proof, RP.sha256_oracle())) and RP.verify(st, proof, RP.sha256_oracle())
    bad = list(files)
    pf = bytearray(bad[1234][1]); pf[40] ^= 1; bad[1234] = (bad[1234][0], bytes(pf))
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and [i for i, s_ in enumerate(status) if s_] == [1234] and status[1234] == 1
    # the host-buffer entry point uploads a batch of this size in four slices (decode of one under the upload of the next): a commitment
    # with no point on the curve in the LAST slice is reported as malformed next to the invalid proof of the second, and the files
    # already resident in HBM give the same answer
    x_bad = next(x for x in range(2, 100) if pow((x**3 + 7) % O.P, (O.P - 1) // 2, O.P) != 1)
    cf = bytearray(bad[4000][0]); cf[8:40] = E.put_field(x_bad); bad[4000] = (bytes(cf), bad[4000][1])
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and {i: s_ for i, s_ in enumerate(status) if s_} == {1234: 1, 4000: 2}
    dc = gpu.to_device(np.frombuffer(b"".join(c for c, _ in bad), dtype=np.uint8)); dp = gpu.to_device(np.frombuffer(b"".join(p for _, p in bad), dtype=np.uint8))
    ok2, status2, _ = nat.verify_batch_device(B, dc, dp, seed, want_status=True)
    gpu.free(dc); gpu.free(dp)
    assert not ok2 and list(status2) == list(status)
    nat.close()


def test_config4_typed_conserved_128by64(gpu):
    """BASELINE config 4: 128 x 64-bit values, typed reciprocal proof WITH conservation (examples/128by64 plus "typed": nrmLen 1152,
    9 rounds, final (3, 1), a 1564-term verifier MSM per proof), batch-verified end to end; a proof whose public input does not
    balance cannot be made; a corrupted member is rejected."""
    from test_rangeproof import EXAMPLES
    schema = json.load(open(os.path.join(EXAMPLES, "128by64", "schema.json")))
    schema = dict(schema, typed=True, public=[{"amount": 128 * 10000, "type": 0}])
    st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    assert (st.nrm_len, st.lin_len, st.rounds, st.final_lens) == (1152, 261, 9, (3, 1)) and st.has_types
    nat = RP.NativeRangeProofs(gpu, st)
    B = 256
    rng = np.random.default_rng(4)
    inputs = []
    for b in range(B):
        # 128 outputs of type 0 that sum to the public input 1 280 000: random split around 10 000 each
        d = rng.integers(-5000, 5000, size=64)
        vals = [10000 + int(x) for x in d] + [10000 - int(x) for x in d]
        inputs.append([(v, 0, int(bl)) for v, bl in zip(vals, rng.integers(1, 2**63, size=128, dtype=np.uint64))])
    files = nat.prove_batch(inputs, [b"cfg4 %04d" % b for b in range(B)])
    seed = hashlib.sha256(b"cfg4").digest()
    assert nat.verify_batch([c for c, _ in files], [p for _, p in files], seed)
    proof = E.decode_proof(4, st.rounds, st.final_lens, E.decode_commitments(128, files[3][0], E.gpu_lift_x(gpu))[0], files[3][1], E.gpu_lift_x(gpu))
    assert RP.verify(st, proof, RP.sha256_oracle())
    unbalanced = [list(r) for r in inputs[:4]]
    unbalanced[1][0] = (unbalanced[1][0][0] + 1, 0, unbalanced[1][0][2])
    with pytest.raises(Exception):
        nat.prove_batch(unbalanced, [b"x%d" % b for b in range(4)])
    bad = list(files)
    cf = bytearray(bad[200][0]); cf[0] ^= 2; bad[200] = (bytes(cf), bad[200][1])        # the sign of one input commitment
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and [i for i, s_ in enumerate(status) if s_] == [200]
    nat.close()


def test_native_layer_on_a_mixed_schema(gpu):
    """A schema that exercises every Phase1 constructor at once (TypedReciprocal.hs:56-60, :133-169): shared digits with a leading bit
    (adds the shared base 2), inline digits with and without a bit, an ASSUMED range (typed only: no digits), inputs and outputs of two
    types balanced by public amounts.  Native prover == host protocol bytes (device-algebra and host-algebra paths), native verifier
    accepts / identifies a forged member, challenges equal the host's."""
    pts = O.hash_points(b"mixed schema", 160)
    rds = [RP.make_range_data(4, 0, 101, True, True, False),        # shared, has_bit
           RP.make_range_data(4, 0, 256, True, False, False),       # shared, no bit, an INPUT
           RP.make_range_data(3, 0, 100, False, True, False),       # inline, has_bit
           RP.make_range_data(16, 0, 2**64, False, True, False),    # inline, no bit
           RP.make_range_data(5, 0, 1000, False, False, True)]      # assumed input: typing only
    assert rds[0].has_bit and not rds[1].has_bit and rds[2].has_bit and not rds[3].has_bit
    # type 3: inputs 200 (range 1) + public 50 = outputs 100 (range 0) + 150 (range 3);  type 9: input 77 (assumed range 4) = output 70 (range 2) + public output 7
    pub = [(False, 3, 50), (True, 9, 7)]
    st = RP.setup(RP.GpuBackend(gpu), pts, True, pub, rds, "NL")
    assert 2 in st.m_bases and 4 in st.m_bases
    nat = RP.NativeRangeProofs(gpu, st)
    rnd = random.Random(99)
    B = 5
    inputs = [[(100, 3, rnd.randrange(O.N)), (200, 3, rnd.randrange(O.N)), (70, 9, rnd.randrange(O.N)), (150, 3, rnd.randrange(O.N)), (77, 9, rnd.randrange(O.N))]
              for _ in range(B)]
    prefixes = [b"mixed %03d" % b for b in range(B)]
    got = nat.prove_batch(inputs, prefixes)
    for b in range(B):
        proof = RP.prove(st, RP.witness(st, inputs[b]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[b]))
        assert got[b] == E.encode_proof(4, proof), b
        assert RP.verify(st, proof, RP.sha256_oracle())
    with _options(nat, host_algebra=1):
        assert nat.prove_batch(inputs, prefixes) == got
    seed = bytes(range(1, 33))
    ok, status, chs = nat.verify_batch([c for c, _ in got], [p for _, p in got], seed, want_status=True, want_challenges=True)
    assert ok and status == [0] * B
    proof0 = E.decode_proof(4, st.rounds, st.final_lens, E.decode_commitments(5, got[0][0], E.gpu_lift_x(gpu))[0], got[0][1], E.gpu_lift_x(gpu))
    assert chs[0] == tuple(RP.verifier_challenges(st, proof0, RP.sha256_oracle()))
    forged = list(got)
    forged[3] = (got[2][0], got[3][1])                              # another proof's input commitments
    ok, status, _ = nat.verify_batch([c for c, _ in forged], [p for _, p in forged], seed, want_status=True)
    assert not ok and status == [0, 0, 0, 1, 0]
    # the same schema under the inner-product argument (tests/test_gpu_native_verify_ip.py)
    st_ip = RP.setup(RP.GpuBackend(gpu), pts, True, pub, rds, "IP")
    nat_ip = RP.NativeRangeProofs(gpu, st_ip)
    p_ip = RP.prove(st_ip, RP.witness(st_ip, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0]))
    c_ip, f_ip = E.encode_proof(4, p_ip)
    t_ip = bytearray(f_ip); t_ip[5] ^= 2
    assert nat_ip.verify_batch([c_ip], [f_ip]) and not nat_ip.verify_batch([c_ip], [bytes(t_ip)])
    assert nat_ip.prove_batch(inputs[:1], prefixes[:1])[0] == (c_ip, f_ip)          # its lockstep prover: the same bytes
    nat_ip.close()
    nat.close()


def test_native_setup_refuses_what_the_reference_missizes_and_what_is_too_large(gpu):
    """bppp_rp_create on untrusted range lists: the padded inline layout of tests/test_rangeproof.py (base - 1 symbols > digits: the reference's own
    setup mis-sizes it) and a base that asks for billions of linear entries are refused with a message, nothing is allocated for them"""
    pts = O.hash_points(b"refused layouts", 40)
    st = RP.setup(RP.GpuBackend(gpu), pts, False, [], [RP.make_range_data(16, 0, 100, False, True, False)], "NL")
    with pytest.raises(Exception, match="unsupported layout"):
        RP.NativeRangeProofs(gpu, st)
    import ctypes as C
    from bulletproofspp_amd.capi import RP_OUTPUT, RP_SHARED, RpRange, int_to_limbs, points_to_array
    rng = (RpRange * 1)()
    rng[0].base = 0xFFFFFFFF; rng[0].flags = RP_OUTPUT | RP_SHARED
    rng[0].min[:] = [0, 0, 0, 0]; rng[0].max[:] = [int(v) for v in int_to_limbs(2**200)]
    arr = points_to_array(pts)
    hnd = C.c_void_p()
    rc = gpu.lib.bppp_rp_create(gpu.h, 0, 0, C.cast(rng, C.c_void_p), 1, None, 0, C.c_void_p(arr.ctypes.data), arr.shape[0], None, C.byref(hnd))
    assert rc != 0 and b"too large" in gpu.lib.bppp_last_error(gpu.h)


def test_device_prover_with_a_digit_base_above_256(gpu):
    """a shared range of base 1000 (linLen 6 + 999) next to a base-4 one: the device phases keep a 1024-entry reciprocal table per proof (round 4; bases above 256
    took the host-algebra route before) — same bytes as the host-algebra route and as the host protocol code, and the batch verifies"""
    rds = [RP.make_range_data(1000, 0, 10**6, True, True, False), RP.make_range_data(4, 0, 256, True, True, False)]
    pts = O.hash_points(b"base 1000", 2 + 6 + 999 + 3 + 8 + 8)
    st = RP.setup(RP.GpuBackend(gpu), pts, False, [], rds, "NL")
    assert st.lin_len == 6 + 999 + 3
    nat = RP.NativeRangeProofs(gpu, st)
    rnd = random.Random(1000)
    B = 4
    inputs = [[(rnd.randrange(10**6), 0, rnd.randrange(O.N)), (rnd.randrange(256), 0, rnd.randrange(O.N))] for _ in range(B)]
    inputs[0][0] = (999999, 0, 5); inputs[1][0] = (0, 0, 6); inputs[2][0] = (999 * 1000 + 999, 0, 7)
    prefixes = [b"base1000 %d" % b for b in range(B)]
    with _options(nat, comb_min=1, comb_bits=6):
        got = nat.prove_batch(inputs, prefixes)
        with _options(nat, host_algebra=1):
            assert nat.prove_batch(inputs, prefixes) == got
    proof = RP.prove(st, RP.witness(st, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0]))
    assert got[0] == E.encode_proof(4, proof)
    assert nat.verify_batch([c for c, _ in got], [p for _, p in got])
    nat.close()
