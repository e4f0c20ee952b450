"""The routes the benchmark numbers are measured on, pinned against the ORACLE at production width (not only against each other):

  * the fixed-base comb at c = 16 (csrc/comb.hip; index arithmetic (w T D + mag - 1) over 17 windows x 32768 multiples per point) —
    the window the 64by64 prover's 27.6-GB table uses — on a basis small enough for the test (80 points: 2.85 GB), with one and with
    several wavefronts per instance;
  * the native lockstep prover (bppp_rp_prove_batch) at the FULL examples/64by64 and examples/128by64 + "typed" shapes, on its
    production route — comb table, argument without point folds, one stream of kernels, oracle on the device and on the host — byte
    for byte against bulletproofspp_amd.rangeproof.prove run over the oracle backend (tests/rp_backends.py: every commit the
    oracle's 256-row Straus restatement of src/Commitment.hs:325-335, the argument oracle/pyoracle.py's proveBPM with
    rationalReduceScalar and projectivePairIP as the reference has them, src/RangeProof/TypedReciprocal.hs:399-446)."""
import json
import os
import random

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd import encoding as E
from bulletproofspp_amd import rangeproof as RP
from bulletproofspp_amd.capi import points_to_array, scalars_to_array
from rp_backends import OracleBackend

pytestmark = pytest.mark.gpu


def test_comb_at_16_bits_equals_oracle(gpu, oracle_lib):
    n = 80
    pts = O.hash_points(b"comb c16", n)
    pts[5] = None
    pts[9] = pts[8]
    rnd = random.Random(1616)
    bas = gpu.basis(points_to_array(pts), batch_hint=64)
    cc, tb = bas.enable_comb(window_bits=16, budget_bytes=4 << 30)
    assert cc == 16 and tb == 17 * n * 32768 * 64
    half = (O.N - 1) // 2
    for batch in (70, 1100):                      # < 1024 instances: two wavefronts per instance + k_comb_join; >= 1024: one
        for n_terms in (n, 64, 65):
            sc = [[rnd.randrange(O.N) for _ in range(n_terms)] for _ in range(batch)]
            sc[0] = [0] * n_terms
            sc[1] = [s_ if (i >> 1) & 1 else 0 for i, s_ in enumerate(sc[1])]                     # sparse: the argument's R scalars
            sc[2] = [half, half + 1, half + 2, O.N - 1, 1, 2**16 - 1, 2**15, 2**15 + 1, 2**255 % O.N, O.N - 2**15] + sc[2][10:]   # sign fold / digit carry boundaries
            sc[3] = [rnd.randrange(256) for _ in range(n_terms)]                                  # range-proof digits: one non-zero window
            sc[4] = [(O.N - rnd.randrange(1, 2**16)) for _ in range(n_terms)]
            sc[5] = [((1 << 15) << (16 * (i % 16))) % O.N for i in range(n_terms)]                # a digit of exactly 2^15 in every window in turn
            sc[6] = [(((1 << 16) - 1) << (16 * (i % 16))) % O.N for i in range(n_terms)]
            d_s = gpu.to_device(np.concatenate([scalars_to_array(r) for r in sc]))
            try:
                got = bas.msm(d_s, n_terms, batch)
            finally:
                gpu.free(d_s)
            assert got[0] is None
            for b in (1, 2, 3, 4, 5, 6, 7, batch - 1):
                assert got[b] == oracle_lib.inner_product(list(zip(sc[b], pts[:n_terms]))), (batch, n_terms, b)
    bas.close()


def _schema(name, typed):
    from test_rangeproof import EXAMPLES
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    if typed:
        count = sum(int(r.get("count", 1)) for r in schema["ranges"])
        schema = dict(schema, typed=True, public=[{"amount": count * 10000, "type": 0}])
    return schema


@pytest.mark.parametrize("name,typed,shape", [("64by64", False, (512, 261, 8, (2, 2))), ("128by64", True, (1152, 261, 9, (3, 1)))])
def test_native_prover_full_size_equals_oracle_backend(gpu, oracle_lib, name, typed, shape):
    schema = _schema(name, typed)
    st_g = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    st_o = RP.setup_from_schema(OracleBackend(oracle_lib), schema)
    assert (st_g.nrm_len, st_g.lin_len, st_g.rounds, tuple(st_g.final_lens)) == shape and st_o.gs == st_g.gs
    count = len(st_g.rds)
    rnd = random.Random(name)
    B = 65

    def one_input(b):
        if typed:                                  # outputs of type 0 summing to the public input: random split around the example's amount
            d = [rnd.randrange(-5000, 5000) for _ in range(count // 2)]
            vals = [10000 + x for x in d] + [10000 - x for x in d]
        else:
            vals = [rnd.randrange(2**64) for _ in range(count)]
            if b == 0:
                vals[0], vals[1], vals[2] = 0, 2**64 - 1, 10000          # range ends, and examples/64by64/witness.json's amount
        return [(v, 0, rnd.randrange(O.N)) for v in vals]
    inputs = [one_input(b) for b in range(B)]
    prefixes = [b"full size %s %03d" % (name.encode(), b) for b in range(B)]
    # the reference's prover over the oracle's group operations, proof 0 (seconds of CPU: ~70 Straus commits + the 8-9 round argument)
    proof = RP.prove(st_o, RP.witness(st_o, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0]))
    want = E.encode_proof(4, proof)
    assert RP.verify(st_o, proof, RP.sha256_oracle())
    nat = RP.NativeRangeProofs(gpu, st_g)
    nat.set_option("comb_min", 1)                  # the production route (a handle takes it from its first 1024 proofs on)
    one = nat.prove_batch(inputs[:1], prefixes[:1])                       # 1 proof: oracle on the host, several wavefronts per comb instance
    assert one[0][0] == want[0], "commitments file differs from the oracle backend's"
    assert one[0][1] == want[1], "proof file differs from the oracle backend's"
    many = nat.prove_batch(inputs, prefixes)                              # 65 proofs: oracle on the device (> 64), multi-wavefront comb launches
    assert many[0] == want
    nat.set_option("host_oracle_max", 0)
    assert nat.prove_batch(inputs[:1], prefixes[:1])[0] == want           # 1 proof with the device oracle
    nat.set_option("host_oracle_max", 2**64 - 1)
    nat.set_option("fold_points", 1)
    assert nat.prove_batch(inputs[:2], prefixes[:2])[0] == want           # the point-folding argument over comb commitments
    nat.set_option("fold_points", 0)
    assert nat.verify_batch([c for c, _ in many], [p for _, p in many])
    nat.close()
    plain = RP.NativeRangeProofs(gpu, st_g)                               # no table: bucket MSMs + point folds
    plain.set_option("comb_budget", 0)
    assert plain.prove_batch(inputs[:1], prefixes[:1])[0] == want
    plain.close()


def test_lane_per_instance_comb_rows_write_the_same_files(monkeypatch, oracle_lib):
    """round 4: >= 1024 long rows of full-width scalars (the argument's X / R rows, the blinded phase rows) over a large comb table run with one LANE
    per instance (k_comb_msm_rows + k_comb_join_rows, csrc/comb.hip) — another schedule of the same sums.  examples/64by64, 1030 proofs (2060 round
    rows in heavy / light pairs, 1030 dense phase rows: neither a multiple of 64), on a context where every table qualifies (BPPP_COMB_ROWS_MIN_MB=0):
    proof 0 is the oracle backend's byte for byte, and the batch equals the one a context with the route switched off writes."""
    import bulletproofspp_amd as b
    schema = _schema("64by64", False)
    st_o = RP.setup_from_schema(OracleBackend(oracle_lib), schema)
    rnd = random.Random("rows")
    B, count = 1030, len(st_o.rds)
    inputs = [[(rnd.randrange(2**64), 0, rnd.randrange(O.N)) for _ in range(count)] for _ in range(B)]
    prefixes = [b"rows %04d" % i for i in range(B)]
    want = E.encode_proof(4, RP.prove(st_o, RP.witness(st_o, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0])))
    files = {}
    for tag, min_mb, waves in (("rows", "0", "0"), ("rows, few wavefronts", "0", "300"), ("off", "100000000", "0")):
        monkeypatch.setenv("BPPP_COMB_ROWS_MIN_MB", min_mb)
        monkeypatch.setenv("BPPP_COMB_ROWS_WAVES", waves)
        g = b.Bppp(0)
        nat = RP.NativeRangeProofs(g, RP.setup_from_schema(RP.GpuBackend(g), schema))
        nat.set_option("comb_min", 1); nat.set_option("comb_bits", 8)
        files[tag] = nat.prove_batch(inputs, prefixes)
        if tag == "rows":
            assert nat.verify_batch([c for c, _ in files[tag]], [p for _, p in files[tag]])
        nat.close()
    assert files["rows"][0] == want
    assert files["rows"] == files["off"] and files["rows, few wavefronts"] == files["off"]


def test_lane_per_instance_comb_rows_binary_shape(monkeypatch):
    """the same at the 64 x 64-bit BINARY shape (rows of 4099 terms; 1027 proofs: 2054 round rows, 1027 dense rows of the blinding commitment)"""
    import bulletproofspp_amd as b
    from bulletproofspp_amd import rangeproof_binary as BRP
    count, amount, B = 64, 10000, 1027
    rds = [BRP.make_range_data(0, 2**64, True, False)] * count
    pts = O.hash_points(b"binary rows", 4 + 64 * count)
    rnd = random.Random("binary rows")
    inputs = []
    for _ in range(B):
        d = [rnd.randrange(-5000, 5000) for _ in range(count // 2)]
        inputs.append([(amount + x, rnd.randrange(RP.N)) for x in d] + [(amount - x, rnd.randrange(RP.N)) for x in d])
    prefixes = [b"binrows %04d" % i for i in range(B)]
    files = {}
    for tag, min_mb in (("rows", "0"), ("off", "100000000")):
        monkeypatch.setenv("BPPP_COMB_ROWS_MIN_MB", min_mb)
        # "rows" also re-bases the argument after three folds (the default for a basis of 4099 points: the level-3 basis of every proof materialised by
        # comb_groups, the last seven rounds bucket MSMs over those 514 points, csrc/nlb.hip); "off" walks the table in every round
        monkeypatch.setenv("BPPP_NLB_REBASE", "0" if tag == "off" else "3")
        g = b.Bppp(0)
        nat = BRP.NativeBinaryRangeProofs(g, BRP.setup(RP.GpuBackend(g), pts, True, rds, amount * count, "NL"))
        nat.set_option("comb_min", 1); nat.set_option("comb_bits", 9)
        if tag == "off":
            nat.set_option("split_min", 0)          # one batch on one context; "rows" runs as two half-batches in flight (>= 1024 binary proofs)
        files[tag] = nat.prove_batch(inputs, prefixes)
        if tag == "rows":
            assert nat.verify_batch([c for c, _ in files[tag]], [p for _, p in files[tag]])
            bad = [list(r) for r in inputs]
            bad[1000][3] = (2**64, bad[1000][3][1])                  # in the SECOND half: the message names the proof's position in the batch
            with pytest.raises(Exception, match="proof 1000: value outside its range"):
                nat.prove_batch(bad, prefixes)
        nat.close()
    assert files["rows"] == files["off"]


@pytest.mark.parametrize("level", [1, 2, 5])
def test_rebased_argument_writes_the_oracle_backends_bytes(monkeypatch, gpu, oracle_lib, level):
    """the re-based lockstep argument (BPPP_NLB_REBASE=<level>: per-proof level basis by comb_groups, later rounds by bucket MSMs over it) on a shape with
    odd lengths on the way down (examples/64by64: nrmLen 512, linLen 261 -> groups of 2 / 4 / 32 points, the last ones short): byte for byte the proof of
    the reference's prover over the oracle backend, with the device and the host oracle"""
    monkeypatch.setenv("BPPP_NLB_REBASE", str(level))
    schema = _schema("64by64", False)
    st_g = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    st_o = RP.setup_from_schema(OracleBackend(oracle_lib), schema)
    rnd = random.Random(level)
    B, count = 5, len(st_g.rds)
    inputs = [[(rnd.randrange(2**64), 0, rnd.randrange(O.N)) for _ in range(count)] for _ in range(B)]
    prefixes = [b"rebase %d %d" % (level, i) for i in range(B)]
    want = E.encode_proof(4, RP.prove(st_o, RP.witness(st_o, inputs[0]), RP.sha256_oracle(), RP.hash_to_scalar(prefixes[0])))
    nat = RP.NativeRangeProofs(gpu, st_g)
    nat.set_option("comb_min", 1); nat.set_option("comb_bits", 8)
    for host_oracle_max in (0, 2**64 - 1):
        nat.set_option("host_oracle_max", host_oracle_max)
        files = nat.prove_batch(inputs, prefixes)
        assert files[0] == want
        assert nat.verify_batch([c for c, _ in files], [p for _, p in files])
    nat.close()
