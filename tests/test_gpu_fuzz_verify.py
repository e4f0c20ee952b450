"""Garbage in, reject out (the short form of benchmarks/fuzz_verify.py): files that are not proofs — random bytes, all 0x00 / 0xFF, honest files with byte ranges
overwritten or a commitment bit flipped — mixed with honest ones through bppp_rp_verify_batch on a typed-reciprocal inner-product setup and a binary one, both
hashing routes.  Every call returns reject + per-proof status, an honest member is never blamed, a wrong file length is refused."""
import json
import os
import random

import pytest

from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as BRP
from test_rangeproof import EXAMPLES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["64bit", "bin_test"])
def test_garbage_files_are_rejected_and_honest_members_kept(gpu, name):
    schema = json.load(open(os.path.join(EXAMPLES, name, "schema.json")))
    wit = json.load(open(os.path.join(EXAMPLES, name, "witness.json")))
    binary = bool(schema.get("binary", False))
    st = (BRP if binary else RP).setup_from_schema(RP.GpuBackend(gpu), schema)
    nat = (BRP.NativeBinaryRangeProofs if binary else RP.NativeRangeProofs)(gpu, st)
    rows = RP.inputs_from_witness(wit, b"fuzz")
    good = nat.prove_batch([[(v, bl) for v, _, bl in rows] if binary else rows] * 4, [b"fuzz %02d" % j for j in range(4)])
    cb, pb = nat.shape["coms_bytes"], nat.shape["proof_bytes"]
    rnd = random.Random(name)
    rejected = 0
    for it in range(60):
        B = rnd.choice([1, 2, 7, 8, 9, 33, 65])
        coms, prfs, honest = [], [], []
        for j in range(B):
            kind = rnd.randrange(6)
            c, p = good[j % 4]
            if kind == 0:
                c, p = bytes(rnd.getrandbits(8) for _ in range(cb)), bytes(rnd.getrandbits(8) for _ in range(pb))
            elif kind == 1:
                c, p = bytes(cb), bytes(pb)
            elif kind == 2:
                c, p = b"\xff" * cb, b"\xff" * pb
            elif kind == 3:
                q = bytearray(p); lo = rnd.randrange(pb)
                for i in range(lo, min(pb, lo + rnd.randrange(1, 40))):
                    q[i] = rnd.getrandbits(8)
                p = bytes(q)
            elif kind == 4:
                q = bytearray(c); q[rnd.randrange(cb)] ^= 1 << rnd.randrange(8); c = bytes(q)
            honest.append(kind == 5)
            coms.append(c); prfs.append(p)
        nat.set_option("host_oracle_max", 0 if it % 2 else 2**64 - 1)
        ok, status, _ = nat.verify_batch(coms, prfs, bytes(rnd.getrandbits(8) for _ in range(32)), want_status=True)
        assert ok == all(s_ == 0 for s_ in status)
        assert all(status[j] == 0 for j in range(B) if honest[j]), "an honest member was blamed (batch %d)" % it
        rejected += sum(1 for s_ in status if s_)
    assert rejected > 300
    assert nat.verify_batch([good[0][0]], [good[0][1][:-1]]) is False
    nat.close()
