"""Garbage in, reject out: the end-to-end batch verifier on files that are not proofs — uniform random bytes, all 0x00, all 0xFF, honest files with random
byte ranges overwritten, truncated to a wrong length — for the typed-reciprocal (both flavours) and the binary handle, both hashing routes, random batch sizes.
Every call must return (reject, per-proof status) or refuse the call; an honest member of a mixed batch must keep status 0.
   python benchmarks/fuzz_verify.py [seconds per setup]"""
import json, os, random, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bulletproofspp_amd as b
from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as BRP
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
gpu = b.Bppp(0)
EX = os.path.join(os.getcwd(), "tests", "golden", "examples")
rnd = random.Random(99)
for name in ("64bit", "rec_test", "bin_test", "32by64"):
    schema = json.load(open(os.path.join(EX, name, "schema.json")))
    wit = json.load(open(os.path.join(EX, name, "witness.json")))
    binary = bool(schema.get("binary", False))
    st = (BRP if binary else RP).setup_from_schema(RP.GpuBackend(gpu), schema)
    nat = (BRP.NativeBinaryRangeProofs if binary else RP.NativeRangeProofs)(gpu, st)
    rows = RP.inputs_from_witness(wit, b"fuzz")
    good = nat.prove_batch([[(v, bl) for v, _, bl in rows] if binary else rows] * 4, [b"fuzz %02d" % j for j in range(4)])
    cb, pb = nat.shape["coms_bytes"], nat.shape["proof_bytes"]
    t_end, it, rejected = time.time() + secs, 0, 0
    while time.time() < t_end:
        B = rnd.choice([1, 2, 7, 8, 9, 33, 64, 65, 200])
        coms, prfs, honest = [], [], []
        for j in range(B):
            kind = rnd.randrange(6)
            c, p = good[j % 4]
            if kind == 0: c, p = os.urandom(cb), os.urandom(pb)
            elif kind == 1: c, p = bytes(cb), bytes(pb)
            elif kind == 2: c, p = b"\xff" * cb, b"\xff" * pb
            elif kind == 3:
                q = bytearray(p); lo = rnd.randrange(pb); q[lo:lo + rnd.randrange(1, 40)] = os.urandom(min(39, pb - lo))[:len(q[lo:lo + 39])]; p = bytes(q[:pb])
            elif kind == 4:
                q = bytearray(c); lo = rnd.randrange(cb); q[lo] ^= 1 << rnd.randrange(8); c = bytes(q)
            honest.append(kind == 5)
            coms.append(c); prfs.append(p)
        nat.set_option("host_oracle_max", 0 if it % 2 else 2**64 - 1)
        ok, status, _ = nat.verify_batch(coms, prfs, os.urandom(32), want_status=True)
        assert ok == all(s_ == 0 for s_ in status), (name, it)
        assert all(status[j] == 0 for j in range(B) if honest[j]), (name, it, "an honest member was blamed")
        rejected += sum(1 for s_ in status if s_)
        it += 1
    # a wrong file length is refused, not read past
    assert nat.verify_batch([good[0][0]], [good[0][1][:-1]]) is False
    print(f"{name}: {it} batches, {rejected} members rejected, no honest member blamed", flush=True)
    nat.close()
print("fuzz ok")
