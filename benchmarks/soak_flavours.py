"""Soak of the inner-product flavour and of RangeProof.Binary through the native layer: random batch sizes of the reference's examples
(64bit, rec_test: IP; bin_test: binary) for a given number of seconds each; every batch must verify on both transcript-hashing routes,
and one tampered member must be identified.   python benchmarks/soak_flavours.py [seconds per example]"""
import json, os, random, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import bulletproofspp_amd as b
from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as BRP
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
gpu = b.Bppp(0)
EX = os.path.join(os.getcwd(), "tests", "golden", "examples")
rnd = random.Random(7)
for name in ("64bit", "rec_test", "bin_test"):
    schema = json.load(open(os.path.join(EX, name, "schema.json")))
    wit = json.load(open(os.path.join(EX, name, "witness.json")))
    binary = bool(schema.get("binary", False))
    if binary:
        st = BRP.setup_from_schema(RP.GpuBackend(gpu), schema); nat = BRP.NativeBinaryRangeProofs(gpu, st)
    else:
        st = RP.setup_from_schema(RP.GpuBackend(gpu), schema); nat = RP.NativeRangeProofs(gpu, st)
    t_end, it = time.time() + secs, 0
    # round 4: a second handle with the comb table forced from the first proof (the device-resident provers) must write the SAME BYTES as the
    # default handle, whose small batches take the host-algebra routes
    if binary:
        dev = BRP.NativeBinaryRangeProofs(gpu, st)
    else:
        dev = RP.NativeRangeProofs(gpu, st)
    dev.set_option("comb_min", 1); dev.set_option("comb_bits", 8)
    while time.time() < t_end:
        B = rnd.choice([1, 2, 3, 8, 9, 17, 64, 65, 130, 300, 1025]) if it % 2 else rnd.randrange(1, 200)
        inputs = []
        for j in range(B):
            rows = RP.inputs_from_witness(wit, b"soak %d %d" % (it, j))
            inputs.append([(v, bl) for v, _, bl in rows] if binary else rows)
        files = nat.prove_batch(inputs, [b"soak %06d %06d" % (it, j) for j in range(B)])
        dev.set_option("host_oracle_max", 0 if it % 2 == 0 else 2**64 - 1)
        assert dev.prove_batch(inputs, [b"soak %06d %06d" % (it, j) for j in range(B)]) == files, (name, it, B, "device-resident prover differs")
        seed = os.urandom(32)
        nat.set_option("host_oracle_max", 0 if it % 3 == 0 else 2**64 - 1)
        assert nat.verify_batch([c for c, _ in files], [p for _, p in files], seed), (name, it, B)
        j = rnd.randrange(B)
        bad = [list(f) for f in files]
        pf = bytearray(bad[j][1]); pf[rnd.randrange(32)] ^= 1 << rnd.randrange(8); bad[j][1] = bytes(pf)
        ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
        assert not ok and [i for i, s_ in enumerate(status) if s_] == [j], (name, it, B, j, status[:8])
        it += 1
    print(f"{name}: {it} batches ok", flush=True)
    nat.close(); dev.close()
print("soak ok")
