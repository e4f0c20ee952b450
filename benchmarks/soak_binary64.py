"""Soak of the binary prover's routes at the 64 x 64-bit shape (4099 basis points): random batch sizes around the thresholds of the lane-per-instance comb kernel
(512 rows), of the re-based argument (384 proofs) and of the two half-batches in flight (1024 proofs); every batch is proved on the default route and again with
the argument never re-based and the batch never split — the files must be identical — and verified, one tampered member identified.
   python benchmarks/soak_binary64.py [seconds]"""
import os, random, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import pyoracle as O                      # only for the hashed basis points of the setup
import bulletproofspp_amd as b
from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as BRP
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
gpu = b.Bppp(0)
count, amount = 64, 10000
rds = [BRP.make_range_data(0, 2**64, True, False)] * count
pts = O.hash_points(b"binary soak", 4 + 64 * count)
st = BRP.setup(RP.GpuBackend(gpu), pts, True, rds, amount * count, "NL")
nat = BRP.NativeBinaryRangeProofs(gpu, st)
nat.set_option("comb_min", 1); nat.set_option("comb_bits", 10)
rnd = random.Random(64)
t_end, it = time.time() + secs, 0
while time.time() < t_end:
    B = rnd.choice([255, 256, 257, 383, 384, 385, 511, 512, 513, 1023, 1024, 1025]) if it % 2 else rnd.randrange(200, 1200)
    inputs = []
    for _ in range(B):
        d = [rnd.randrange(-5000, 5000) for _ in range(count // 2)]
        inputs.append([(amount + x, rnd.randrange(RP.N)) for x in d] + [(amount - x, rnd.randrange(RP.N)) for x in d])
    prefixes = [b"soak %06d %06d" % (it, j) for j in range(B)]
    os.environ.pop("BPPP_NLB_REBASE", None); nat.set_option("split_min", 4096)
    files = nat.prove_batch(inputs, prefixes)
    os.environ["BPPP_NLB_REBASE"] = "0"; nat.set_option("split_min", 0)
    assert nat.prove_batch(inputs, prefixes) == files, (it, B, "routes differ")
    seed = os.urandom(32)
    assert nat.verify_batch([c for c, _ in files], [p for _, p in files], seed), (it, B)
    j = rnd.randrange(B)
    bad = [list(f) for f in files]
    pf = bytearray(bad[j][1]); pf[rnd.randrange(32)] ^= 1 << rnd.randrange(8); bad[j][1] = bytes(pf)
    ok, status, _ = nat.verify_batch([c for c, _ in bad], [p for _, p in bad], seed, want_status=True)
    assert not ok and [i for i, s_ in enumerate(status) if s_] == [j], (it, B, j)
    it += 1
    print(f"batch {it}: B = {B} ok", flush=True)
nat.close()
print(f"soak ok: {it} batches")
