"""Soak: random batch sizes through bppp_rp_prove_batch / bppp_rp_verify_batch_device on the 64by64 setup for a given number of seconds;
every batch must verify, and with one byte of one proof flipped the batch must be rejected with that proof identified.
   python benchmarks/soak_rp.py [seconds] [shape: 64by64 | 128by64+typed]"""
import os, sys, time, random, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bulletproofspp_amd as b
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
gpu = b.Bppp(0)
dev = torch.device("cuda", 0)
shape = sys.argv[2] if len(sys.argv) > 2 else "64by64"
st, nat, count, typed, amount, rng = bench.make_rp_setup(gpu, torch, dev, 0, shape)
rnd = random.Random(1)
cb, pb = nat.shape["coms_bytes"], nat.shape["proof_bytes"]
vp = lambda a: C.c_void_p(a.ctypes.data)
t_end = time.time() + secs
it = 0
sizes = [1, 2, 7, 8, 9, 33, 64, 65, 100, 257, 1000, 1024, 2048, 4095, 4096, 5000] if shape == "64by64" else [1, 2, 7, 8, 9, 33, 64, 65, 100, 257, 1000, 1024, 2048, 2500]
while time.time() < t_end:
    B = rnd.choice(sizes) if it % 3 else rnd.randrange(1, 600)
    if typed:                              # conserved: random splits around the example's amount that keep the sum
        dlt = rng.integers(-5000, 5000, size=(B, count // 2))
        vals = np.concatenate([amount + dlt, amount - dlt], axis=1).astype(np.uint64)
    else:
        vals = rng.integers(0, 2**64, size=(B, count), dtype=np.uint64)
    amt = np.zeros((B, count, 4), dtype=np.uint64); amt[:, :, 0] = vals
    typ = np.zeros((B, count, 4), dtype=np.uint64)
    bld = rng.integers(0, 2**64, size=(B, count, 4), dtype=np.uint64); bld[:, :, 3] >>= np.uint64(1)
    pre = np.frombuffer(b"".join(b"soak %04d %014d" % (it, i) for i in range(B)), dtype=np.uint8)
    cf = np.zeros(B * cb, dtype=np.uint8); pf = np.zeros(B * pb, dtype=np.uint8)
    gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, B, vp(amt), vp(typ), vp(bld), vp(pre), 24, vp(cf), vp(pf)), "prove")
    seed = bytes([it & 255]) * 32
    dc, dp = gpu.to_device(cf), gpu.to_device(pf)
    ok = nat.verify_batch_device(B, dc, dp, seed)
    assert ok, ("honest batch rejected", it, B)
    j = rnd.randrange(B)
    pf2 = pf.copy(); pf2[j * pb + 5] ^= 0x10
    dp2 = gpu.to_device(pf2)
    ok2, status, _ = nat.verify_batch_device(B, dc, dp2, seed, want_status=True)
    assert not ok2 and [i for i, s_ in enumerate(status) if s_] == [j], ("tampered proof not identified", it, B, j)
    gpu.free(dc); gpu.free(dp); gpu.free(dp2)
    it += 1
    if it % 10 == 0: print("iterations", it, "last B", B, flush=True)
print("soak ok:", it, "batches")
