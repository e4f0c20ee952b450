import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bulletproofspp_amd as b
import bench
gpu = b.Bppp(0)
dev = torch.device("cuda", 0)
st, nat, count, typed, amount, rng = bench.make_rp_setup(gpu, torch, dev, 0, "64by64")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vals = rng.integers(0, 2**64, size=(B, count), dtype=np.uint64)
amt = np.zeros((B, count, 4), dtype=np.uint64); amt[:, :, 0] = vals
typ = np.zeros((B, count, 4), dtype=np.uint64)
bld = rng.integers(0, 2**64, size=(B, count, 4), dtype=np.uint64); bld[:, :, 3] >>= np.uint64(1)
pre = np.frombuffer(b"".join(b"timing %017d" % i for i in range(B)), dtype=np.uint8)
cf = np.zeros(B * nat.shape["coms_bytes"], dtype=np.uint8); pf = np.zeros(B * nat.shape["proof_bytes"], dtype=np.uint8)
vp = lambda a: C.c_void_p(a.ctypes.data)
for it in range(int(os.environ.get("PROVE_REPS", "2"))):
    t0 = time.perf_counter()
    gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, B, vp(amt), vp(typ), vp(bld), vp(pre), 24, vp(cf), vp(pf)), "prove")
    print("total ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
