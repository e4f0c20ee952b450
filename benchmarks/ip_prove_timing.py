"""Lockstep prover of the inner-product flavour at the examples/64bit shape (1 x 64-bit value, base 16, nrmLen 16, linLen 6, 3 rounds):
python benchmarks/ip_prove_timing.py [batch ...]   (BPPP_RP_TIMING=1 for the phase times, BPPP_RP_HOST_ALGEBRA=1 for the host-core route)"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bulletproofspp_amd as b
from bulletproofspp_amd import rangeproof as RP
gpu = b.Bppp(0)
schema = json.load(open(os.path.join("tests", "golden", "examples", "64bit", "schema.json")))
st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
nat = RP.NativeRangeProofs(gpu, st)
rng = np.random.default_rng(0x1664)
vp = lambda a: C.c_void_p(a.ctypes.data)
for B in [int(a) for a in sys.argv[1:]] or [1 << 14]:
    vals = rng.integers(0, 2**64, size=B, dtype=np.uint64)
    amt = np.zeros((B, 1, 4), dtype=np.uint64); amt[:, 0, 0] = vals
    typ = np.zeros((B, 1, 4), dtype=np.uint64)
    bld = rng.integers(0, 2**64, size=(B, 1, 4), dtype=np.uint64); bld[:, :, 3] >>= np.uint64(1)
    pre = np.frombuffer(b"".join(b"bench ip %010d " % i for i in range(B)), dtype=np.uint8)
    cf = np.zeros(B * nat.shape["coms_bytes"], dtype=np.uint8); pf = np.zeros(B * nat.shape["proof_bytes"], dtype=np.uint8)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, B, vp(amt), vp(typ), vp(bld), vp(pre), 20, vp(cf), vp(pf)), "prove")
        ts.append(time.perf_counter() - t0)
    cb, pb = nat.shape["coms_bytes"], nat.shape["proof_bytes"]
    ok = nat.verify_batch([cf[i * cb:(i + 1) * cb].tobytes() for i in range(B)], [pf[i * pb:(i + 1) * pb].tobytes() for i in range(B)])
    print(f"B={B}: prove first {ts[0] * 1e3:.1f} ms, best {min(ts) * 1e3:.1f} ms ({B / min(ts):.0f} proofs/s), verify {ok}", flush=True)
nat.close()
