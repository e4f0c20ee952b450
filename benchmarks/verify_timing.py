"""Time bppp_rp_verify_batch_device on one batch of distinct 64by64 proofs (made by the library's own prover).
   python benchmarks/verify_timing.py [batch]     (under rocprofv3 --kernel-trace for a kernel timeline)"""
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
cache = os.environ.get("PROOF_FILES")               # profiling passes: the proofs come from a file made by an earlier run, so that only the verifier's kernels are in the trace
if cache and os.path.exists(cache):
    z = np.load(cache); cf, pf = z["cf"], z["pf"]
    assert cf.size == B * nat.shape["coms_bytes"] and pf.size == B * nat.shape["proof_bytes"]
else:
    gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, B, vp(amt), vp(typ), vp(bld), vp(pre), 24, vp(cf), vp(pf)), "prove")
    if cache:
        np.savez(cache, cf=cf, pf=pf)
dc = torch.from_numpy(cf).to(dev); dp = torch.from_numpy(pf).to(dev)
seed = np.frombuffer(b"\x05" * 32, dtype=np.uint8)
acc = C.c_int(0)
hostbuf = int(os.environ.get("HOSTBUF", "0"))        # 1: bppp_rp_verify_batch from pageable host arrays, 2: from page-locked ones (bppp_host_alloc)
if hostbuf == 2:
    pcf, ppf = gpu.host_alloc(cf.nbytes), gpu.host_alloc(pf.nbytes)
    pcf[:] = cf; ppf[:] = pf
    cf, pf = pcf, ppf
for it in range(int(os.environ.get("VERIFY_REPS", "4"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if hostbuf:
        gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, B, vp(cf), vp(pf), vp(seed), C.byref(acc), None, None, None), "verify")
    else:
        gpu._check(gpu.lib.bppp_rp_verify_batch_device(nat.h, B, C.c_void_p(dc.data_ptr()), C.c_void_p(dp.data_ptr()), vp(seed), C.byref(acc), None, None, None), "verify")
    print("verify ms", (time.perf_counter() - t0) * 1e3, "accept", acc.value, file=sys.stderr)
