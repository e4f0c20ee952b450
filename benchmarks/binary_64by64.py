"""RangeProof.Binary at the 64 x 64-bit shape (BASELINE config 3 read literally: "64x64-bit aggregated binary range proof"): 64 outputs in
[0, 2^64), one bit per norm position (nrmLen 4096, linLen 2), norm-linear argument, through the library's own lockstep prover and batch
verifier.   python benchmarks/binary_64by64.py [batch ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bulletproofspp_amd as b
from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as RB
from bulletproofspp_amd.capi import array_to_point
gpu = b.Bppp(0)
dev = torch.device("cuda", 0)
count = 64
rds = [RB.make_range_data(0, 2**64, True, False)] * count
need = 4 + sum(len(rd.base_coeffs) for rd in rds)
rng = np.random.default_rng(0xB1)
pts = None
while pts is None or pts.shape[0] < need:
    xs = rng.integers(0, 2**64, size=(3 * need, 4), dtype=np.uint64)
    dx = torch.from_numpy(xs.view(np.int64)).to(dev)
    dp = torch.zeros((3 * need, 8), dtype=torch.int64, device=dev)
    gpu.lift_x(dx.data_ptr(), 3 * need, dp.data_ptr())
    pts = dp[(dp != 0).any(dim=1)]
P = pts[:need].cpu().numpy().view(np.uint64)
basis = [array_to_point(P[i]) for i in range(need)]
amount = 10000                                   # examples/*/witness.json; one public input balances the 64 outputs (witnessBRP needs a conserved schema)
st = RB.setup(RP.GpuBackend(gpu), basis, True, rds, amount * count, "NL")
nat = RB.NativeBinaryRangeProofs(gpu, st, h=basis[0])
print("shape", nat.shape, flush=True)
for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
    dlt = rng.integers(-5000, 5000, size=(B, count // 2))
    vals = np.concatenate([amount + dlt, amount - dlt], axis=1).astype(np.uint64)
    bld = rng.integers(1, 2**63, size=(B, count), dtype=np.uint64)
    inputs = [[(int(v), int(x)) for v, x in zip(vals[i], bld[i])] for i in range(B)]
    cache = os.environ.get("PROOF_FILES")           # profiling passes of the VERIFIER: proofs from a file made by an earlier run
    if cache and os.path.exists(cache):
        z = np.load(cache); cf, pf = z["cf"], z["pf"]; tp = float("nan")
    else:
        t0 = time.perf_counter()
        files = nat.prove_batch(inputs, [b"bin64 %08d" % i for i in range(B)])
        tp = time.perf_counter() - t0
        cf = np.frombuffer(b"".join(c for c, _ in files), dtype=np.uint8); pf = np.frombuffer(b"".join(p for _, p in files), dtype=np.uint8)
        if cache:
            np.savez(cache, cf=cf, pf=pf)
    dc, dpf = gpu.to_device(cf), gpu.to_device(pf)
    seed = os.urandom(32)
    ok = nat.verify_batch_device(B, dc, dpf, seed)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ok = ok and nat.verify_batch_device(B, dc, dpf, seed); ts.append(time.perf_counter() - t0)
    if not cache:                                   # (profiling passes skip the rejection check: its bisection would add verifier calls to the trace)
        pf2 = pf.copy(); pf2[(B // 2) * nat.shape["proof_bytes"] + 9] ^= 4
        dp2 = gpu.to_device(pf2)
        ok2, status, _ = nat.verify_batch_device(B, dc, dp2, seed, want_status=True)
        assert ok and not ok2 and [i for i, s_ in enumerate(status) if s_] == [B // 2], (ok, ok2)
        gpu.free(dp2)
    assert ok
    gpu.free(dc); gpu.free(dpf)
    print(f"B={B}: prove {tp * 1e3:.1f} ms ({B / tp:.0f} proofs/s), verify {min(ts) * 1e3:.3f} ms ({B / min(ts):.0f} verifies/s), proof {nat.shape['proof_bytes']} B + coms {nat.shape['coms_bytes']} B", flush=True)
nat.close()
