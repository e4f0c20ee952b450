"""Rate of the general batched MSM (bppp_msm_batch_device) at the shapes a per-proof basis would give the lockstep provers: `batch` instances of n terms, two
consecutive instances sharing a basis (the X and R rows of a proof).   python benchmarks/batch_msm_small.py [batch] [n ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bulletproofspp_amd as b
import bench
gpu = b.Bppp(0)
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for n in [int(a) for a in sys.argv[2:]] or [67, 131, 259, 515, 1027]:
    dsc, dpts = bench.make_inputs(gpu, torch, dev, batch * n, 11)          # batch * n scalars, batch * n points: the first batch / 2 * n points serve as bases
    out = np.zeros((batch, 8), dtype=np.uint64)
    for c in (0,):
        ts = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gpu._check(gpu.lib.bppp_msm_batch_device(gpu.h, C.c_void_p(dsc.data_ptr()), C.c_void_p(dpts.data_ptr()), n, batch, 2, c, C.c_void_p(out.ctypes.data)), "batch msm")
            ts.append(time.perf_counter() - t0)
        print(f"batch {batch} x n {n}: {min(ts) * 1e3:.3f} ms  ({batch * n / min(ts) / 1e6:.1f} M terms/s)", flush=True)
