"""Time bppp_msm_device on one MSM of n random terms:  python benchmarks/msm_timing.py n   (under rocprofv3 --kernel-trace for a timeline)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bulletproofspp_amd as b
import bench
gpu = b.Bppp(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dsc, dpts = bench.make_inputs(gpu, torch, dev, n, 7)
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)
    print("msm ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
