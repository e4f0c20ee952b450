"""Correctness sweep of the automatic MSM plan over sizes around every threshold of choose_window / the slice rule and random ones: the automatic window must give the same point as a forced one."""
import os, sys, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from bulletproofspp_amd.capi import Bppp
gpu = Bppp(0); dev = torch.device("cuda:0")
nmax = (1 << 20) + 777
dsc, dpts = bench.make_inputs(gpu, torch, dev, nmax, 11)
rnd = random.Random(5)
sizes = [4095, 4096, 4097, 12287, 12288, 45999, 46000, 199999, 200000, 65535, 65537, 131071, 262145, 524287, nmax] + [rnd.randrange(4096, nmax) for _ in range(12)]
for n in sizes:
    a = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)
    b = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 12 if n < 300000 else 15)
    assert a == b, n
    print(n, "ok", flush=True)
print("all ok")
