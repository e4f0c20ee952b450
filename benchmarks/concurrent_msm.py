import sys, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bulletproofspp_amd as b
sys.argv=['x']
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
g0 = b.Bppp(0)
n = 1 << 20
dsc, dpts = bench.make_inputs(g0, torch, dev, n, 0xB9B9)
torch.cuda.synchronize()
for T in (1, 2, 3, 4):
    ctxs = [b.Bppp(0) for _ in range(T)]
    for c in ctxs: c.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 16)
    K = 12
    res = [None]*T
    def work(i):
        for _ in range(K): res[i] = ctxs[i].msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 16)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    assert all(r == res[0] for r in res)
    print(f"T={T}: {dt/(K*T)*1e3:.3f} ms per MSM, {n*K*T/dt/1e6:.1f} M pairs/s")
    for c in ctxs: c.close()
