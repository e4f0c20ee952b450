"""MSM at 2^20 through the endomorphism decomposition (bppp_msm_glv_device) against the plain route, same inputs."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
import bulletproofspp_amd as b
sys.argv = ['x']
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
g = b.Bppp(0)
n = 1 << 20
dsc, dpts = bench.make_inputs(g, torch, dev, n, 0xB9B9)
r0 = g.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)
r1 = g.msm_glv_device(dsc.data_ptr(), dpts.data_ptr(), n)
assert r0 == r1
for name, fn in (("plain", lambda: g.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)), ("glv", lambda: g.msm_glv_device(dsc.data_ptr(), dpts.data_ptr(), n))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms")
