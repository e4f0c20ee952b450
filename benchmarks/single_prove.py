"""One 64by64-shaped norm-linear argument at a time (bppp_nl_*): wall time per proof and per call."""
import sys, time, random
sys.path.insert(0, '/root/repo')
import numpy as np
import bulletproofspp_amd as b
from bulletproofspp_amd.bulletproof import NormLinearBP, proveBPM, N_ORDER
from bulletproofspp_amd import rangeproof as RP
g = b.Bppp(0)
pts = RP.basis_points(b"single", 1 + 512 + 261)
rnd = random.Random(1)
r = lambda n: [rnd.randrange(N_ORDER) for _ in range(n)]
orc = RP.sha256_oracle()
def one():
    com = NormLinearBP(g, rnd.randrange(N_ORDER), pts[0], rnd.randrange(N_ORDER), r(261), r(512), pts[1:513], r(261), pts[513:])
    t0 = time.perf_counter()
    tc = tl = 0.0
    tr = RP.Transcript(orc)
    for _ in range(8):
        t1 = time.perf_counter(); _, X, _, R = com.makeScalarsComs(); t2 = time.perf_counter()
        e = tr.oracle([X, R], 1)[0]; t3 = time.perf_counter()
        com.collapse(e); t4 = time.perf_counter()
        tc += t2 - t1; tl += t4 - t3
    com.getWitness()
    dt = time.perf_counter() - t0
    com.close()
    return dt, tc, tl
one()
res = [one() for _ in range(5)]
print("ms per proof %.2f  (round_commit %.2f, round_collapse %.2f)" % tuple(1e3 * sum(x[i] for x in res) / len(res) for i in range(3)))
