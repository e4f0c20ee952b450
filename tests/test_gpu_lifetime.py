"""Handle lifetime across the C ABI: a child handle (bppp_nl / bppp_ip / bppp_nlb / bppp_trrp) may be destroyed AFTER its
context (a Haskell ForeignPtr finaliser or a Python __del__ at interpreter exit runs in any order).  The context is
reference-counted by its children (csrc/ctx.hpp), so that order neither touches freed memory nor leaks: run once, in a child
process, and require a clean exit."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import ctypes as C, random, sys
    sys.path.insert(0, %r); sys.path.insert(0, %r + "/oracle")
    import numpy as np
    import bulletproofspp_amd as b
    from bulletproofspp_amd.bulletproof import NormLinearBP, NormLinearIP, NormLinearBatch
    from bulletproofspp_amd import rangeproof as RP
    import pyoracle as O
    rnd = random.Random(5)
    pts = O.hash_points(b"lifetime", 40)
    r = lambda n: [rnd.randrange(O.N) for _ in range(n)]
    gpu = b.Bppp(0)
    lib = gpu.lib
    nl = NormLinearBP(gpu, 1, pts[0], 3, r(5), r(9), pts[1:10], r(5), pts[10:15])
    ip = NormLinearIP(gpu, 1, pts[0], 3, r(5), r(8), pts[1:9], r(5), pts[10:15])
    nb = NormLinearBatch(gpu, r(2), pts[0], r(2), [r(5), r(5)], [r(9), r(9)], pts[1:10], [r(5), r(5)], pts[10:15])
    rd = RP.make_range_data(16, 0, 2**64)
    st = RP.setup(RP.GpuBackend(gpu), pts, False, [], [rd])
    tabs = RP.DeviceVerifierTables(gpu, st)
    nl.makeScalarsComs()
    # the context goes FIRST, straight through the C ABI (not Bppp.close(), which would close the children for us)
    h, gpu.h = gpu.h, None
    lib.bppp_ctx_destroy(h)
    # a call on a child of a destroyed context fails cleanly
    sX = np.zeros(4, dtype=np.uint64); X = np.zeros(8, dtype=np.uint64)
    rc = lib.bppp_nl_round_commit(nl.h, sX.ctypes.data, X.ctypes.data, sX.ctypes.data, X.ctypes.data)
    assert rc == -1, rc
    # the children are destroyed afterwards; the last one tears the context down
    for ch in (nl, ip, nb, tabs):
        ch.close()
    print("lifetime ok")
''') % (ROOT, ROOT)


def test_context_destroyed_before_its_children_exits_cleanly():
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "lifetime ok" in p.stdout
