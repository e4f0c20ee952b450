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


def test_page_locked_host_buffers(gpu, oracle_lib):
    """bppp_host_alloc / bppp_host_free: a host-buffer entry point takes a page-locked buffer like any other (here the scalars and points
    of an MSM), and freeing one that was never allocated through the library is refused by the binding."""
    import numpy as np
    import pyoracle as O
    from bulletproofspp_amd.capi import points_to_array, scalars_to_array
    pts = O.hash_points(b"pinned", 33)
    sc = [(977 * i + 5) % O.N for i in range(33)]
    a, b_ = scalars_to_array(sc), points_to_array(pts)
    pa, pb = gpu.host_alloc(a.nbytes), gpu.host_alloc(b_.nbytes)
    pa[:] = a.view(np.uint8).reshape(-1); pb[:] = b_.view(np.uint8).reshape(-1)
    got = gpu.msm(pa.view(np.uint64).reshape(-1, 4), pb.view(np.uint64).reshape(-1, 8))
    assert got == oracle_lib.inner_product(list(zip(sc, pts)))
    gpu.host_free(pa); gpu.host_free(pb)
    try:
        gpu.host_free(np.zeros(8, dtype=np.uint8))
        assert False, "freeing a foreign array must be refused"
    except ValueError:
        pass
