"""N > 1 path on CPU: two gloo ranks shard one MSM by terms, all-gather one partial point each and add
them locally (bulletproofspp_amd/dist.py).  On the GPU box the same code runs with backend nccl (RCCL);
here the per-rank MSM and the point sum are the oracle's, so only the sharding / collective logic is under test."""
import os
import random
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import pyoracle as O
    from bulletproofspp_amd import dist as bd
    from bulletproofspp_amd.capi import points_to_array, array_to_point
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ec = O.CEC()
    rnd = random.Random(42)
    pts = O.hash_points(b"gloo", n)
    sc = [rnd.randrange(O.N) for _ in range(n)]

    def local_msm(lo, hi):
        return points_to_array([ec.inner_product(list(zip(sc[lo:hi], pts[lo:hi])))])[0]

    def sum_points(allp):
        acc = None
        for r in range(allp.shape[0]):
            acc = ec.add(acc, array_to_point(allp[r]))
        return points_to_array([acc])[0]

    total = bd.sharded_msm(n, rank, world, local_msm, sum_points, dist)
    want = ec.inner_product(list(zip(sc, pts)))
    # the non-blocking exchange used by bench.py for N > 1: several gathers in flight, results in issue order
    lo, hi = bd.shard_range(n, rank, world)
    part = local_msm(lo, hi)
    pend = [bd.all_gather_points_async(part, dist) for _ in range(3)]
    ok_async = all(array_to_point(sum_points(p.result())) == want for p in pend)
    q.put((rank, array_to_point(total) == want and ok_async, bd.shard_range(n, rank, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 64])
def test_sharded_msm_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    (lo0, hi0), (lo1, hi1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == n


def test_sharded_msm_world8_gloo():
    """the 8-rank exchange of BASELINE config 5 (8 x MI355X) on CPU cores: eight gloo ranks, one 64-byte point each, gathered blocking and
    non-blocking; every rank ends with the whole MSM"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    n, world = 43, 8
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r for r, _, _ in res] == list(range(world)) and all(ok for _, ok, _ in res)
    rs = [rg for _, _, rg in res]
    assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))


def test_bench_plain_start_eight_ranks_selftest():
    """`python bench.py --gpus 8` started plainly (no GPU: --launcher-selftest): eight children rendezvous, the 4096-proof job of config 5 is cut
    into 8 x 512"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BPPP_SELFTEST_FAIL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launcher-selftest"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["ranks"] == list(range(8)) and out["shards"] == [[512 * r, 512 * (r + 1)] for r in range(8)]


def test_shard_ranges_partition():
    from bulletproofspp_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1 << 20, 344838):
        for world in (1, 2, 4, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_bench_plain_start_launches_its_own_ranks():
    """`python bench.py --gpus N` started plainly is its own launcher (bench.py launch_ranks): N child processes with RANK / WORLD_SIZE /
    MASTER_* set rendezvous over gloo; rank 0's JSON line is relayed; a failing rank makes the launcher exit non-zero and ends the others.
    (--launcher-selftest keeps the ranks off the GPU; the GPU tier runs the real legs this way, tests/test_gpu_two_ranks.py.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BPPP_SELFTEST_FAIL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launcher-selftest"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 3 and out["ranks"] == [0, 1, 2]
    assert out["shards"][0][0] == 0 and out["shards"][-1][1] == 4096 and all(out["shards"][i][1] == out["shards"][i + 1][0] for i in range(2))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, BPPP_SELFTEST_FAIL_RANK="1"), cwd=ROOT)
    assert p.returncode != 0 and "rank 1 exited with 3" in p.stderr
