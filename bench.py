#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Bulletproofs++ hot path.

Metric (BASELINE.json): MSM scalar-point pairs/sec at N = 2^20 (the Pedersen multi-scalar
multiplication every commit / verifyBPM call bottoms out in, src/Commitment.hs:325-335, :416-417).
A "step" is one complete MSM of 2^20 (scalar, affine point) pairs per GPU, inputs resident in
HBM, output = the canonical affine sum on the host.  With N > 1 GPUs (one process per GPU,
torch.distributed over RCCL) every rank owns its own 2^20-term slice of one N*2^20-term MSM (weak
scaling, no data-path collective); the only exchange is an all-gather of one 64-byte partial point
per rank, summed locally through the same library.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_acc_points) against the
HBM roofline with the ALGORITHMIC 96 bytes per pair (32-B scalar + 64-B affine point, SURVEY.md
§8d); `cpu_baseline` times the oracle's restatement of the reference's Straus loop on this host.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_PAIR = 96


def make_inputs(gpu, torch, dev, n: int, seed: int):
    """Synthetic MSM inputs generated on the GPU box: scalars uniform 256-bit (PCG64, documented seed),
    points = pointX-style lifts (app/Main.hs:68-72) of pseudo-random x, even y; one zero scalar and one
    point at infinity per 2^16 terms (dotWith's padding values, src/Commitment.hs:423-424)."""
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))  # < n (n's top limb is 0xFFFF...FFFF, next 0xFF..FE)
    got, chunks = 0, []
    while got < n:
        m = int((n - got) * 2.2) + 1024
        xs = rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64)
        dx = torch.from_numpy(xs.view(np.int64)).to(dev)
        dpts = torch.zeros((m, 8), dtype=torch.int64, device=dev)
        gpu.lift_x(dx.data_ptr(), m, dpts.data_ptr())
        ok = (dpts != 0).any(dim=1)
        good = dpts[ok]
        chunks.append(good)
        got += good.shape[0]
    pts = torch.cat(chunks)[:n].contiguous()
    for i in range(0, n, 1 << 16):
        sc[i + 1 if i + 1 < n else i] = 0
        if i + 2 < n:
            pts[i + 2] = 0
    dsc = torch.from_numpy(sc.view(np.int64)).to(dev)
    return dsc, pts


def cpu_baseline(sc_np: np.ndarray, pts_np: np.ndarray, sample: int):
    """Oracle restatement of the reference's 256-row Straus loop (oracle/bppp_oracle.c), single thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    ec = pyoracle.CEC()
    s = np.ascontiguousarray(sc_np[:sample])
    p = np.ascontiguousarray(pts_np[:sample])
    u64p = ctypes.POINTER(ctypes.c_uint64)
    t0 = time.perf_counter()
    res = ec.inner_product_raw(s.ctypes.data_as(u64p), p.ctypes.data_as(u64p), sample)
    dt = time.perf_counter() - t0
    return res, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--window", type=int, default=0, help="Pippenger window bits (0 = library heuristic)")
    ap.add_argument("--cpu-sample-log2", type=int, default=17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import bulletproofspp_amd as b
    from bulletproofspp_amd.capi import array_to_point, points_to_array

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = 1 << args.log2n
    gpu = b.Bppp(local)
    # one HIP stream for torch's copies/collectives and the library's kernels (ordering by stream, no extra syncs)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    gpu.set_stream(work_stream.cuda_stream)
    dsc, dpts = make_inputs(gpu, torch, dev, n, seed=0xB9B9 + rank)
    ones = torch.zeros((world, 4), dtype=torch.int64, device=dev)
    ones[:, 0] = 1
    gathered = torch.zeros((world, 8), dtype=torch.int64, device=dev)

    def step():
        part = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
        if world == 1:
            return part
        mine = torch.from_numpy(points_to_array([part]).view(np.int64)).to(dev)
        dist.all_gather_into_tensor(gathered, mine)          # 64 B per rank over xGMI; no mod-p reduce exists in RCCL
        return gpu.msm_device(ones.data_ptr(), gathered.data_ptr(), world, 0)

    for _ in range(args.warmup):
        res = step()
    # cross-check two different window decompositions of the same MSM (size-independent property)
    alt = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 13 if args.window != 13 else 12)
    ref_part = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
    assert alt == ref_part, "MSM results differ between window widths"

    gpu.profile_enable(True)
    gpu.profile_read(reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    stages, calls = gpu.profile_read(reset=True)
    gpu.profile_enable(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        per_call = {k: v / max(calls, 1) for k, v in stages.items()}
        # with N > 1 each step makes two library calls (the slice MSM and the tiny combine): the dominant
        # kernel's time is that of the big call; the combine adds ~0 to acc_points
        launches = args.steps
        acc_ms = stages["acc_points"] / launches
        achieved = BYTES_PER_PAIR * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_acc_points_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "msm_scalar_point_pairs_per_sec", "value": world * n * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (8x32-bit limbs, modular integer)", "data": "synthetic",
            "config": {"workload": f"pedersen_msm_2^{args.log2n}_secp256k1", "pairs_per_gpu": n, "window_bits": args.window or "auto",
                       "algorithm": "signed-digit Pippenger, affine in / XYZZ buckets", "sharding": f"terms/{world}" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_acc_points", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic 96 B/pair x 2^%d pairs per launch / mean k_acc_points duration (HIP events); "
                                 "the kernel is VALU-bound (256-bit modular multiplies), see DESIGN.md" % args.log2n},
            "stages_ms_per_step": {k: v * calls / launches for k, v in per_call.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = 1 << min(args.cpu_sample_log2, args.log2n)
            sc_np = dsc[:sample].cpu().numpy().view(np.uint64)
            pts_np = dpts[:sample].cpu().numpy().view(np.uint64)
            want, cdt = cpu_baseline(sc_np, pts_np, sample)
            got = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), sample, 0)
            assert got == want, "GPU MSM differs from the oracle on the CPU-baseline sample"
            out["cpu_baseline"] = {"value": sample / cdt, "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": f"first 2^{min(args.cpu_sample_log2, args.log2n)} pairs of the same workload, "
                                             "oracle/bppp_oracle.c restatement of the reference's 256-row Straus loop "
                                             "(Commitment.hs:325-335), single thread; the Haskell reference itself cannot be built "
                                             "here (no GHC)", "seconds": cdt, "gpu_matches": True}
        print(json.dumps(out), flush=True)
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
