#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Bulletproofs++ hot path.

Metric (BASELINE.json): MSM scalar-point pairs/sec at N = 2^20 (the Pedersen multi-scalar
multiplication every commit / verifyBPM call bottoms out in, src/Commitment.hs:325-335, :416-417).
A "step" is one complete MSM of 2^20 (scalar, affine point) pairs per GPU, inputs resident in
HBM, output = the canonical affine sum on the host.  With N > 1 GPUs (one process per GPU,
torch.distributed over RCCL) every rank owns its own 2^20-term slice of one N*2^20-term MSM (weak
scaling, no data-path collective); the only exchange is an all-gather of one 64-byte partial point
per rank, summed locally through the same library.  `python bench.py --gpus N` started plainly
launches its N ranks itself (launch_ranks: child processes made before torch or HIP are touched);
under `python -m torch.distributed.run` the ranks are the launcher's.  For N > 1 the line also holds
the STRONG-scaling figures: `msm_strong_scaling` (one 2^20-term MSM over N ranks) and the verify
legs as BASELINE.json states configs 4 / 5 (one job of proofs split over the ranks).

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


def cpu_baseline_threads(sc_np: np.ndarray, pts_np: np.ndarray, per_thread: int):
    """The same restatement on every host core at once: one slice of `per_thread` pairs per thread
    (the ctypes call releases the GIL); the partial points are not combined — this is a rate, the single-thread leg is the checker."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
    cores = min(cores, sc_np.shape[0] // per_thread)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    slices = [(np.ascontiguousarray(sc_np[i * per_thread:(i + 1) * per_thread]), np.ascontiguousarray(pts_np[i * per_thread:(i + 1) * per_thread]))
              for i in range(cores)]
    ecs = [pyoracle.CEC() for _ in range(cores)]
    th = [threading.Thread(target=lambda i=i: ecs[i].inner_product_raw(slices[i][0].ctypes.data_as(u64p), slices[i][1].ctypes.data_as(u64p), per_thread))
          for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return cores * per_thread / dt, cores, dt


def profile_table(key: str):
    """profiles/traffic.json[key]: the per-kernel counter table of one verifier call (rocprofv3 passes of benchmarks/profile_round.sh verifypmc)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:
        return None


def prover_rows_roofline():
    """k_comb_msm_rows, the norm-linear prover's dominant kernel (lane = instance; csrc/comb.hip), from the committed profiles of
    `BPPP_RP_NO_SPLIT=1 python benchmarks/prove_timing.py 4096` (benchmarks/profile_round.sh: a --kernel-trace --stats pass for the mean duration,
    separate --pmc passes for FETCH_SIZE / WRITE_SIZE / SQ_*): 16 round launches of 4096 X rows (774 terms) + 4096 R rows (387 non-zero terms) and 4 launches of
    4096 dense phase rows; algorithmic bytes per term: its 32-B scalar and one 64-B table row for each of the 16 non-zero windows (c = 16)."""
    import csv
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["prover_4096_proofs_64by64"]["by_kernel"]
        key = next(k for k in tab if "k_comb_msm_rows" in k)
        v = tab[key]
        rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_prove_4096_kernel_stats.csv"))))
        ms = next(float(r["AverageNs"]) / 1e6 for r in rows if "k_comb_msm_rows" in r["Name"])
    except Exception:
        return None
    terms = (16 * 4096 * (774 + 387) + 4 * 4096 * 774) / 20.0
    alg = terms * (32 + 16 * 64)
    sq = v["sq_per_launch"]
    return {"bound": "hbm", "kernel": "k_comb_msm_rows", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"], "fetch_bytes_raw": v["fetch_bytes_per_launch_raw"], "ms_per_launch": ms,
            "valu_busy_est": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024 * ms * 1e-3 * 2.4e9), "wait_inst_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
            "valu_insts_per_addition": sq["SQ_INSTS_VALU"] * 64 / (terms * 16),
            "traffic_source": "static: profiles/traffic.json + profiles/r04_prove_4096_kernel_stats.csv (rocprofv3 passes of benchmarks/profile_round.sh over the one-context "
                              "prover, 20 launches per batch), not measured in this run; the gathers are 64-B requests, for which gfx950's x2 on FETCH_SIZE overstates — the raw "
                              "count equals the algorithmic bytes",
            "limiter": "valu",
            "note": "mean over the 20 launches of a 4096-proof batch: %.2f M terms x (32-B scalar + 16 windows x 64-B table row) per launch / mean duration; the kernel is "
                    "VALU-bound (one mixed addition per table row: ~1.9 k VALU instructions, 0.74 VALU-busy)" % (terms / 1e6)}


def stage_rows(tab, top: int = 8):
    """the largest kernels of a profiled verifier call: ms, HBM bytes (FETCH x2 + WRITE), wait fraction, VALU-busy estimate"""
    if not tab:
        return None
    rows = sorted(tab["by_kernel"].items(), key=lambda kv: -kv[1]["ms_per_call"])[:top]
    return [{"kernel": k.replace("bppp::", "").replace("void ", ""), "ms": round(v["ms_per_call"], 4), "hbm_bytes": int(v["fetch_bytes_x2"] + v["write_bytes"]),
             "wait_inst_frac": v.get("wait_inst_frac"), "valu_busy_est": v.get("valu_busy_est")} for k, v in rows]


def live_msm_traffic(log2n: int, timeout_s: float = 90.0):
    """HBM bytes of one k_acc_points launch measured IN THIS RUN: two child processes of this script under rocprofv3 --pmc (FETCH_SIZE, then
    WRITE_SIZE: separate passes, counters only, the program directly after `--`), headline leg only.  gfx950 corrections as the guide's HBM
    section prescribes: both counters in KB, FETCH_SIZE counts a 128-B request as 64 B (x2).  Returns (bytes, source) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bppp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--headline-only", "--no-cpu-baseline",
               "--steps", "4", "--warmup", "2", "--log2n", str(log2n)]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"))
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (counter, p.returncode)
            tot, cnt = 0.0, 0
            for dp_, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        with open(os.path.join(dp_, f)) as fh:
                            for r in csv.DictReader(fh):
                                if r["Counter_Name"] == counter and "k_acc_points" in r["Kernel_Name"]:
                                    tot += float(r["Counter_Value"]); cnt += 1
            if not cnt:
                return None, "no k_acc_points rows in the %s pass" % counter
            vals[counter] = tot / cnt * 1024.0
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --pmc %s timed out" % counter
        except Exception as e:
            return None, "rocprofv3 --pmc %s: %s" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of `bench.py --headline-only` "
                                                          "(mean per k_acc_points launch; FETCH_SIZE %.0f B raw x 2 + WRITE_SIZE %.0f B)" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]))


SHAPES = {   # (nrmLen, linLen, rounds, final norm, final lin, transcript commitments)
    "64by64": (512, 261, 8, 2, 2, 68),              # examples/64by64: 64 values, base 256 shared, NL argument
    "128by64+typed": (1152, 261, 9, 3, 1, 132),     # examples/128by64 with "typed": true (BASELINE config 4)
}


def make_rp_setup(gpu, torch, dev, rank: int, shape: str):
    """One typed-reciprocal setup of an examples/ shape over a synthetic basis (points lifted on the GPU, layout h : g : hs ++ gs,
    TypedReciprocal.hs:334, :348-349), registered with the library (bppp_rp_create)."""
    from bulletproofspp_amd import rangeproof as RP
    from bulletproofspp_amd.capi import array_to_point
    nlen, llen, k, fn, fl, ninit = SHAPES[shape]
    count, typed = (64, False) if shape == "64by64" else (128, True)
    rng = np.random.default_rng(0x64B)            # ONE setup for every rank (a sharded job is verified against the same basis everywhere);
    need = 2 + llen + nlen                          # the ranks' witnesses and randomness differ (the generator returned below)
    pts = None
    while pts is None or pts.shape[0] < need:
        xs = rng.integers(0, 2**64, size=(3 * need, 4), dtype=np.uint64)
        dx = torch.from_numpy(xs.view(np.int64)).to(dev)
        dp = torch.zeros((3 * need, 8), dtype=torch.int64, device=dev)
        gpu.lift_x(dx.data_ptr(), 3 * need, dp.data_ptr())
        pts = dp[(dp != 0).any(dim=1)]
    P = pts[:need].cpu().numpy().view(np.uint64)
    basis_pts = [array_to_point(P[i]) for i in range(need)]
    rd = RP.make_range_data(256, 0, 2**64, True, True, False)
    amount = 10000                                                     # examples/*/witness.json
    pub_vt = [(False, 0, amount * count)] if typed else []             # conservation: one public input balances the outputs
    st = RP.setup(RP.GpuBackend(gpu), basis_pts, typed, pub_vt, [rd] * count)
    assert (st.nrm_len, st.lin_len, st.rounds, st.final_lens) == (nlen, llen, k, (fn, fl)), "shape table out of date"
    nat = RP.NativeRangeProofs(gpu, st, h=basis_pts[0])
    return st, nat, count, typed, amount, np.random.default_rng(0x64B0 + rank)


def bench_rangeproofs(gpu, torch, dev, rank, world, dist, combine, batch: int, steps: int, warmup: int, shape: str = "64by64",
                      cpu_baseline_leg: bool = False, prove_steps: int = 1, coll_dev=None):
    """Both range-proof legs on `batch` DISTINCT real proofs of an examples/ shape per GPU, through the library's end-to-end entry points.

    prove  (bppp_rp_prove_batch): proveM of RangeProof (src/RangeProof.hs:93-97) for the whole batch in lockstep — phases 1-3 of
           proveTRRPM, every commitment on the GPU, the norm-linear argument, transcript hashing (shaOracle) and encodeProof';
           inputs are host arrays (values, blindings), outputs the reference's files.  N > 1: independent replicas.
    verify (bppp_rp_verify_batch_device): the metric "aggregated 64-bit range-proof verifies/sec".  Timed with the ENCODED proofs
           (commitments file + proof file per proof, as the reference writes them) resident in HBM: decodeProof (square roots, sign
           selection), all SHA-256 transcript hashing of verifyTRRPM and verifyBPM, the public scalars, challenge expansion,
           shared-basis merge and ONE combined MSM.  Nothing of a verification is left outside the timed region.  N > 1: proofs are
           sharded per GPU; the ranks all-gather their 64-byte combined points and add them (the only exchange).  Two figures for
           N > 1: STRONG scaling — the configuration BASELINE.json states (config 5: ONE job of `batch` proofs, batch / N per rank;
           config 4: one job of 128by64+typed proofs on N ranks), which is `value` — and WEAK scaling (`batch` proofs per rank),
           reported beside it under "weak"."""
    import ctypes as C
    st, nat, count, typed, amount, rng = make_rp_setup(gpu, torch, dev, rank, shape)
    nlen, llen, k, fn, fl, ninit = SHAPES[shape]
    shp = nat.shape
    # distinct inputs: random 64-bit values (typed/conserved: random splits around the example's amount that keep the sum)
    if typed:
        dlt = rng.integers(-5000, 5000, size=(batch, count // 2))
        vals = np.concatenate([amount + dlt, amount - dlt], axis=1).astype(np.uint64)
    else:
        vals = rng.integers(0, 2**64, size=(batch, count), dtype=np.uint64)
    amt = np.zeros((batch, count, 4), dtype=np.uint64); amt[:, :, 0] = vals
    typ = np.zeros((batch, count, 4), dtype=np.uint64)
    bld = rng.integers(0, 2**64, size=(batch, count, 4), dtype=np.uint64); bld[:, :, 3] >>= np.uint64(1)
    plen = 24
    pre = np.frombuffer(b"".join(b"bench r%03d %012d " % (rank, b) for b in range(batch)), dtype=np.uint8)
    assert pre.size == batch * plen
    cf = np.zeros(batch * shp["coms_bytes"], dtype=np.uint8)
    pf = np.zeros(batch * shp["proof_bytes"], dtype=np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)

    def prove_once():
        gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, batch, vp(amt), vp(typ), vp(bld), vp(pre), plen, vp(cf), vp(pf)), "bppp_rp_prove_batch")

    prove_once()                                     # warm-up (workspaces, fixed-base table)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    for _ in range(prove_steps):
        prove_once()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    pdt = time.perf_counter() - tp0
    if dist is not None:
        t = torch.tensor([pdt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pdt = float(t.item())
    assert len({pf[b * shp["proof_bytes"]:(b + 1) * shp["proof_bytes"]].tobytes() for b in range(min(batch, 512))}) == min(batch, 512), "proofs are not distinct"
    # the library runs a batch of this size as two half-batches in flight (twin handle on its own context, csrc/rpprove.hip): the host
    # shares of one half (digits, half-GCDs of the argument's rounds, challenge round trips) fall under the kernels of the other.  The
    # one-context figure is reported beside `value`; both write the same bytes.
    single_context = None
    if world == 1 and batch >= 4096:
        cf2, pf2 = np.zeros_like(cf), np.zeros_like(pf)
        nat.set_option("split_min", 0)
        try:
            def one():
                gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, batch, vp(amt), vp(typ), vp(bld), vp(pre), plen, vp(cf2), vp(pf2)), "bppp_rp_prove_batch")
            one()
            torch.cuda.synchronize()
            tq0 = time.perf_counter()
            for _ in range(prove_steps):
                one()
            torch.cuda.synchronize()
            qdt = time.perf_counter() - tq0
        finally:
            nat.set_option("split_min", 4096)
        assert np.array_equal(cf2, cf) and np.array_equal(pf2, pf), "one-context prover output differs"
        single_context = {"value": batch * prove_steps / qdt, "unit": "proofs/s", "ms_per_batch": qdt / prove_steps * 1e3}
    prove = {"metric": "range_proofs_proved_per_sec", "value": world * batch * prove_steps / pdt, "unit": "proofs/s", "ms_per_batch": pdt / prove_steps * 1e3,
             "half_batches_in_flight": 2 if batch >= 4096 else 1, "single_context": single_context,
             "batch_per_gpu": batch, "replicas": world, "shape": f"{shape}: {count} x 64-bit values per proof, nrmLen {nlen}, linLen {llen}, {k} rounds",
             "scope": "proveM of RangeProof (src/RangeProof.hs:93-97) end to end: host inputs in, the reference's commitments / proof files out; "
                      "every commitment a fixed-base comb MSM over the setup's basis, the argument's rounds without point folds, per-proof field algebra, "
                      "randomness and SHA-256 transcripts on the GPU as one stream of kernels; digit extraction of the plain amounts on the host cores "
                      "(bppp_rp_prove_batch)"}

    # ---- verify
    d_c = torch.from_numpy(cf).to(dev)
    d_p = torch.from_numpy(pf).to(dev)
    seed = os.urandom(32)        # fresh secret weights per rank (include/bppp.h: never a constant outside tests)

    def step():
        ok, part = nat.verify_batch_device_point(batch, d_c.data_ptr(), d_p.data_ptr(), seed)
        assert ok, "batch of valid proofs did not verify"
        return part if world == 1 else combine(part)

    for _ in range(max(warmup, 1)):
        res = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert res is None
    # ---- STRONG scaling (BASELINE configs 4 / 5 as stated): ONE job of `batch` proofs, rank r holds proofs [lo_r, hi_r) of it — its own
    # first hi_r - lo_r files; the proofs of different ranks are distinct — verified by bppp_rp_verify_shard_device with the job-wide seed
    # (rank 0's os.urandom, broadcast) and the rank's offset; the combined points are all-gathered and added: identity <=> the job verifies
    strong = None
    if world > 1:
        from bulletproofspp_amd.dist import shard_range
        lo, hi = shard_range(batch, rank, world)
        mine = hi - lo
        sd_t = torch.zeros(32, dtype=torch.uint8, device=coll_dev if coll_dev is not None else "cpu")
        if rank == 0:
            sd_t.copy_(torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8))
        dist.broadcast(sd_t, 0)
        job_seed = bytes(sd_t.cpu().numpy().tobytes())

        def sstep():
            ok, part = nat.verify_batch_device_point(mine, d_c.data_ptr(), d_p.data_ptr(), job_seed, index_offset=lo)
            assert ok, "shard of valid proofs did not verify"
            return combine(part)
        for _ in range(max(warmup, 1)):
            sres = sstep()
        dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(steps):
            sres = sstep()
        torch.cuda.synchronize()
        dist.barrier()
        sdt = time.perf_counter() - ts0
        t = torch.tensor([sdt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sdt = float(t.item())
        assert sres is None, "the sum of the rank points of a valid job is not the identity"
        strong = {"scaling": "strong", "job_proofs": batch, "batch_per_gpu": mine, "value": batch * steps / sdt, "unit": "verifies/s",
                  "ms_per_job": sdt / steps * 1e3}
        # one corrupted proof on ONE rank must fail the whole job on EVERY rank: rank `tamper_rank` flips a bit of a final witness scalar of
        # its middle proof (outside the timed region), every rank forms the sum of the rank points again
        tr_ = TAMPER_RANK if 0 <= TAMPER_RANK < world else world - 1
        d_pt = d_p
        if rank == tr_ and mine:
            d_pt = d_p.clone()
            d_pt[(mine // 2) * nat.shape["proof_bytes"] + 9] ^= 4
        ok_t, part_t = nat.verify_batch_device_point(mine, d_c.data_ptr(), d_pt.data_ptr(), job_seed, index_offset=lo)
        assert ok_t == (rank != tr_ or not mine), "the tampered shard verified (or an honest one did not)"
        assert combine(part_t) is not None, "a job with a corrupted proof on rank %d was accepted" % tr_
        strong["tamper_check"] = "one corrupted proof on rank %d of %d: its shard rejects and the summed rank points are not the identity on every rank" % (tr_, world)
    # two verifier handles on two contexts (stream + workspace each), one host thread each, the same files: the hashing stage of one
    # batch — one wavefront per SIMD at this batch size, i.e. half of the chip's issue slots — overlaps the arithmetic of the other.
    # Reported beside the single-call figure (which stays `value`), like the MSM's `concurrent` leg.
    concurrent = None
    if world == 1:
        import threading
        from bulletproofspp_amd.capi import Bppp
        from bulletproofspp_amd import rangeproof as RP
        ctx2 = Bppp(gpu_device(dev))
        nats = [nat, RP.NativeRangeProofs(ctx2, st, h=st.g)]
        nats[1].verify_batch_device_point(batch, d_c.data_ptr(), d_p.data_ptr(), seed)          # warm-up of the second handle
        oks = [True, True]

        def work(i):
            for _ in range(steps):
                ok_, _ = nats[i].verify_batch_device_point(batch, d_c.data_ptr(), d_p.data_ptr(), seed)
                oks[i] = oks[i] and ok_
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize()
        cdt = time.perf_counter() - tc0
        assert all(oks)
        concurrent = {"handles": 2, "value": 2 * batch * steps / cdt, "unit": "verifies/s", "ms_per_batch": cdt / (2 * steps) * 1e3}
        nats[1].close(); ctx2.close()
    # the host-buffer entry point (files in pageable host memory: PCIe-inclusive, never `value`)
    th0 = time.perf_counter()
    acc = C.c_int(0)
    sd = np.frombuffer(seed, dtype=np.uint8)
    gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, batch, vp(cf), vp(pf), vp(sd), C.byref(acc), None, None, None), "bppp_rp_verify_batch")
    hdt = time.perf_counter() - th0
    assert acc.value == 1
    # the same call with the files in page-locked buffers of the library (bppp_host_alloc): DMA transfers beside the decode kernels
    pcf, ppf = gpu.host_alloc(cf.nbytes), gpu.host_alloc(pf.nbytes)
    pcf[:] = cf; ppf[:] = pf
    gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, batch, vp(pcf), vp(ppf), vp(sd), C.byref(acc), None, None, None), "bppp_rp_verify_batch")
    tp0 = time.perf_counter()
    for _ in range(3):
        gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, batch, vp(pcf), vp(ppf), vp(sd), C.byref(acc), None, None, None), "bppp_rp_verify_batch")
    pdt_pinned = (time.perf_counter() - tp0) / 3
    assert acc.value == 1
    gpu.host_free(pcf); gpu.host_free(ppf)
    # a corrupted member must be rejected (one flipped sign bit)
    pf_bad = pf.copy(); pf_bad[(batch // 2) * shp["proof_bytes"] + 32 * (fn + fl)] ^= 1
    gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, batch, vp(cf), vp(pf_bad), vp(sd), C.byref(acc), None, None, None), "bppp_rp_verify_batch")
    assert acc.value == 0, "a corrupted proof was accepted"
    terms = nlen + llen + 1 + batch * (ninit + 2 * k)
    vtab = profile_table("verify_4096_64by64")
    # algorithmic bytes per proof: (ninit + 2k) per-proof pairs x 96 B + (nlen + llen + 1) shared-basis scalars x 32 B
    bytes_per_proof = (ninit + 2 * k) * 96 + (nlen + llen + 1) * 32
    file_bytes = shp["coms_bytes"] + shp["proof_bytes"]
    verify = {"metric": "aggregated_64bit_range_proof_verifies_per_sec", "value": world * batch * steps / dt, "unit": "verifies/s", "scaling": "weak" if world > 1 else "strong",
              "ms_per_batch": dt / steps * 1e3, "batch_per_gpu": batch, "job_proofs": world * batch, "combined_msm_terms": terms,
              "algorithmic_bytes_per_proof": bytes_per_proof, "encoded_bytes_per_proof": file_bytes,
              "achieved_GBps": world * batch * steps * bytes_per_proof / dt / 1e9, "hbm_frac": world * batch * steps * bytes_per_proof / dt / 1e9 / (HBM_PEAK_GBS * world),
              "shape": f"{shape}: nrmLen {nlen}, linLen {llen}, {k} rounds, {ninit}+{2 * k} per-proof points (SURVEY.md App. B)",
              "proofs": f"{batch} DISTINCT real range proofs per GPU ({count} x 64-bit values each{', typed/conserved' if typed else ''}) made by the lockstep "
                        "prover in this run; all verify; one corrupted member is rejected",
              "scope": "verifyM of RangeProof END TO END (src/RangeProof.hs:99-105) from the encoded files resident in HBM: decodeProof (square roots, "
                       "signs), every SHA-256 transcript hash of verifyTRRPM / verifyBPM (shaOracle, app/Main.hs:64-80), public scalars, challenge "
                       "expansion, shared-basis merge, ONE combined MSM — all on the GPU, all timed (bppp_rp_verify_batch_device)",
              "roofline": {"bound": "hbm", "kernel": "whole call (decode, text, SHA-256, public scalars, expansion, combined MSM: profiles/r04_verify_4096_kernel_stats.csv)",
                           "achieved": world * batch * steps * bytes_per_proof / dt / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                           "frac": world * batch * steps * bytes_per_proof / dt / 1e9 / (HBM_PEAK_GBS * world),
                           "traffic": (vtab or {}).get("per_call", {}).get("bytes_per_call") if (batch == 4096 and shape == "64by64") else None,
                           "traffic_source": "static: profiles/traffic.json verify_4096_64by64 (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over one call of this shape and "
                                             "batch, summed over its kernels: profiles/r04_pmc_verify_per_kernel.csv)" if (vtab and batch == 4096 and shape == "64by64") else None,
                           "limiter": "valu / latency",
                           "note": "algorithmic %d B per proof (SURVEY.md 8d) x proofs / call time; every stage is VALU- or latency-bound (256-bit field "
                                   "arithmetic, SHA-256), none HBM-bound: see `stages`" % bytes_per_proof},
              "stages": stage_rows(vtab) if (batch == 4096 and shape == "64by64") else None,
              "concurrent": concurrent,
              "host_buffer_call": {"entry": "bppp_rp_verify_batch (files in pageable host memory, %d B per proof over PCIe)" % file_bytes,
                                   "ms_per_batch": hdt * 1e3, "value": batch / hdt, "unit": "verifies/s",
                                   "pinned": {"entry": "the same call with the files in page-locked host buffers (bppp_host_alloc)", "ms_per_batch": pdt_pinned * 1e3,
                                              "value": batch / pdt_pinned, "unit": "verifies/s"}}}
    if strong is not None:
        # N > 1: the stated configuration (one job, strong scaling) is the leg's `value`; the weak figure (a full batch per GPU) stays beside it
        weak = {k_: verify[k_] for k_ in ("value", "unit", "ms_per_batch", "batch_per_gpu", "job_proofs", "achieved_GBps", "hbm_frac")}
        weak["scaling"] = "weak"
        verify.update({"value": strong["value"], "scaling": "strong", "ms_per_batch": strong["ms_per_job"], "batch_per_gpu": strong["batch_per_gpu"],
                       "job_proofs": batch, "achieved_GBps": strong["value"] * bytes_per_proof / 1e9,
                       "hbm_frac": strong["value"] * bytes_per_proof / 1e9 / (HBM_PEAK_GBS * world), "weak": weak,
                       "combined_msm_terms": nlen + llen + 1 + strong["batch_per_gpu"] * (ninit + 2 * k), "tamper_check": strong["tamper_check"]})
    if cpu_baseline_leg and rank == 0:
        # the reference verifies ONE proof with ONE 256-row Straus MSM over nlen + llen + 1 + ninit + 2k terms
        # (src/Bulletproof.hs:377): time the oracle's restatement of that on this host (single thread)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        from bulletproofspp_amd.capi import points_to_array
        ec = pyoracle.CEC()
        nterm = nlen + llen + 1 + ninit + 2 * k
        sc = rng.integers(0, 2**64, size=(nterm, 4), dtype=np.uint64)
        sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))
        basis = points_to_array(list(st.gs) + list(st.hs) + [st.g])
        pts_np = np.ascontiguousarray(np.concatenate([basis, np.tile(basis[:1], (ninit + 2 * k, 1))]))
        u64p = C.POINTER(C.c_uint64)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ec.inner_product_raw(sc.ctypes.data_as(u64p), pts_np.ctypes.data_as(u64p), nterm)
        cdt = (time.perf_counter() - t0) / reps
        verify["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "verifies/s", "cores": 1, "kind": "port",
                                  "sample": f"{reps} single-proof verifier MSMs of {nterm} terms (the reference's one commit per verify, Bulletproof.hs:377) "
                                            "through oracle/bppp_oracle.c's 256-row Straus restatement; hashing and decoding not included (they favour the baseline)"}
    nat.close()
    return verify, prove


def bench_binary_verify(gpu, torch, dev, batch: int, steps: int, cpu_baseline_leg: bool = True):
    """RangeProof.Binary (src/RangeProof/Binary.hs) at the 64 x 64-bit shape — BASELINE config 3 read literally ("64x64-bit aggregated BINARY
    range proof"): 64 outputs in [0, 2^64), one bit per norm position (nrmLen 4096, linLen 2, 10 rounds, an 867-byte proof), conserved
    against one public input as witnessBRP requires (:158-166), norm-linear argument.  `batch` DISTINCT proofs made in this run by the
    library's lockstep prover (bppp_rp_prove_batch on a bppp_rp_create_binary handle: proveBRPM + proveBPM as one stream of kernels once the
    comb table of the setup exists) and verified end to end from their files in HBM (bppp_rp_verify_batch_device: verifyBRPM's two oracle calls, verifyBPM, one
    combined MSM of 4100 + batch * 86 terms); one corrupted member must be rejected and identified."""
    from bulletproofspp_amd import rangeproof as RP, rangeproof_binary as RB
    from bulletproofspp_amd.capi import array_to_point
    count, amount = 64, 10000
    rds = [RB.make_range_data(0, 2**64, True, False)] * count
    need = 4 + sum(len(rd.base_coeffs) for rd in rds)
    rng = np.random.default_rng(0xB1)
    pts = None
    while pts is None or pts.shape[0] < need:
        xs = rng.integers(0, 2**64, size=(3 * need, 4), dtype=np.uint64)
        dx = torch.from_numpy(xs.view(np.int64)).to(dev)
        dp = torch.zeros((3 * need, 8), dtype=torch.int64, device=dev)
        gpu.lift_x(dx.data_ptr(), 3 * need, dp.data_ptr())
        pts = dp[(dp != 0).any(dim=1)]
    P = pts[:need].cpu().numpy().view(np.uint64)
    basis = [array_to_point(P[i]) for i in range(need)]
    st = RB.setup(RP.GpuBackend(gpu), basis, True, rds, amount * count, "NL")
    nat = RB.NativeBinaryRangeProofs(gpu, st, h=basis[0])
    shp = nat.shape
    dlt = rng.integers(-5000, 5000, size=(batch, count // 2))
    vals = np.concatenate([amount + dlt, amount - dlt], axis=1).astype(np.uint64)
    bld = rng.integers(1, 2**63, size=(batch, count), dtype=np.uint64)
    import ctypes as C
    amt = np.zeros((batch, count, 4), dtype=np.uint64); amt[:, :, 0] = vals
    typ = np.zeros((batch, count, 4), dtype=np.uint64)
    bl4 = np.zeros((batch, count, 4), dtype=np.uint64); bl4[:, :, 0] = bld
    pre = np.frombuffer(b"".join(b"bench bin %010d" % i for i in range(batch)), dtype=np.uint8)
    cf = np.zeros(batch * shp["coms_bytes"], dtype=np.uint8); pf = np.zeros(batch * shp["proof_bytes"], dtype=np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    tp0 = time.perf_counter()
    gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, batch, vp(amt), vp(typ), vp(bl4), vp(pre), 20, vp(cf), vp(pf)), "bppp_rp_prove_batch (binary)")
    pdt = time.perf_counter() - tp0
    cf1, pf1 = cf.copy(), pf.copy()
    tp0 = time.perf_counter()
    gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, batch, vp(amt), vp(typ), vp(bl4), vp(pre), 20, vp(cf), vp(pf)), "bppp_rp_prove_batch (binary)")
    pdt2 = time.perf_counter() - tp0
    assert np.array_equal(cf, cf1) and np.array_equal(pf, pf1), "the binary prover is not deterministic in its inputs"
    dc, dpf = gpu.to_device(cf), gpu.to_device(pf)
    seed = os.urandom(32)
    assert nat.verify_batch_device(batch, dc, dpf, seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ok = nat.verify_batch_device(batch, dc, dpf, seed)
    dt = time.perf_counter() - t0
    assert ok
    pf2 = pf.copy(); pf2[(batch // 2) * shp["proof_bytes"] + 9] ^= 4
    dp2 = gpu.to_device(pf2)
    ok2, status, _ = nat.verify_batch_device(batch, dc, dp2, seed, want_status=True)
    assert not ok2 and [i for i, s_ in enumerate(status) if s_] == [batch // 2], "a corrupted binary proof was not identified"
    gpu.free(dc); gpu.free(dpf); gpu.free(dp2)
    k, ninit = shp["rounds"], 2 + count
    out = {"metric": "aggregated_64x64bit_binary_range_proof_verifies_per_sec", "value": batch * steps / dt, "unit": "verifies/s", "ms_per_batch": dt / steps * 1e3,
           "batch": batch, "combined_msm_terms": shp["norm_len"] + shp["lin_len"] + 1 + batch * (ninit + 2 * k),
           "encoded_bytes_per_proof": shp["coms_bytes"] + shp["proof_bytes"],
           "shape": "RangeProof.Binary, 64 x 64-bit outputs conserved against one public input: nrmLen %d, linLen %d, %d rounds, proof %d B" % (
               shp["norm_len"], shp["lin_len"], k, shp["proof_bytes"]),
           "scope": "verifyBRPM + verifyBPM end to end from the encoded files resident in HBM (bppp_rp_verify_batch_device on a bppp_rp_create_binary handle); one corrupted "
                    "member rejected and identified",
           "prove": {"value": batch / pdt2, "unit": "proofs/s", "ms_per_batch": pdt2 * 1e3, "first_call_ms": pdt * 1e3,
                     "scope": "bppp_rp_prove_batch on the binary handle, second call (the first builds the comb table of the 4099 basis points: first_call_ms): "
                              "proveBRPM + proveBPM as ONE stream of kernels (csrc/brpprove_dev.hip, csrc/nlb.hip fixed-basis mode, csrc/rpp_transcript.hip), two "
                              "half-batches in flight; the phase commitments and the first three rounds are comb MSMs over the setup's 4099 points, then the level-3 "
                              "basis of every proof is materialised by one table walk and the last seven rounds are bucket MSMs over those 514 points; the host "
                              "extracts the binary digits and writes the files",
                     "bound": "per proof: 3 rounds x (4099 + 2050) + 4099 (blCom) + 4098 (level basis) full-width terms x 20 windows (c = 13) = 0.53 M mixed additions, "
                              "+ 7 rounds x 2 bucket MSMs of 514 terms (~25 k additions each) = 0.88 M in all (1.31 M with every round on the table): at the "
                              "accumulate kernels' ~15 G additions/s ~59 us per proof, ~17 k proofs/s for this schedule"}}
    bytes_per_proof = (ninit + 2 * k) * 96 + (shp["norm_len"] + shp["lin_len"] + 1) * 32
    btab = profile_table("verify_binary_1024_64x64bit") if batch == 1024 else None
    ach = batch * steps * bytes_per_proof / dt / 1e9
    out["algorithmic_bytes_per_proof"] = bytes_per_proof
    out["roofline"] = {"bound": "hbm", "kernel": "whole call (decode, text, SHA-256, k_brp_public, expansion, combined MSM: profiles/r04_binary_verify_1024_kernel_stats.csv)",
                       "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "traffic": (btab or {}).get("per_call", {}).get("bytes_per_call"),
                       "traffic_source": "static: profiles/traffic.json verify_binary_1024_64x64bit (profiles/r04_pmc_binary_verify_per_kernel.csv)" if btab else None,
                       "limiter": "valu / latency",
                       "note": "algorithmic %d B per proof ((2 + 64 + 2 x 10) per-proof pairs x 96 B + 4099 shared-basis scalars x 32 B) x proofs / call time" % bytes_per_proof}
    out["stages"] = stage_rows(btab)
    if cpu_baseline_leg:
        # the reference verifies ONE such proof with ONE 256-row Straus commit over 4099 + 66 + 20 = 4185 terms (src/Bulletproof.hs:377)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        from bulletproofspp_amd.capi import points_to_array
        ec = pyoracle.CEC()
        nterm = shp["norm_len"] + shp["lin_len"] + 1 + ninit + 2 * k
        sc = rng.integers(0, 2**64, size=(nterm, 4), dtype=np.uint64)
        sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))
        pts_np = np.ascontiguousarray(np.tile(P[:need], (2, 1))[:nterm])
        u64p = C.POINTER(C.c_uint64)
        reps = 3
        tc0 = time.perf_counter()
        for _ in range(reps):
            ec.inner_product_raw(sc.ctypes.data_as(u64p), pts_np.ctypes.data_as(u64p), nterm)
        cdt = (time.perf_counter() - tc0) / reps
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "verifies/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} single-proof verifier MSMs of {nterm} terms (the reference's one commit per verifyBPM, Bulletproof.hs:377) through "
                                         "oracle/bppp_oracle.c's 256-row Straus restatement; hashing and decoding not included (they favour the baseline)"}
    nat.close()
    return out


def bench_ip_verify(gpu, torch, dev, batch: int, steps: int, cpu_baseline_leg: bool = True):
    """The inner-product flavour at batch scale (the CLI's DEFAULT argument, app/Parse.hs:100): `batch` DISTINCT encoded
    proofs of the examples/64bit shape — ONE 64-bit value, base 16 inline, nrmLen 16, linLen 6, 3 rounds, the paper's 416-byte proof
    (README.md:169-172) — made in this run by the library's lockstep prover (bppp_rp_prove_batch over a flavour-1 setup: range-proof phases,
    then csrc/ipb.hip's device-resident argument — every commitment an MSM over the original basis, no basis change, no point fold) and
    verified end to end from their files in HBM by bppp_rp_verify_batch_device: decode, all SHA-256 transcript hashing, verifyTRRPM's
    scalars, expandChallenges of InnerProductArgument.hs:103-124 / :172-181 with makeNorm's basis change folded into the shared-basis
    scalars, ONE combined MSM of 23 + batch * 11 terms."""
    import ctypes as C
    from bulletproofspp_amd import encoding as E
    from bulletproofspp_amd import rangeproof as RP
    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "examples", "64bit", "schema.json")))      # the reference's examples/64bit/schema.json (data fixture)
    st = RP.setup_from_schema(RP.GpuBackend(gpu), schema)
    assert (st.flavour, st.nrm_len, st.lin_len, st.rounds, tuple(st.final_lens)) == ("IP", 16, 6, 3, (2, 1))
    nat = RP.NativeRangeProofs(gpu, st)
    rng = np.random.default_rng(0x1664)
    vals = rng.integers(0, 2**64, size=batch, dtype=np.uint64)
    amt = np.zeros((batch, 1, 4), dtype=np.uint64); amt[:, 0, 0] = vals
    typ = np.zeros((batch, 1, 4), dtype=np.uint64)
    bld = rng.integers(0, 2**64, size=(batch, 1, 4), dtype=np.uint64); bld[:, :, 3] >>= np.uint64(1)
    plen = 20
    pre = np.frombuffer(b"".join(b"bench ip %010d " % b_ for b_ in range(batch)), dtype=np.uint8)
    assert pre.size == batch * plen
    cf = np.zeros(batch * nat.shape["coms_bytes"], dtype=np.uint8)
    pf = np.zeros(batch * nat.shape["proof_bytes"], dtype=np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)

    def prove_once():
        gpu._check(gpu.lib.bppp_rp_prove_batch(nat.h, batch, vp(amt), vp(typ), vp(bld), vp(pre), plen, vp(cf), vp(pf)), "bppp_rp_prove_batch")
    prove_once()
    tp0 = time.perf_counter()
    prove_once()
    prove_s = time.perf_counter() - tp0
    pb = nat.shape["proof_bytes"]
    assert pb == 418 and len({pf[b_ * pb:(b_ + 1) * pb].tobytes() for b_ in range(min(batch, 512))}) == min(batch, 512), "proofs are not distinct"
    # one proof of the batch against the host protocol code over the per-proof device argument (bppp_ip_*): the same bytes
    p0 = RP.prove(st, RP.witness(st, [(int(vals[0]), 0, int(bld[0, 0, 0]) | int(bld[0, 0, 1]) << 64 | int(bld[0, 0, 2]) << 128 | int(bld[0, 0, 3]) << 192)]),
                  RP.sha256_oracle(), RP.hash_to_scalar(b"bench ip %010d " % 0))
    c0, f0 = E.encode_proof(4, p0)
    assert cf[:len(c0)].tobytes() == c0 and pf[:pb].tobytes() == f0, "lockstep inner-product prover differs from the host protocol code"
    kept = [p0]
    d_c, d_p = torch.from_numpy(cf).to(dev), torch.from_numpy(pf).to(dev)
    seed = os.urandom(32)
    ok, _ = nat.verify_batch_device_point(batch, d_c.data_ptr(), d_p.data_ptr(), seed)
    assert ok, "batch of valid inner-product proofs did not verify"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ok, _ = nat.verify_batch_device_point(batch, d_c.data_ptr(), d_p.data_ptr(), seed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert ok
    pf_bad = pf.copy(); pf_bad[(batch // 3) * nat.shape["proof_bytes"] + 3] ^= 1
    acc = C.c_int(1)
    sd = np.frombuffer(seed, dtype=np.uint8)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    gpu._check(gpu.lib.bppp_rp_verify_batch(nat.h, batch, vp(cf), vp(pf_bad), vp(sd), C.byref(acc), None, None, None), "bppp_rp_verify_batch")
    assert acc.value == 0, "a corrupted inner-product proof was accepted"
    nlen, llen, k, ninit = st.nrm_len, st.lin_len, st.rounds, 4 + len(st.rds)
    bytes_per_proof = (ninit + 2 * k) * 96 + (nlen + llen + 1) * 32
    out = {"metric": "single_64bit_range_proof_verifies_per_sec (inner-product argument)", "value": batch / dt, "unit": "verifies/s", "ms_per_batch": dt * 1e3,
           "batch": batch, "distinct_proofs": batch, "combined_msm_terms": nlen + llen + 1 + batch * (ninit + 2 * k),
           "algorithmic_bytes_per_proof": bytes_per_proof, "encoded_bytes_per_proof": nat.shape["coms_bytes"] + nat.shape["proof_bytes"],
           "achieved_GBps": batch * bytes_per_proof / dt / 1e9, "hbm_frac": batch * bytes_per_proof / dt / 1e9 / HBM_PEAK_GBS,
           "shape": "examples/64bit: 1 x 64-bit value, base 16 inline, IP argument, nrmLen 16, linLen 6, 3 rounds, 34-term verifier MSM per proof, 418-byte proof file",
           "proofs": f"{batch} DISTINCT real proofs made by the lockstep inner-product prover in this run (one checked byte for byte against the host "
                     "protocol code); all verify; one corrupted member is rejected",
           "prove": {"metric": "single_64bit_range_proofs_proved_per_sec (inner-product argument)", "value": batch / prove_s, "unit": "proofs/s", "ms_per_batch": prove_s * 1e3,
                     "scope": "bppp_rp_prove_batch, flavour 1: range-proof phases (csrc/rpprove_dev.hip), transcript (csrc/rpp_transcript.hip) and the argument "
                              "(csrc/ipb.hip) as one stream of kernels, every commitment a comb MSM over the setup's ORIGINAL basis (no basis change, no point "
                              "fold); the host extracts digits and writes the files"},
           "roofline": {"bound": "hbm", "kernel": "whole call (profiles/r04_ip_prove_verify_16384_kernel_stats.csv)", "achieved": batch * bytes_per_proof / dt / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": batch * bytes_per_proof / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "limiter": "valu / latency",
                        "note": "algorithmic %d B per proof x proofs / call time; no counter pass was taken on this leg" % bytes_per_proof},
           "scope": "verifyM of RangeProof end to end from the encoded files in HBM, inner-product flavour (bppp_rp_verify_batch_device, flavour 1)"}
    if cpu_baseline_leg:
        # the reference's verifier for ONE such proof: makeNorm's basis change (one 256-row scalar multiplication per basis pair,
        # InnerProductArgument.hs:204), expandChallenges, one 34-term commit — through the oracle restatement, single thread
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pyoracle
        from rp_backends import OracleBackend
        ob = OracleBackend(pyoracle.CEC())
        sample = [RP.verify_inputs(st, p_, RP.sha256_oracle()) for p_ in kept] * 3
        tc0 = time.perf_counter()
        for v in sample:
            assert ob.verify_bp("IP", v["q"], v["sp"], st.g, v["pub_norm"], st.gs, v["pub_lin_c"], v["pub_lin_x"], st.hs, v["es"], v["responses"], v["wit_norm"],
                                v["wit_lin"], v["init_terms"])
        cdt = (time.perf_counter() - tc0) / len(sample)
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "verifies/s", "cores": 1, "kind": "port",
                               "sample": f"{len(sample)} proofs of the batch through the oracle's verifyBPM of the inner-product flavour (basis change by 8 full scalar "
                                         "multiplications, then one 34-term Straus commit; oracle/pyoracle.py driving oracle/bppp_oracle.c); hashing and decoding not included"}
    nat.close()
    return out


def prove_cpu_baseline(shape: str, reps: int = 2):
    """The reference's proveBPM for one argument of this shape through the oracle (oracle/pyoracle.py driving the C restatement's
    256-row Straus commits and 129-row pair folds), single thread: a reported baseline for the prove leg."""
    import random
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    nlen, llen, k, _, _, _ = SHAPES[shape]
    ec = O.CEC()
    pts = O.hash_points(b"cpu prove", 1 + nlen + llen)
    rnd = random.Random(7)
    r = lambda n: [rnd.randrange(O.N) for _ in range(n)]
    t0 = time.perf_counter()
    for j in range(reps):
        com = O.PSV(rnd.randrange(O.N), pts[0], O.NormLinear.make(1, rnd.randrange(O.N), r(llen), r(nlen), pts[1:1 + nlen], r(llen), pts[1 + nlen:]))
        O.prove_bp(k, com, O.Transcript(O.sha_oracle_fn(b"cpu%d" % j)), ec)
    dt = (time.perf_counter() - t0) / reps
    return {"value": 1.0 / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": f"{reps} norm-linear arguments of the same shape, proveBPM through the oracle restatement (Straus commits, 129-row pair folds), single "
                      "thread; the range-proof phases before the argument (4 + #values more commits) are NOT included, which favours the baseline"}


def launcher_selftest(args):
    """What a rank does under --launcher-selftest: gloo rendezvous from the launcher's environment, one all-gather of (rank, shard of a
    4096-proof job), one line from rank 0.  Exercises launch_ranks and the job-sharding arithmetic of the strong-scaling legs without a GPU."""
    import torch
    import torch.distributed as dist
    from bulletproofspp_amd.dist import shard_range
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BPPP_SELFTEST_FAIL_RANK") == str(rank):
        raise SystemExit(3)                                         # the launcher must report this and end the other ranks
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(4096, rank, world)
    mine = torch.tensor([[rank, lo, hi]], dtype=torch.int64)
    out = torch.zeros((world, 3), dtype=torch.int64)
    if world > 1:
        dist.all_gather_into_tensor(out, mine)
        dist.barrier()
    else:
        out = mine
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "ranks": [int(v) for v in out[:, 0]], "shards": [[int(a), int(b_)] for a, b_ in out[:, 1:]]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


TAMPER_RANK = -1      # --tamper-rank: which rank corrupts a proof in the strong-scaling check (-1 = the last one)


def gpu_device(dev) -> int:
    return dev.index if getattr(dev, "index", None) is not None else 0


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` started plainly: this process — which has imported neither torch nor the library and has made no
    HIP call — starts N fresh child processes of this same script (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, rendezvous on 127.0.0.1), relays rank 0's JSON line and returns non-zero if any rank failed.  Nothing is
    exec'ed from a GPU-initialised process; `python -m torch.distributed.run ... bench.py --gpus N` keeps working (the children of
    that launcher find WORLD_SIZE set and go straight to the benchmark)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p_ in enumerate(procs):
            if p_.poll() not in (None, 0):
                failed = r
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, p_ in enumerate(procs) if p_.returncode != 0), None)
    if failed is not None:                      # the others may sit in a collective waiting for the dead rank: end exactly the processes started here
        time.sleep(2.0)
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p_.kill()
    reader.join(timeout=10)
    text = out0[0] if out0 else ""
    sys.stdout.write(text)
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: rank {failed} exited with {procs[failed].returncode}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--window", type=int, default=0, help="Pippenger window bits (0 = library heuristic)")
    ap.add_argument("--cpu-sample-log2", type=int, default=17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-batch", type=int, default=4096, help="distinct proofs per GPU proved in lockstep and then batch-verified end to end (0 = skip both legs)")
    ap.add_argument("--ip-batch", type=int, default=1 << 14, help="inner-product flavour leg: single 64-bit proofs (examples/64bit) verified per batch (0 = skip; N = 1 only)")
    ap.add_argument("--binary-batch", type=int, default=1024, help="RangeProof.Binary leg: 64 x 64-bit binary proofs verified per batch (0 = skip; N = 1 only)")
    ap.add_argument("--msm-streams", type=int, default=2, help="extra leg: MSMs in flight on that many contexts (1 = skip; N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--tamper-rank", type=int, default=-1, help="N > 1: the rank that corrupts one of its proofs in the strong-scaling rejection check (-1 = the last rank)")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure the HBM bytes of k_acc_points in this run")
    ap.add_argument("--headline-only", action="store_true", help="only the 2^log2n MSM leg (profiling passes: one launch shape per kernel)")
    ap.add_argument("--check-combined", action="store_true", help="N > 1: rank 0 also computes the whole sharded MSM alone (all ranks' inputs regenerated "
                                                                  "from their seeds) and asserts the combined point equals it")
    ap.add_argument("--launcher-selftest", action="store_true", help="no GPU: the ranks only rendezvous over gloo, all-gather their rank numbers and shard ranges, "
                                                                    "and rank 0 prints them (the CPU-tier test of the plain `--gpus N` start)")
    args = ap.parse_args()
    global TAMPER_RANK
    TAMPER_RANK = args.tamper_rank

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # plain start: become the launcher BEFORE torch / HIP are touched

    if args.launcher_selftest:
        return launcher_selftest(args)

    import torch
    import bulletproofspp_amd as b
    from bulletproofspp_amd.capi import array_to_point, points_to_array

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` plainly, or under torch.distributed.run with --nproc-per-node N")
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    coll_dev = dev if args.backend == "nccl" else None      # gloo rehearsal gathers through host tensors
    from bulletproofspp_amd.dist import all_gather_points, all_gather_points_async

    n = 1 << args.log2n
    gpu = b.Bppp(local)
    # one HIP stream for torch's copies/collectives and the library's kernels (ordering by stream, no extra syncs)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    gpu.set_stream(work_stream.cuda_stream)
    dsc, dpts = make_inputs(gpu, torch, dev, n, seed=0xB9B9 + rank)
    ones_np = np.zeros((world, 4), dtype=np.uint64)
    ones_np[:, 0] = 1

    def combine(part):
        """all-gather one 64-B partial point per rank (RCCL has no mod-p reduction) and add them with the library"""
        allp = all_gather_points(points_to_array([part])[0], dist, coll_dev)
        return gpu.sum_points(allp)

    def step():
        part = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
        return part if world == 1 else combine(part)

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
    if world == 1:
        for _ in range(args.steps):
            res = step()
    else:
        # every step's 64-B all-gather is issued without blocking and overlaps the next step's kernels; all K sums are
        # completed (waited for and added) before the clock stops
        pend = []
        for _ in range(args.steps):
            part = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
            pend.append(all_gather_points_async(points_to_array([part])[0], dist, coll_dev))
        sums = [gpu.sum_points(p.result()) for p in pend]
        assert all(s_ == sums[0] for s_ in sums), "sharded MSM results differ between steps"
        res = sums[-1]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    # STRONG scaling of the MSM metric beside the weak headline: ONE 2^log2n-term MSM, rank r owns the contiguous slice
    # shard_range(n, r, N) of it (here: the first hi - lo pairs of its own inputs), partial points all-gathered and added
    msm_strong = None
    if world > 1:
        from bulletproofspp_amd.dist import shard_range
        lo_, hi_ = shard_range(n, rank, world)
        ns_ = hi_ - lo_

        def strong_step():
            part = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), ns_, args.window)
            return combine(part)
        for _ in range(max(1, args.warmup)):
            sres_ = strong_step()
        dist.barrier()
        torch.cuda.synchronize()
        tS = time.perf_counter()
        for _ in range(args.steps):
            sres_ = strong_step()
        torch.cuda.synchronize()
        dist.barrier()
        sdt_ = time.perf_counter() - tS
        t = torch.tensor([sdt_], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sdt_ = float(t.item())
        msm_strong = {"scaling": "strong", "workload": f"ONE pedersen_msm_2^{args.log2n}_secp256k1 sharded terms/{world}", "pairs_per_gpu": ns_,
                      "value": n * args.steps / sdt_, "unit": "pairs/s", "ms_per_msm": sdt_ / args.steps * 1e3}
    combined_check = None
    if world > 1 and args.check_combined and rank == 0:
        # the sharded job as ONE single-rank MSM: every rank's slice regenerated from its seed, concatenated in rank order
        parts = [make_inputs(gpu, torch, dev, n, seed=0xB9B9 + r) for r in range(world)]
        all_sc, all_pt = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
        whole = gpu.msm_device(all_sc.data_ptr(), all_pt.data_ptr(), n * world, 0)
        assert whole == res, "combined point of the sharded MSM differs from the single-rank MSM over all terms"
        combined_check = "sum of %d rank-local MSMs == single-rank MSM over all %d terms" % (world, n * world)
        del parts, all_sc, all_pt
    stages, calls = gpu.profile_read(reset=True)
    gpu.profile_enable(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # BASELINE config 2: the 2^16-term MSM (the first 2^16 pairs of the same synthetic inputs), same entry point
    small = None
    if world == 1 and args.log2n > 16 and not args.headline_only:
        n16 = 1 << 16
        r16 = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n16, 0)
        gpu.profile_read(reset=True); gpu.profile_enable(True)
        torch.cuda.synchronize()
        t16 = time.perf_counter()
        reps16 = max(10, args.steps)
        for _ in range(reps16):
            r16b = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n16, 0)
        torch.cuda.synchronize()
        d16 = time.perf_counter() - t16
        st16, c16 = gpu.profile_read(reset=True)
        gpu.profile_enable(False)
        assert r16b == r16
        small = {"workload": "pedersen_msm_2^16_secp256k1 (BASELINE config 2)", "value": n16 * reps16 / d16, "unit": "pairs/s", "ms_per_msm": d16 / reps16 * 1e3,
                 "stages_ms_per_msm": {k_: v_ / max(c16, 1) for k_, v_ in st16.items()}}

    # fixed Pedersen basis: the SAME 2^log2n points registered once (bppp_basis: table 2^(c w) P_i built outside the timed region), then
    # MSMs over fresh scalars with one bucket set for all windows.  A separate leg: the headline stays the arbitrary-point MSM.
    fixed = None
    if world == 1 and not args.headline_only:
        tb0 = time.perf_counter()
        bas = gpu.basis(dpts.data_ptr(), device=True, n=n)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - tb0
        assert bas.msm(dsc.data_ptr(), n, 1)[0] == res, "fixed-basis MSM differs from the arbitrary-point MSM"
        gpu.profile_read(reset=True); gpu.profile_enable(True)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            rf = bas.msm(dsc.data_ptr(), n, 1)[0]
        torch.cuda.synchronize()
        fdt = time.perf_counter() - tf0
        stf, cf_ = gpu.profile_read(reset=True)
        gpu.profile_enable(False)
        assert rf == res
        fixed = {"workload": f"pedersen_msm_2^{args.log2n}_secp256k1 over a REGISTERED basis (bppp_basis + bppp_msm_basis)", "value": n * args.steps / fdt, "unit": "pairs/s",
                 "ms_per_msm": fdt / args.steps * 1e3, "window_bits": bas.window_bits, "table_bytes": bas.table_bytes, "table_build_ms": build_s * 1e3,
                 "stages_ms_per_msm": {k_: v_ / max(cf_, 1) for k_, v_ in stf.items()},
                 "note": "precomputation outside the timed region; same result as the arbitrary-point MSM (asserted)"}
        bas.close()

    # the prover's shape: THOUSANDS of MSMs over ONE short basis (4096 instances x 774 terms).  Three routes, same results (asserted):
    # arbitrary points (bucket method per instance), the registered basis (one bucket set per instance), and the registered basis with
    # its comb table (every multiple of every window in HBM: one mixed addition per non-zero digit, csrc/comb.hip).
    fixed_batch = None
    if world == 1 and not args.headline_only:
        nb, inst = 774, 4096
        rngb = np.random.default_rng(77)
        scb = rngb.integers(0, 2**64, size=(inst * nb, 4), dtype=np.uint64)
        scb[:, 3] >>= np.uint64(1)
        dscb = torch.from_numpy(scb.view(np.int64)).to(dev)
        ptsb = dpts[:nb].contiguous()
        legs = {}
        import ctypes as C_
        vpp = lambda v: C_.c_void_p(v)
        outs_b = [np.zeros((inst, 8), dtype=np.uint64) for _ in range(3)]       # the C entry points write [batch][8] straight into these

        def plain_call():
            gpu._check(gpu.lib.bppp_msm_batch_device(gpu.h, vpp(dscb.data_ptr()), vpp(ptsb.data_ptr()), nb, inst, 1, 0, vpp(outs_b[0].ctypes.data)), "bppp_msm_batch_device")

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / reps
        dtb = timed(plain_call)
        legs["arbitrary_points"] = {"ms": dtb * 1e3, "pairs_per_s": nb * inst / dtb}
        basb = gpu.basis(ptsb.data_ptr(), device=True, n=nb, batch_hint=inst)
        dtb = timed(lambda: gpu._check(gpu.lib.bppp_msm_basis(basb.h, vpp(dscb.data_ptr()), nb, inst, vpp(outs_b[1].ctypes.data)), "bppp_msm_basis"))
        assert np.array_equal(outs_b[1], outs_b[0])
        legs["registered_basis"] = {"ms": dtb * 1e3, "pairs_per_s": nb * inst / dtb, "window_bits": basb.window_bits, "table_bytes": basb.table_bytes}
        tcb0 = time.perf_counter()
        cwb, ctb = basb.enable_comb()
        torch.cuda.synchronize()
        comb_build = time.perf_counter() - tcb0
        dtb = timed(lambda: gpu._check(gpu.lib.bppp_msm_basis(basb.h, vpp(dscb.data_ptr()), nb, inst, vpp(outs_b[2].ctypes.data)), "bppp_msm_basis"))
        assert np.array_equal(outs_b[2], outs_b[0])
        legs["registered_basis_comb"] = {"ms": dtb * 1e3, "pairs_per_s": nb * inst / dtb, "window_bits": cwb, "table_bytes": ctb, "table_build_ms": comb_build * 1e3}
        # the prover's dominant kernel (k_comb_msm: 63 of 78 kernel-ms per 4096 proofs) on the prover's shape, timed live above (wall clock around
        # the call: the kernel plus the 256-KB copy of the results).  Algorithmic bytes: per term its 32-B scalar and one 64-B table row per window
        # (ceil(257 / c) rows: the fixed-base method trades these reads for the additions it saves); HBM traffic from the PMC passes over the
        # prove command (profiles/traffic.json, static)
        wcomb = -(-257 // cwb)
        comb_bytes = inst * nb * (32 + 64 * wcomb)
        comb_traffic = None
        try:
            pk = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["prover_4096_proofs_64by64"]["by_kernel"]
            ck = next(v for k_, v in pk.items() if "k_comb_msm<" in k_)        # (not k_comb_msm_rows: this leg runs bppp_msm_basis, one wavefront per instance)
            # the profile's launches average SQ_WAVES instances (one wavefront per instance): scale to this launch's `inst` instances
            comb_traffic = (ck["fetch_bytes_per_launch_x2"] + ck["write_bytes_per_launch"]) / ck["sq_per_launch"]["SQ_WAVES"] * inst
        except Exception:
            pass
        comb_roofline = {"bound": "hbm", "kernel": "k_comb_msm", "achieved": comb_bytes / dtb / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": comb_bytes / dtb / 1e9 / HBM_PEAK_GBS, "traffic": comb_traffic,
                         "traffic_source": "static: profiles/traffic.json, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE over benchmarks/prove_timing.py 4096: bytes per "
                                           "instance (mean over the 22 k_comb_msm launches of two batches, 7447 instances of 774 terms each on average) x this launch's instances",
                         "note": "%d instances x %d terms x (32-B scalar + %d windows x 64-B table row) per launch / live duration; the kernel is VALU-bound "
                                 "(2.4 k instructions per mixed addition; profiles/r03_pmc_prover_per_kernel.csv: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = 0.36, "
                                 "SQ_WAIT_ANY 0.06)" % (inst, nb, wcomb)}
        basb.close()
        fixed_batch = {"workload": f"{inst} MSMs of {nb} terms over one basis (the prover's commitments), results on the host", "routes": legs, "roofline": comb_roofline}
        del dscb

    # throughput with several MSMs in flight (one context = one stream + one host thread each): the latency-bound stages of one
    # (bucket reduction, window combine, the host round trip) overlap the accumulate kernel of another.  Reported beside the
    # single-stream headline, whose per-kernel durations are what the roofline and the rocprof summaries refer to.
    concurrent = None
    if world == 1 and args.msm_streams > 1 and not args.headline_only:
        import threading
        ctxs = [b.Bppp(local) for _ in range(args.msm_streams)]
        for c_ in ctxs:
            c_.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
        per = max(2, args.steps // 2)
        outs = [None] * len(ctxs)

        def work(i):
            for _ in range(per):
                outs[i] = ctxs[i].msm_device(dsc.data_ptr(), dpts.data_ptr(), n, args.window)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        cdt = time.perf_counter() - tc0
        assert all(o == res for o in outs), "concurrent MSMs disagree with the single-stream result"
        for c_ in ctxs:
            c_.close()
        concurrent = {"streams": len(ctxs), "value": n * per * len(ctxs) / cdt, "unit": "pairs/s", "ms_per_msm": cdt / (per * len(ctxs)) * 1e3,
                      "msms": per * len(ctxs)}

    # the drop-in entry point takes HOST buffers (bppp_msm, what innerProduct's FFI stub calls): PCIe-inclusive rate, never `value`
    host_call = None
    if world == 1 and not args.headline_only:
        sc_h = np.ascontiguousarray(dsc.cpu().numpy().view(np.uint64))
        pt_h = np.ascontiguousarray(dpts.cpu().numpy().view(np.uint64))
        assert gpu.msm(sc_h, pt_h) == res
        reps = 3
        th0 = time.perf_counter()
        for _ in range(reps):
            gpu.msm(sc_h, pt_h)
        hdt = (time.perf_counter() - th0) / reps
        host_call = {"entry": "bppp_msm (pageable host buffers in, 96 B per pair over PCIe)", "ms_per_call": hdt * 1e3,
                     "value": n / hdt, "unit": "pairs/s"}
        del sc_h, pt_h

    verify = prove = None
    if args.verify_batch > 0 and not args.headline_only:
        vsteps = max(3, args.steps // 2)
        verify, prove = bench_rangeproofs(gpu, torch, dev, rank, world, dist, combine, args.verify_batch, vsteps, 1, "64by64",
                                          cpu_baseline_leg=(world == 1 and not args.no_cpu_baseline), coll_dev=coll_dev)
        v2, p2 = bench_rangeproofs(gpu, torch, dev, rank, world, dist, combine, max(1, args.verify_batch // 2), vsteps, 1, "128by64+typed", coll_dev=coll_dev)
        verify["other_shapes"] = [v2]
        prove["other_shapes"] = [p2]
        if world == 1 and not args.no_cpu_baseline:
            prove["cpu_baseline"] = prove_cpu_baseline("64by64")
    verify_ip = None
    if world == 1 and args.ip_batch > 0 and not args.headline_only:
        verify_ip = bench_ip_verify(gpu, torch, dev, args.ip_batch, max(3, args.steps // 2), cpu_baseline_leg=not args.no_cpu_baseline)

    verify_binary = None
    if world == 1 and args.binary_batch > 0 and not args.headline_only:
        verify_binary = bench_binary_verify(gpu, torch, dev, args.binary_batch, max(3, args.steps // 2), cpu_baseline_leg=not args.no_cpu_baseline)

    if rank == 0:
        per_call = {k: v / max(calls, 1) for k, v in stages.items()}
        # with N > 1 each step makes two library calls (the slice MSM and the tiny combine): the dominant
        # kernel's time is that of the big call; the combine adds ~0 to acc_points
        launches = args.steps
        acc_ms = stages["acc_points"] / launches
        achieved = BYTES_PER_PAIR * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic, traffic_source = None, None
        if world == 1 and not args.headline_only and not args.no_live_traffic:
            traffic, traffic_source = live_msm_traffic(args.log2n)
            if traffic is None:
                traffic_source = "live passes failed (%s); " % traffic_source
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if traffic is None and os.path.exists(tpath) and args.log2n == 20:    # the committed PMC passes were taken on the 2^20 workload
            try:
                traffic = json.load(open(tpath)).get("k_acc_points_bytes_per_launch")
                traffic_source = (traffic_source or "") + "static: profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload), not measured in this run"
            except Exception:
                traffic = None
        out = {
            "metric": "msm_scalar_point_pairs_per_sec", "value": world * n * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 modular integer (Fq: 10x26-bit lazy limbs, Fr: 8x32-bit limbs)", "data": "synthetic",
            "config": {"workload": f"pedersen_msm_2^{args.log2n}_secp256k1", "pairs_per_gpu": n, "window_bits": args.window or "auto",
                       "algorithm": "signed-digit Pippenger, affine in / XYZZ buckets", "sharding": f"terms/{world}" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_acc_points", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "limiter": "valu",
                         "note": "`bound` names the roofline BASELINE.json's north_star asks for (fraction of HBM peak): algorithmic 96 B/pair x 2^%d pairs per launch / "
                                 "mean k_acc_points duration (HIP events).  What binds the kernel is the integer multiplier (256-bit modular multiplications): "
                                 "`valu` below prices it against the instruction-issue bound" % args.log2n},
            "stages_ms_per_step": {k: v * calls / launches for k, v in per_call.items()},
        }
        # the bound that actually binds: modular multiplications of the accumulate kernel against the
        # measured rate of a multiply-only kernel on this chip (test hook bppp_test_mulmod_rate)
        try:
            if args.headline_only:
                raise RuntimeError("skipped (--headline-only)")
            import ctypes as C
            from bulletproofspp_amd.capi import load_test_library
            rate = C.c_double(0.0)
            if load_test_library().bppp_test_mulmod_rate(gpu.h, 2000, C.byref(rate)) == 0 and rate.value > 0 and acc_ms > 0:
                c_eff = args.window or 16
                full, r = 254 // c_eff, 255 - c_eff * (254 // c_eff)
                adds = n * (full + 1 + (0.5 if r == c_eff else 0.0))
                mm = adds * 10.0 / (acc_ms * 1e-3)
                # the issue bound of the multiplier: 1024 SIMDs x 2.4 GHz / 4.8 cycles per v_mad_u64_u32 wave-instruction (benchmarks/valu_microbench.hip)
                # x 64 lanes / 119 products per multiplication (csrc/fq26.hip.h: 100 + 19 for the 2^260 fold)
                mad_bound = 1024 * 2.4e9 / 4.8 * 64 / 119.0
                out["valu"] = {"kernel": "k_acc_points", "achieved": mm / 1e9, "peak": mad_bound / 1e9, "unit": "G mulmod/s", "frac": mm / mad_bound,
                               "multiply_only_kernel": {"value": rate.value / 1e9, "frac_of_peak": rate.value / mad_bound, "frac_of_it_achieved": mm / rate.value,
                                                        "note": "bppp_test_mulmod_rate: independent fq_mul chains, 8 waves/SIMD, measured in this run"},
                               "note": "≈ %.1f M mixed additions per launch x 10 field multiplications (8M+2S) / mean kernel duration, against the v_mad_u64_u32 "
                                       "issue bound (1024 SIMDs x 2.4 GHz / 4.8 cycles x 64 lanes / 119 products); a multiplication also issues ~32 other 64-bit "
                                       "instructions at the same cost (profiles/r04_fq_mul_isa_histogram.txt), which is what the multiply-only kernel shows" % (adds / 1e6)}
        except Exception as e:          # the hook is test-only; the headline does not depend on it
            out["valu"] = {"error": str(e)}
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
            per_thread = 1 << max(10, min(args.cpu_sample_log2, args.log2n) - 3)
            mt_rate, mt_cores, mt_dt = cpu_baseline_threads(sc_np, pts_np, per_thread)
            out["cpu_baseline_all_cores"] = {"value": mt_rate, "unit": "pairs/s", "cores": mt_cores, "kind": "port",
                                             "sample": f"{mt_cores} threads x {per_thread} pairs, same restatement, one slice per thread", "seconds": mt_dt}
        if msm_strong is not None:
            out["msm_strong_scaling"] = msm_strong
        if combined_check is not None:
            out["combined_check"] = combined_check
        if small is not None:
            out["msm_2_16"] = small
        if fixed is not None:
            out["msm_fixed_basis"] = fixed
        if fixed_batch is not None:
            out["msm_fixed_basis_batch"] = fixed_batch
        if concurrent is not None:
            out["concurrent"] = concurrent
        if host_call is not None:
            out["host_buffer_call"] = host_call
        if verify is not None:
            out["verify"] = verify
        if prove is not None:
            rr = prover_rows_roofline()
            if rr is not None:
                prove["roofline"] = rr                            # the prover's dominant kernel from the committed profile of the same command
            elif fixed_batch is not None:
                prove["roofline"] = fixed_batch["roofline"]       # the comb kernel of the unhinted route, timed live on the prover's shape in that leg
            out["prove"] = prove
        if verify_ip is not None:
            out["verify_ip"] = verify_ip
        if verify_binary is not None:
            out["verify_binary"] = verify_binary
        print(json.dumps(out), flush=True)
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
