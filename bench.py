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


SHAPES = {   # (nrmLen, linLen, rounds, final norm, final lin, transcript commitments)
    "64by64": (512, 261, 8, 2, 2, 68),              # examples/64by64: 64 values, base 256 shared, NL argument
    "128by64+typed": (1152, 261, 9, 3, 1, 132),     # examples/128by64 with "typed": true (BASELINE config 4)
}


def bench_verify(gpu, torch, dev, rank, world, dist, combine, batch: int, n_real: int, steps: int, warmup: int, shape: str = "64by64",
                 cpu_baseline_leg: bool = False):
    """Secondary metric (BASELINE.json: "aggregated 64-bit range-proof verifies/sec"): batch verification of `batch`
    aggregated range proofs of the examples/64by64 shape (64 values of 64 bits, base 256 shared digits, NL argument: nrmLen 512,
    linLen 261, 8 rounds, 68 transcript commitments + 16 responses per proof) per GPU with ONE combined MSM
    (bppp_nl_verify_batch_device).  The proofs are REAL typed-reciprocal range proofs: produced here by
    bulletproofspp_amd.rangeproof (host protocol logic; every commitment and the whole norm-linear argument on the GPU) with a
    SHA-256 stand-in oracle, `n_real` distinct ones tiled to `batch` with fresh random rho.  Timed, with the proofs (points, final
    witness scalars) and their challenges resident in HBM: the derivation of every proof's public scalars from its challenges
    (verifyTRRPM's arithmetic, TypedReciprocal.hs:449-467, as bppp_trrp_public_device), challenge expansion, shared-basis merge and
    the combined MSM.  Not timed: the Fiat-Shamir hashing that produces the challenges (the injected oracle; reported separately)."""
    from bulletproofspp_amd import rangeproof as RP
    from bulletproofspp_amd.bulletproof import N_ORDER
    from bulletproofspp_amd.capi import scalars_to_array, points_to_array, array_to_point, _ptr
    nlen, llen, k, fn, fl, ninit = SHAPES[shape]
    count, typed = (64, False) if shape == "64by64" else (128, True)
    rng = np.random.default_rng(0x64B + rank)
    # basis layout h : g : hs(llen) ++ gs(nlen) (TypedReciprocal.hs:334, :348-349), lifted on the GPU
    need = 2 + llen + nlen
    pts = None
    while pts is None or pts.shape[0] < need:
        xs = rng.integers(0, 2**64, size=(3 * need, 4), dtype=np.uint64)
        dx = torch.from_numpy(xs.view(np.int64)).to(dev)
        dp = torch.zeros((3 * need, 8), dtype=torch.int64, device=dev)
        gpu.lift_x(dx.data_ptr(), 3 * need, dp.data_ptr())
        pts = dp[(dp != 0).any(dim=1)]
    P = pts[:need].cpu().numpy().view(np.uint64)
    basis_pts = [array_to_point(P[i]) for i in range(need)]
    rand_fr = lambda n: [int.from_bytes(rng.bytes(32), "little") % N_ORDER for _ in range(n)]
    rd = RP.make_range_data(256, 0, 2**64, True, True, False)
    amount = 10000                                                     # examples/*/witness.json
    pub_vt = [(False, 0, amount * count)] if typed else []             # conservation: one public input balances the outputs
    st = RP.setup(RP.GpuBackend(gpu), basis_pts, typed, pub_vt, [rd] * count)
    assert (st.nrm_len, st.lin_len, st.rounds, st.final_lens) == (nlen, llen, k, (fn, fl)), "shape table out of date"
    g, hs, gs = st.g, st.hs, st.gs

    proofs = []
    t_prove0 = time.perf_counter()
    derive_s = 0.0
    for j in range(n_real):
        vals = [amount] * count if (typed or j == 0) else [int(v) for v in rng.integers(0, 2**63, size=count, dtype=np.uint64) * 2]
        wit = RP.witness(st, [(v, 0, bl) for v, bl in zip(vals, rand_fr(count))])
        orc = RP.sha256_oracle(b"bench%d-%d" % (rank, j))
        prf = RP.prove(st, wit, orc, RP.hash_to_scalar(b"bench rand %d-%d" % (rank, j)))
        t1 = time.perf_counter()
        ch, es_ = RP.verifier_challenges(st, prf, orc)                  # the host's whole share: the oracle calls
        derive_s += time.perf_counter() - t1
        assert len(prf.coms) == ninit and len(es_) == k
        proofs.append({"ch": ch, "es": es_, "resp": [p_ for xr in prf.responses for p_ in xr], "nw": prf.wit_nrm, "lw": prf.wit_lin, "init": list(prf.coms)})
    prove_s = (time.perf_counter() - t_prove0 - derive_s) / max(n_real, 1)
    derive_s /= max(n_real, 1)

    def tile(rows_per_proof):
        one = np.concatenate(rows_per_proof)
        reps = (batch + n_real - 1) // n_real
        return np.concatenate([one] * reps)[: batch * (one.shape[0] // n_real)]

    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    d = {
        "g": up(points_to_array([g])), "G": up(points_to_array(gs)), "H": up(points_to_array(hs)),
        "rho": up(scalars_to_array([1] + rand_fr(batch - 1))), "ch": up(tile([scalars_to_array(p["ch"]) for p in proofs])),
        # written by bppp_trrp_public_device every step (the verifier's public scalars, derived from the challenges on the GPU)
        "q": up(np.zeros((batch, 4), dtype=np.uint64)), "sp": up(np.zeros((batch, 4), dtype=np.uint64)),
        "pub_norm": up(np.zeros((batch * nlen, 4), dtype=np.uint64)), "pub_lin_c": up(np.zeros((batch * llen, 4), dtype=np.uint64)),
        "pub_lin_x": up(np.zeros((batch * llen, 4), dtype=np.uint64)), "is": up(np.zeros((batch * ninit, 4), dtype=np.uint64)),
        "es": up(tile([scalars_to_array(p["es"]) for p in proofs])), "wn": up(tile([scalars_to_array(p["nw"]) for p in proofs])),
        "wl": up(tile([scalars_to_array(p["lw"]) for p in proofs])),
        "ip": up(tile([points_to_array(p["init"]) for p in proofs])), "rp": up(tile([points_to_array(p["resp"]) for p in proofs])),
    }
    out = np.zeros(8, dtype=np.uint64)
    tabs = RP.DeviceVerifierTables(gpu, st)

    def step():
        tabs.public_device(batch, d["ch"].data_ptr(), d["q"].data_ptr(), d["sp"].data_ptr(), d["pub_norm"].data_ptr(), d["pub_lin_c"].data_ptr(),
                           d["is"].data_ptr())
        rc = gpu.lib.bppp_nl_verify_batch_device(gpu.h, batch, nlen, llen, k, fn, fl, ninit, *[_ptr(d[x].data_ptr()) for x in
                                                 ("g", "G", "H", "rho", "q", "sp", "pub_norm", "pub_lin_c", "pub_lin_x", "es", "wn", "wl", "is", "ip", "rp")],
                                                 _ptr(out))
        gpu._check(rc, "bppp_nl_verify_batch_device")
        part = array_to_point(out)
        return part if world == 1 else combine(part)

    for _ in range(warmup):
        res = step()
    assert res is None, "batch of valid proofs did not verify"
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
    terms = nlen + llen + 1 + batch * (ninit + 2 * k)
    tabs.close()
    # algorithmic bytes per proof: (ninit + 2k) per-proof pairs x 96 B + (nlen + llen + 1) shared-basis scalars x 32 B
    bytes_per_proof = (ninit + 2 * k) * 96 + (nlen + llen + 1) * 32
    res_d = {"metric": "aggregated_64bit_range_proof_verifies_per_sec", "value": world * batch * steps / dt, "unit": "verifies/s",
            "ms_per_batch": dt / steps * 1e3, "batch_per_gpu": batch, "combined_msm_terms": terms,
            "algorithmic_bytes_per_proof": bytes_per_proof,
            "achieved_GBps": world * batch * steps * bytes_per_proof / dt / 1e9, "hbm_frac": world * batch * steps * bytes_per_proof / dt / 1e9 / (HBM_PEAK_GBS * world),
            "shape": f"{shape}: nrmLen {nlen}, linLen {llen}, {k} rounds, {ninit}+{2 * k} per-proof points (SURVEY.md App. B)",
            "proofs": f"{n_real} real range proofs ({count} x 64-bit values each{', typed/conserved' if typed else ''}) from the GPU-backed prover, "
                      f"tiled to {batch}; all verify (combined MSM = infinity)",
            "scope": "verifyM of RangeProof (src/RangeProof.hs:103-105) from the challenges on: public scalars (verifyTRRPM's arithmetic, "
                     "bppp_trrp_public_device) + challenge expansion + shared-basis merge + the combined MSM, all timed, all on the GPU; the "
                     "Fiat-Shamir hashing (injected oracle, host Python stand-in) is reported as host_hash_ms_per_proof",
            "gpu_prove_ms_per_proof": prove_s * 1e3, "host_hash_ms_per_proof": derive_s * 1e3}
    if cpu_baseline_leg and rank == 0:
        # the reference verifies ONE proof with ONE 256-row Straus MSM over nlen + llen + 1 + ninit + 2k terms
        # (src/Bulletproof.hs:377): time the oracle's restatement of that on this host (single thread)
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        ec = pyoracle.CEC()
        nterm = nlen + llen + 1 + ninit + 2 * k
        sc = rng.integers(0, 2**64, size=(nterm, 4), dtype=np.uint64)
        sc[:, 3] = np.minimum(sc[:, 3], np.uint64(0xFFFFFFFFFFFFFFFD))
        pts_np = np.ascontiguousarray(np.concatenate([points_to_array(gs), points_to_array(hs), points_to_array([g]),
                                                      np.concatenate([points_to_array(p["init"]) for p in proofs[:1]]),
                                                      np.concatenate([points_to_array(p["resp"]) for p in proofs[:1]])]))
        u64p = ctypes.POINTER(ctypes.c_uint64)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ec.inner_product_raw(sc.ctypes.data_as(u64p), pts_np.ctypes.data_as(u64p), nterm)
        cdt = (time.perf_counter() - t0) / reps
        res_d["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "verifies/s", "cores": 1, "kind": "port",
                                 "sample": f"{reps} single-proof verifier MSMs of {nterm} terms (the reference's one commit per verify, Bulletproof.hs:377) "
                                           "through oracle/bppp_oracle.c's 256-row Straus restatement"}
    return res_d


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
            "sample": f"{reps} arguments of the same shape, proveBPM through the oracle restatement (Straus commits, 129-row pair folds), single thread"}


def bench_prove(gpu, torch, dev, rank, batch: int, steps: int, shape: str = "64by64", pipelines: int = 2, world: int = 1, dist=None):
    """Lockstep batch prover (bppp_nlb_*): `batch` norm-linear arguments of the examples/64by64 shape advanced round by round
    together (proveBPM, src/Bulletproof.hs:357-359).  The injected oracle is a SHA-256 stand-in over the raw 128 bytes of each
    proof's (X, R) chained with that proof's previous digest; hashing runs on the host inside the timed region (it is part of
    a prover's round trip), everything else on the GPU.  The batch is split over `pipelines` contexts (one HIP stream and
    one host thread each): a round's host share (challenge hashing, half-GCDs) and its latency-bound window combine
    of one part overlap the other part's kernels."""
    import ctypes as C
    import hashlib
    import threading
    from bulletproofspp_amd.bulletproof import N_ORDER
    from bulletproofspp_amd.capi import Bppp, _ptr
    nlen, llen, k, fn, fl, _ = SHAPES[shape]
    rng = np.random.default_rng(0x9E0 + rank)
    need = 1 + llen + nlen
    pts = None
    while pts is None or pts.shape[0] < need:
        xs = rng.integers(0, 2**64, size=(3 * need, 4), dtype=np.uint64)
        dx = torch.from_numpy(xs.view(np.int64)).to(dev)
        dp = torch.zeros((3 * need, 8), dtype=torch.int64, device=dev)
        gpu.lift_x(dx.data_ptr(), 3 * need, dp.data_ptr())
        pts = dp[(dp != 0).any(dim=1)]
    P = np.ascontiguousarray(pts[:need].cpu().numpy().view(np.uint64))
    rnd_fr = lambda shape_: np.minimum(rng.integers(0, 2**64, size=shape_ + (4,), dtype=np.uint64), np.uint64(0xFFFFFFFFFFFFFFFD))
    xs_, ls_, cs_, qs_ = rnd_fr((batch, nlen)), rnd_fr((batch, llen)), rnd_fr((batch, llen)), rnd_fr((batch,))
    ss_ = rnd_fr((batch,))        # the scalar on g only shifts the commitments; any value exercises the same work
    lib = gpu.lib
    pipelines = max(1, min(pipelines, batch // 64 if batch >= 128 else 1))
    ctxs = [gpu] + [Bppp(gpu_device(dev)) for _ in range(pipelines - 1)]
    bounds = [batch * i // pipelines for i in range(pipelines + 1)]
    tm = {"create": 0.0, "round_commit": 0.0, "oracle_hash": 0.0, "round_collapse": 0.0, "get_witness": 0.0}
    lock = threading.Lock()
    errors = []

    def part(g, lo, hi):
        try:
            nb = hi - lo
            loc = dict.fromkeys(tm, 0.0)
            t = time.perf_counter()
            h = C.c_void_p()
            c_ = lambda a: np.ascontiguousarray(a)
            rc = lib.bppp_nlb_create(g.h, nb, _ptr(c_(ss_[lo:hi])), _ptr(P[0:1]), _ptr(c_(qs_[lo:hi])), _ptr(c_(xs_[lo:hi]).reshape(-1, 4)),
                                     _ptr(P[1 + llen:1 + llen + nlen]), nlen, _ptr(c_(cs_[lo:hi]).reshape(-1, 4)), _ptr(c_(ls_[lo:hi]).reshape(-1, 4)),
                                     _ptr(P[1:1 + llen]), llen, C.byref(h))
            g._check(rc, "bppp_nlb_create")
            sX, sR = np.zeros((nb, 4), dtype=np.uint64), np.zeros((nb, 4), dtype=np.uint64)
            X, R = np.zeros((nb, 8), dtype=np.uint64), np.zeros((nb, 8), dtype=np.uint64)
            digests = [b"bppp%d" % b for b in range(lo, hi)]
            t2 = time.perf_counter(); loc["create"] += t2 - t; t = t2
            for _ in range(k):
                g._check(lib.bppp_nlb_round_commit(h, _ptr(sX), _ptr(X), _ptr(sR), _ptr(R)), "bppp_nlb_round_commit")
                t2 = time.perf_counter(); loc["round_commit"] += t2 - t; t = t2
                xb, rb = X.tobytes(), R.tobytes()
                eb = []
                for b in range(nb):
                    d = digests[b] = hashlib.sha256(digests[b] + xb[64 * b:64 * b + 64] + rb[64 * b:64 * b + 64]).digest()
                    eb.append((int.from_bytes(d, "little") % N_ORDER).to_bytes(32, "little"))
                es = np.frombuffer(b"".join(eb), dtype=np.uint64).reshape(nb, 4)
                t2 = time.perf_counter(); loc["oracle_hash"] += t2 - t; t = t2
                g._check(lib.bppp_nlb_round_collapse(h, _ptr(es)), "bppp_nlb_round_collapse")
                t2 = time.perf_counter(); loc["round_collapse"] += t2 - t; t = t2
            nw, lw, s = np.zeros((nb * fn, 4), dtype=np.uint64), np.zeros((nb * fl, 4), dtype=np.uint64), np.zeros((nb, 4), dtype=np.uint64)
            g._check(lib.bppp_nlb_get_witness(h, _ptr(nw), _ptr(lw), _ptr(s)), "bppp_nlb_get_witness")
            lib.bppp_nlb_destroy(h)
            loc["get_witness"] += time.perf_counter() - t
            with lock:
                for kk in tm:
                    tm[kk] += loc[kk] / pipelines
        except Exception as e:      # surfaced by one_batch
            errors.append(e)

    def one_batch():
        th = [threading.Thread(target=part, args=(ctxs[i], bounds[i], bounds[i + 1])) for i in range(1, pipelines)]
        for t_ in th:
            t_.start()
        part(ctxs[0], bounds[0], bounds[1])
        for t_ in th:
            t_.join()
        if errors:
            raise errors[0]

    one_batch()
    torch.cuda.synchronize()
    for kk in tm:
        tm[kk] = 0.0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_batch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:                       # the prover does not shard (sequential challenges): N independent replicas, max time over ranks
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    for g in ctxs[1:]:
        g.close()
    return {"metric": "norm_linear_arguments_proved_per_sec", "value": world * batch * steps / dt, "unit": "proofs/s", "ms_per_batch": dt / steps * 1e3,
            "batch_per_gpu": batch, "replicas": world, "pipelines": pipelines, "rounds": k, "shape": f"{shape}: nrmLen {nlen}, linLen {llen}",
            "host_call_ms_per_batch": {kk: v / steps * 1e3 for kk, v in tm.items()},
            "note": "lockstep batch prover (bppp_nlb_*): 2*batch round commitments per round as one batched MSM, all basis folds as one launch; "
                    "host SHA-256 stand-in oracle inside the timed region; state upload + final opening download included; "
                    "host_call_ms_per_batch is the mean over pipelines of the wall time inside each call"}


def gpu_device(dev) -> int:
    return dev.index if getattr(dev, "index", None) is not None else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--window", type=int, default=0, help="Pippenger window bits (0 = library heuristic)")
    ap.add_argument("--cpu-sample-log2", type=int, default=17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-batch", type=int, default=4096, help="proofs per GPU in the batch-verify leg (0 = skip)")
    ap.add_argument("--verify-real", type=int, default=16, help="distinct real proofs generated by the GPU prover")
    ap.add_argument("--msm-streams", type=int, default=2, help="extra leg: MSMs in flight on that many contexts (1 = skip; N = 1 only)")
    ap.add_argument("--prove-pipelines", type=int, default=2, help="contexts (stream + host thread) the prover batch is split over")
    ap.add_argument("--prove-batch", type=int, default=2048, help="proofs advanced in lockstep per GPU in the prover leg (0 = skip); N > 1 runs N independent replicas")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
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
    stages, calls = gpu.profile_read(reset=True)
    gpu.profile_enable(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # throughput with several MSMs in flight (one context = one stream + one host thread each): the latency-bound stages of one
    # (bucket reduction, window combine, the host round trip) overlap the accumulate kernel of another.  Reported beside the
    # single-stream headline, whose per-kernel durations are what the roofline and the rocprof summaries refer to.
    concurrent = None
    if world == 1 and args.msm_streams > 1:
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
    if world == 1:
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

    verify = None
    if args.verify_batch > 0:
        vsteps = max(3, args.steps // 2)
        verify = bench_verify(gpu, torch, dev, rank, world, dist, combine, args.verify_batch, args.verify_real, vsteps, 1, "64by64",
                              cpu_baseline_leg=(world == 1 and not args.no_cpu_baseline))
        verify["other_shapes"] = [bench_verify(gpu, torch, dev, rank, world, dist, combine, max(1, args.verify_batch // 2), max(2, args.verify_real // 4),
                                               vsteps, 1, "128by64+typed")]

    prove = None
    if args.prove_batch > 0:
        prove = bench_prove(gpu, torch, dev, rank, args.prove_batch, 2, pipelines=args.prove_pipelines, world=world, dist=dist)
        if world == 1 and not args.no_cpu_baseline:
            prove["cpu_baseline"] = prove_cpu_baseline("64by64")

    if rank == 0:
        per_call = {k: v / max(calls, 1) for k, v in stages.items()}
        # with N > 1 each step makes two library calls (the slice MSM and the tiny combine): the dominant
        # kernel's time is that of the big call; the combine adds ~0 to acc_points
        launches = args.steps
        acc_ms = stages["acc_points"] / launches
        achieved = BYTES_PER_PAIR * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.log2n == 20:    # the PMC passes were taken on the 2^20 workload
            try:
                traffic = json.load(open(tpath)).get("k_acc_points_bytes_per_launch")
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic 96 B/pair x 2^%d pairs per launch / mean k_acc_points duration (HIP events); "
                                 "the kernel is VALU-bound (256-bit modular multiplies), see DESIGN.md" % args.log2n},
            "stages_ms_per_step": {k: v * calls / launches for k, v in per_call.items()},
        }
        # the bound that actually binds: modular multiplications of the accumulate kernel against the
        # measured rate of a multiply-only kernel on this chip (test hook bppp_test_mulmod_rate)
        try:
            import ctypes as C
            from bulletproofspp_amd.capi import load_test_library
            rate = C.c_double(0.0)
            if load_test_library().bppp_test_mulmod_rate(gpu.h, 2000, C.byref(rate)) == 0 and rate.value > 0 and acc_ms > 0:
                c_eff = args.window or 16
                full, r = 254 // c_eff, 255 - c_eff * (254 // c_eff)
                adds = n * (full + 1 + (0.5 if r == c_eff else 0.0))
                mm = adds * 10.0 / (acc_ms * 1e-3)
                out["valu"] = {"kernel": "k_acc_points", "achieved": mm / 1e9, "peak": rate.value / 1e9, "unit": "G mulmod/s",
                               "frac": mm / rate.value,
                               "note": "≈ %.1f M mixed additions per launch x 10 field multiplications (8M+2S) / mean kernel duration, "
                                       "against a multiply-only Fq kernel at 8 waves/SIMD measured in this run" % (adds / 1e6)}
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
        if concurrent is not None:
            out["concurrent"] = concurrent
        if host_call is not None:
            out["host_buffer_call"] = host_call
        if verify is not None:
            out["verify"] = verify
        if prove is not None:
            out["prove"] = prove
        print(json.dumps(out), flush=True)
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
