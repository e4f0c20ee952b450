"""Host-side mirror of the reference's Bulletproof interface for the norm-linear argument, over the C ABI.

Names follow the reference (src/Bulletproof.hs, src/Bulletproof/NormArgument.hs) so the parity tests read like
the reference's own call sites: makeNormLinearBP / makeScalarsComs / collapse / getWitness / proveRoundM /
proveBPM / verifyBPM.  The vectors and bases live in HBM (`bppp_nl`); the injected oracle (src/ZKP.hs:73-77)
runs on the host, exactly one call per round."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .capi import (Bppp, BpppError, _ptr, array_to_point, array_to_scalars, int_to_limbs, limbs_to_int, points_to_array,
                   scalars_to_array)

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
Point = Optional[Tuple[int, int]]


class NormLinearBP:
    """PedersenScalarVector (NormLinear []) on the device: scalar s on g + norm (x, G) + linear (c, x, H)."""

    def __init__(self, gpu: Bppp, s: int, g: Point, q: int, cs: Sequence[int], nss: Sequence[int], ngs: Sequence[Point],
                 lss: Sequence[int], lgs: Sequence[Point]):
        # zipWithDef'' padding of makeNorm / makeLinear (NormArgument.hs:47-48, :98-99): shorter side padded with 0 / zeroV
        nlen, llen = max(len(nss), len(ngs)), max(len(cs), len(lss), len(lgs))
        pad = lambda xs, n, z: list(xs) + [z] * (n - len(xs))
        self.gpu, self.g = gpu, g
        h = C.c_void_p()
        rc = gpu.lib.bppp_nl_create(gpu.h, _ptr(int_to_limbs(s % N_ORDER)), _ptr(points_to_array([g])), _ptr(int_to_limbs(q % N_ORDER)),
                                    _ptr(scalars_to_array(pad(nss, nlen, 0))), _ptr(points_to_array(pad(ngs, nlen, None))), nlen,
                                    _ptr(scalars_to_array(pad(cs, llen, 0))), _ptr(scalars_to_array(pad(lss, llen, 0))),
                                    _ptr(points_to_array(pad(lgs, llen, None))), llen, C.byref(h))
        gpu._check(rc, "bppp_nl_create")
        self.h = h
        gpu._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            self.gpu.lib.bppp_nl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lengths(self) -> Tuple[int, int]:
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.gpu._check(self.gpu.lib.bppp_nl_lengths(self.h, C.byref(a), C.byref(b)), "bppp_nl_lengths")
        return int(a.value), int(b.value)

    def makeScalarsComs(self) -> Tuple[int, Point, int, Point]:
        """(sX, commit(sX·g + X-opening), sR, commit(sR·g + R-opening))  — Bulletproof.hs:348-350"""
        sX, sR = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        X, R = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_nl_round_commit(self.h, _ptr(sX), _ptr(X), _ptr(sR), _ptr(R)), "bppp_nl_round_commit")
        return limbs_to_int(sX), array_to_point(X), limbs_to_int(sR), array_to_point(R)

    def collapse(self, e: int):
        self.gpu._check(self.gpu.lib.bppp_nl_round_collapse(self.h, _ptr(int_to_limbs(e % N_ORDER))), "bppp_nl_round_collapse")

    def getWitness(self) -> Tuple[List[int], List[int]]:
        n, l = self.lengths()
        nw, lw = np.zeros((max(n, 1), 4), dtype=np.uint64), np.zeros((max(l, 1), 4), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_nl_get_witness(self.h, _ptr(nw), _ptr(lw)), "bppp_nl_get_witness")
        return array_to_scalars(nw)[:n], array_to_scalars(lw)[:l]

    def download(self) -> dict:
        n, l = self.lengths()
        nx, lc, lx = (np.zeros((max(k, 1), 4), dtype=np.uint64) for k in (n, l, l))
        ng, lh = np.zeros((max(n, 1), 8), dtype=np.uint64), np.zeros((max(l, 1), 8), dtype=np.uint64)
        s, q, nn, ln = (np.zeros(4, dtype=np.uint64) for _ in range(4))
        self.gpu._check(self.gpu.lib.bppp_nl_download(self.h, _ptr(nx), _ptr(ng), _ptr(lc), _ptr(lx), _ptr(lh), _ptr(s), _ptr(q), _ptr(nn), _ptr(ln)),
                        "bppp_nl_download")
        return {"norm_x": array_to_scalars(nx)[:n], "norm_g": [array_to_point(ng[i]) for i in range(n)], "lin_c": array_to_scalars(lc)[:l],
                "lin_x": array_to_scalars(lx)[:l], "lin_h": [array_to_point(lh[i]) for i in range(l)], "s": limbs_to_int(s), "q": limbs_to_int(q),
                "norm_n": limbs_to_int(nn), "lin_n": limbs_to_int(ln)}


OracleFn = Callable[[List[Point]], int]


def proveRoundM(com: NormLinearBP, oracle: OracleFn) -> Tuple[Tuple[Point, Point], int]:
    """src/Bulletproof.hs:346-355: commit X and R on the GPU, hash on the host, collapse on the GPU."""
    _, ac, _, bc = com.makeScalarsComs()
    e = oracle([ac, bc]) % N_ORDER
    com.collapse(e)
    return (ac, bc), e


def proveBPM(n_rounds: int, com: NormLinearBP, oracle: OracleFn):
    """src/Bulletproof.hs:357-359: responses (and the challenges) come out LAST ROUND FIRST."""
    resps, es = [], []
    for _ in range(n_rounds):
        r, e = proveRoundM(com, oracle)
        resps.insert(0, r)
        es.insert(0, e)
    return resps, es


ORACLE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64))


def _wrap_oracle(fn: OracleFn):
    """adapts a Python oracle (list of points, newest first -> scalar) to bppp_oracle_fn"""
    def cb(_user, pts, n, out):
        arr = np.ctypeslib.as_array(pts, shape=(n, 8))
        e = fn([array_to_point(arr[i]) for i in range(n)]) % N_ORDER
        for k_, v in enumerate(int_to_limbs(e)):
            out[k_] = int(v)
    return ORACLE_CB(cb)


def proveBPM_native(n_rounds: int, com: NormLinearBP, oracle_whole_transcript: OracleFn, transcript: Sequence[Point] = ()):
    """src/Bulletproof.hs:357-359 through bppp_nl_prove: the round loop runs in the library (C++), calling back into the
    injected oracle once per round with the whole transcript (newest first), exactly as ZKPT does (src/ZKP.hs:96-101)."""
    gpu = com.gpu
    cap = len(transcript) + 2 * n_rounds + 2
    tr = np.zeros((cap, 8), dtype=np.uint64)
    if len(transcript):
        tr[:len(transcript)] = points_to_array(list(transcript))
    ntr = C.c_size_t(len(transcript))
    resp = np.zeros((max(n_rounds, 1), 16), dtype=np.uint64)
    es = np.zeros((max(n_rounds, 1), 4), dtype=np.uint64)
    cb = _wrap_oracle(oracle_whole_transcript)
    rc = gpu.lib.bppp_nl_prove(com.h, n_rounds, C.cast(cb, C.c_void_p), None, _ptr(tr), C.byref(ntr), cap, _ptr(resp), _ptr(es))
    gpu._check(rc, "bppp_nl_prove")
    resps = [(array_to_point(resp[i, :8]), array_to_point(resp[i, 8:])) for i in range(n_rounds)]
    return resps, array_to_scalars(es)[:n_rounds], [array_to_point(tr[i]) for i in range(ntr.value)]


def verifyBPM(gpu: Bppp, q: int, sp: int, g: Point, pub_norm: Sequence[int], ngs: Sequence[Point], pub_lin_c: Sequence[int],
              pub_lin_x: Sequence[int], lgs: Sequence[Point], es: Sequence[int], responses: Sequence[Tuple[Point, Point]],
              wit_norm: Sequence[int], wit_lin: Sequence[int], init_terms: Sequence[Tuple[int, Point]]) -> bool:
    """src/Bulletproof.hs:370-378 with the challenges already derived (es, last round first): zeroV == commit(...)"""
    k = len(es)
    assert len(responses) == k
    out = np.zeros(8, dtype=np.uint64)
    flat = [p for xr in responses for p in xr]
    rc = gpu.lib.bppp_nl_verify(
        gpu.h, _ptr(int_to_limbs(q % N_ORDER)), _ptr(int_to_limbs(sp % N_ORDER)), _ptr(points_to_array([g])),
        _ptr(scalars_to_array(pub_norm)) if len(pub_norm) else None, _ptr(points_to_array(ngs)) if len(ngs) else None, len(ngs),
        _ptr(scalars_to_array(pub_lin_c)) if len(lgs) else None, _ptr(scalars_to_array(pub_lin_x)) if len(lgs) else None,
        _ptr(points_to_array(lgs)) if len(lgs) else None, len(lgs),
        _ptr(scalars_to_array(es)) if k else None, k, _ptr(scalars_to_array(wit_norm)) if len(wit_norm) else None, len(wit_norm),
        _ptr(scalars_to_array(wit_lin)) if len(wit_lin) else None, len(wit_lin),
        _ptr(scalars_to_array([s for s, _ in init_terms])) if init_terms else None,
        _ptr(points_to_array([p for _, p in init_terms])) if init_terms else None, len(init_terms),
        _ptr(points_to_array(flat)) if k else None, _ptr(out))
    gpu._check(rc, "bppp_nl_verify")
    return array_to_point(out) is None


def verifyBatch(gpu: Bppp, proofs: Sequence[dict], g: Point, ngs: Sequence[Point], lgs: Sequence[Point], rhos: Sequence[int]) -> bool:
    """Batch verifier (no reference implementation): one MSM for all proofs.  Each proof is a dict with keys
    q, sp, pub_norm, pub_lin_c, pub_lin_x, es, responses, wit_norm, wit_lin, init_terms (shapes equal across the batch)."""
    B = len(proofs)
    if B == 0:
        return True
    if len(rhos) != B or any(r % N_ORDER == 0 for r in rhos):
        raise ValueError("verifyBatch needs one non-zero weight rho per proof")
    p0 = proofs[0]
    nlen, llen, k = len(ngs), len(lgs), len(p0["es"])
    fn, fl, ninit = len(p0["wit_norm"]), len(p0["wit_lin"]), len(p0["init_terms"])
    # every proof must have the shape of the first one: the device arrays are [batch][...] with those strides, a short list would
    # make them smaller than the kernels assume
    want = {"pub_norm": nlen, "pub_lin_c": llen, "pub_lin_x": llen, "es": k, "responses": k, "wit_norm": fn, "wit_lin": fl, "init_terms": ninit}
    for p in proofs:
        if any(len(p[key]) != n for key, n in want.items()):
            return False
    cat_s = lambda key: np.concatenate([scalars_to_array(p[key]) for p in proofs]) if len(p0[key]) else np.zeros((1, 4), dtype=np.uint64)
    arrs = {
        "g": points_to_array([g]), "G": points_to_array(ngs) if nlen else np.zeros((1, 8), dtype=np.uint64),
        "H": points_to_array(lgs) if llen else np.zeros((1, 8), dtype=np.uint64), "rho": scalars_to_array([r % N_ORDER for r in rhos]),
        "q": scalars_to_array([p["q"] % N_ORDER for p in proofs]), "sp": scalars_to_array([p["sp"] % N_ORDER for p in proofs]),
        "pub_norm": cat_s("pub_norm"), "pub_lin_c": cat_s("pub_lin_c"), "pub_lin_x": cat_s("pub_lin_x"), "es": cat_s("es"),
        "wit_norm": cat_s("wit_norm"), "wit_lin": cat_s("wit_lin"),
        "init_s": np.concatenate([scalars_to_array([s for s, _ in p["init_terms"]]) for p in proofs]) if ninit else np.zeros((1, 4), dtype=np.uint64),
        "init_p": np.concatenate([points_to_array([q_ for _, q_ in p["init_terms"]]) for p in proofs]) if ninit else np.zeros((1, 8), dtype=np.uint64),
        "resp": np.concatenate([points_to_array([q_ for xr in p["responses"] for q_ in xr]) for p in proofs]) if k else np.zeros((1, 8), dtype=np.uint64),
    }
    dev = {name: gpu.to_device(a) for name, a in arrs.items()}
    out = np.zeros(8, dtype=np.uint64)
    try:
        rc = gpu.lib.bppp_nl_verify_batch_device(gpu.h, B, nlen, llen, k, fn, fl, ninit, _ptr(dev["g"]), _ptr(dev["G"]), _ptr(dev["H"]), _ptr(dev["rho"]),
                                                 _ptr(dev["q"]), _ptr(dev["sp"]), _ptr(dev["pub_norm"]), _ptr(dev["pub_lin_c"]), _ptr(dev["pub_lin_x"]),
                                                 _ptr(dev["es"]), _ptr(dev["wit_norm"]), _ptr(dev["wit_lin"]), _ptr(dev["init_s"]), _ptr(dev["init_p"]),
                                                 _ptr(dev["resp"]), _ptr(out))
        gpu._check(rc, "bppp_nl_verify_batch_device")
    finally:
        for p in dev.values():
            gpu.free(p)
    return array_to_point(out) is None


# ----------------------------------------------------------------------------- inner-product flavour
class NormLinearIP:
    """IP-flavour NormLinear (src/Bulletproof/InnerProductArgument.hs:239-267) on the device; `r` as in makeNorm (:194)."""

    def __init__(self, gpu: Bppp, s: int, g: Point, r: int, cs: Sequence[int], nss: Sequence[int], ngs: Sequence[Point],
                 lss: Sequence[int], lgs: Sequence[Point]):
        nlen, llen = max(len(nss), len(ngs)), max(len(cs), len(lss), len(lgs))
        pad = lambda xs, n, z: list(xs) + [z] * (n - len(xs))
        self.gpu = gpu
        h = C.c_void_p()
        rc = gpu.lib.bppp_ip_create(gpu.h, _ptr(int_to_limbs(s % N_ORDER)), _ptr(points_to_array([g])), _ptr(int_to_limbs(r % N_ORDER)),
                                    _ptr(scalars_to_array(pad(nss, nlen, 0))) if nlen else None, _ptr(points_to_array(pad(ngs, nlen, None))) if nlen else None, nlen,
                                    _ptr(scalars_to_array(pad(cs, llen, 0))) if llen else None, _ptr(scalars_to_array(pad(lss, llen, 0))) if llen else None,
                                    _ptr(points_to_array(pad(lgs, llen, None))) if llen else None, llen, C.byref(h))
        gpu._check(rc, "bppp_ip_create")
        self.h = h
        gpu._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            self.gpu.lib.bppp_ip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lengths(self) -> Tuple[int, int]:
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.gpu._check(self.gpu.lib.bppp_ip_lengths(self.h, C.byref(a), C.byref(b)), "bppp_ip_lengths")
        return int(a.value), int(b.value)

    def makeScalarsComs(self) -> Tuple[int, Point, int, Point]:
        sL, sR = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        L, R = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_ip_round_commit(self.h, _ptr(sL), _ptr(L), _ptr(sR), _ptr(R)), "bppp_ip_round_commit")
        return limbs_to_int(sL), array_to_point(L), limbs_to_int(sR), array_to_point(R)

    def collapse(self, e: int):
        self.gpu._check(self.gpu.lib.bppp_ip_round_collapse(self.h, _ptr(int_to_limbs(e % N_ORDER))), "bppp_ip_round_collapse")

    def getWitness(self) -> Tuple[List[int], List[int], int]:
        m, l = self.lengths()
        nw, lw = np.zeros((max(2 * m, 1), 4), dtype=np.uint64), np.zeros((max(l, 1), 4), dtype=np.uint64)
        s = np.zeros(4, dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_ip_get_witness(self.h, _ptr(nw), _ptr(lw), _ptr(s)), "bppp_ip_get_witness")
        return array_to_scalars(nw)[:2 * m], array_to_scalars(lw)[:l], limbs_to_int(s)


def verifyBPM_IP(gpu: Bppp, r: int, sp: int, g: Point, pub_norm: Sequence[int], ngs: Sequence[Point], pub_lin_c: Sequence[int],
                 pub_lin_x: Sequence[int], lgs: Sequence[Point], es: Sequence[int], responses: Sequence[Tuple[Point, Point]],
                 wit_norm: Sequence[int], wit_lin: Sequence[int], init_terms: Sequence[Tuple[int, Point]]) -> bool:
    k = len(es)
    out = np.zeros(8, dtype=np.uint64)
    flat = [p for xr in responses for p in xr]
    opt_s = lambda xs: _ptr(scalars_to_array(xs)) if len(xs) else None
    opt_p = lambda ps: _ptr(points_to_array(ps)) if len(ps) else None
    rc = gpu.lib.bppp_ip_verify(gpu.h, _ptr(int_to_limbs(r % N_ORDER)), _ptr(int_to_limbs(sp % N_ORDER)), _ptr(points_to_array([g])),
                                opt_s(pub_norm), opt_p(ngs), len(ngs), opt_s(pub_lin_c), opt_s(pub_lin_x), opt_p(lgs), len(lgs),
                                opt_s(es), k, opt_s(wit_norm), len(wit_norm), opt_s(wit_lin), len(wit_lin),
                                opt_s([s for s, _ in init_terms]), opt_p([p for _, p in init_terms]), len(init_terms), opt_p(flat), _ptr(out))
    gpu._check(rc, "bppp_ip_verify")
    return array_to_point(out) is None


# ----------------------------------------------------------------------------- lockstep batch prover
class NormLinearBatch:
    """`batch` NormLinear arguments of one shape proved in lockstep on the device (bppp_nlb_*)."""

    def __init__(self, gpu: Bppp, ss: Sequence[int], g: Point, qs: Sequence[int], cs: Sequence[Sequence[int]], nss: Sequence[Sequence[int]],
                 ngs: Sequence[Point], lss: Sequence[Sequence[int]], lgs: Sequence[Point]):
        B, nlen, llen = len(ss), len(ngs), len(lgs)
        self.gpu, self.B = gpu, B
        cat = lambda rows, n: np.concatenate([scalars_to_array(list(r) + [0] * (n - len(r))) for r in rows]) if n else None
        h = C.c_void_p()
        rc = gpu.lib.bppp_nlb_create(gpu.h, B, _ptr(scalars_to_array([s % N_ORDER for s in ss])), _ptr(points_to_array([g])),
                                     _ptr(scalars_to_array([q % N_ORDER for q in qs])), _ptr(cat(nss, nlen)), _ptr(points_to_array(ngs)) if nlen else None, nlen,
                                     _ptr(cat(cs, llen)), _ptr(cat(lss, llen)), _ptr(points_to_array(lgs)) if llen else None, llen, C.byref(h))
        gpu._check(rc, "bppp_nlb_create")
        self.h = h
        gpu._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            self.gpu.lib.bppp_nlb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lengths(self):
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self.gpu._check(self.gpu.lib.bppp_nlb_lengths(self.h, C.byref(a), C.byref(b), C.byref(c)), "bppp_nlb_lengths")
        return int(a.value), int(b.value), int(c.value)

    def makeScalarsComs(self):
        B = self.B
        sX, sR = np.zeros((B, 4), dtype=np.uint64), np.zeros((B, 4), dtype=np.uint64)
        X, R = np.zeros((B, 8), dtype=np.uint64), np.zeros((B, 8), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_nlb_round_commit(self.h, _ptr(sX), _ptr(X), _ptr(sR), _ptr(R)), "bppp_nlb_round_commit")
        return array_to_scalars(sX), [array_to_point(X[b]) for b in range(B)], array_to_scalars(sR), [array_to_point(R[b]) for b in range(B)]

    def collapse(self, es: Sequence[int]):
        self.gpu._check(self.gpu.lib.bppp_nlb_round_collapse(self.h, _ptr(scalars_to_array([e % N_ORDER for e in es]))), "bppp_nlb_round_collapse")

    def getWitness(self):
        B, n, l = self.lengths()
        nw, lw, s = np.zeros((max(B * n, 1), 4), dtype=np.uint64), np.zeros((max(B * l, 1), 4), dtype=np.uint64), np.zeros((B, 4), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_nlb_get_witness(self.h, _ptr(nw), _ptr(lw), _ptr(s)), "bppp_nlb_get_witness")
        nws, lws = array_to_scalars(nw), array_to_scalars(lw)
        return [nws[b * n:(b + 1) * n] for b in range(B)], [lws[b * l:(b + 1) * l] for b in range(B)], array_to_scalars(s)


def proveBPM_batch(n_rounds: int, com: NormLinearBatch, oracles: Sequence[OracleFn]):
    """proveBPM (src/Bulletproof.hs:357-359) for every proof of the batch, one injected oracle per proof; per-proof responses
    and challenges come out last round first."""
    B = com.B
    resps, es = [[] for _ in range(B)], [[] for _ in range(B)]
    for _ in range(n_rounds):
        _, X, _, R = com.makeScalarsComs()
        e = [oracles[b]([X[b], R[b]]) % N_ORDER for b in range(B)]
        com.collapse(e)
        for b in range(B):
            resps[b].insert(0, (X[b], R[b]))
            es[b].insert(0, e[b])
    return resps, es
