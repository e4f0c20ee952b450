"""Binary range proofs (RangeProof.Binary of the reference, src/RangeProof/Binary.hs) over the same Backend as rangeproof.py:
host protocol logic here, every commitment and the whole norm-linear argument on the GPU (`GpuBackend`).

The protocol: digits d of every value in base 2 (one coefficient b_n for the top digit so that any width works), one digit
commitment, one blinding commitment carrying the two error terms of |bl + t d|^2_q inline, then the norm-linear argument with
linear weights [0, r t] over the two blinding generators h0, h1.

Quirks of the reference kept on purpose (they decide which proofs exist):
  * witnessBRP (:158-166) yields a witness only when `conserved` is set AND the amounts balance; an unconserved binary schema
    has no prover in the reference.
  * the reference's prover fixes the number of rounds as integerLog 2 nrmLen - 1 (:199) while its verifier and decoder use
    optimalWitnessSize (:219, RangeProof.hs:68).  They agree for examples/bin_test (6 rounds); where they differ the reference's
    own proofs do not verify.  DEVIATION: both sides here use optimalWitnessSize, which coincides with the reference on every
    shape the reference can verify.
  * makeDigits (:56-69) emits n1 + 2 digits for the single value min + 2^n1 of a power-of-two-wide range; make_digits gives that
    value the top coefficient instead (DEVIATION, one value per such range).
Parity with Haskell-produced proofs is UNPINNED (no GHC); pinned by closure and by the bin_test shape of SURVEY.md App. B.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from .rangeproof import (N, RPW, Backend, NativeRangeProofs, OracleN, Point, RandFn, RangeProof, SetupBP, Transcript, integer_log, inv, optimal_witness_size,
                         powers1)


@dataclass
class BinRangeData:
    lo: int
    hi: int
    is_output: bool
    is_assumed: bool
    base_coeffs: List[int]


def make_range_data(lo: int, hi: int, is_output: bool = False, is_assumed: bool = False) -> Optional[BinRangeData]:
    """makeRangeData (Binary.hs:48-54): coefficients b_n : 2^(n1-1) ... 1 with b_n = (max - min) - 2^n1"""
    if not (hi > lo and hi - lo < N):
        return None
    n1 = integer_log(2, hi - lo - 1)
    return BinRangeData(lo, hi, is_output, is_assumed, [(hi - lo) - 2 ** n1] + [2 ** (n1 - i) for i in range(1, n1 + 1)])


def make_digits(rd: BinRangeData, n: int) -> List[int]:
    """makeDigits (Binary.hs:56-69)"""
    if rd.is_assumed:
        return []
    n_adj = n - rd.lo
    if not (0 <= n_adj < rd.hi - rd.lo):
        raise ValueError("value outside its range")
    n1 = integer_log(2, rd.hi - rd.lo - 1)
    bn = rd.base_coeffs[0]
    # the reference takes the top coefficient only when nAdj > b_n (:63); for a power-of-two width (b_n = 2^n1) and nAdj == b_n
    # its low part then needs n1 + 1 bits and the digit list comes out one too long — here that value takes the top digit
    dn, rest = (1, n_adj - bn) if (n_adj > bn or n_adj >> n1) else (0, n_adj)
    return [dn] + [(rest >> (n1 - 1 - i)) & 1 for i in range(n1)]


def input_coeffs(cons: bool, is_os: Sequence[bool], is_as: Sequence[bool], x: int) -> List[int]:
    """inputCoeffs (Binary.hs:127-129)"""
    return [((0 if a else x2) + (((-x) if o else x) if cons else 0)) % N for o, a, x2 in zip(is_os, is_as, powers1(x * x % N, len(is_os)))]


@dataclass
class SetupBRP:
    nrm_len: int
    rds: List[BinRangeData]
    net_public: int
    conserve: bool
    g: Point
    hs: List[Point]           # [h0, h1]
    gs: List[Point]
    rounds: int
    final_lens: Tuple[int, int]
    backend: Backend
    flavour: str = "NL"
    lin_len: int = 2

    def q_powers(self, q: int, n: int) -> List[int]:
        return powers1(q * q % N if self.flavour == "NL" else (-q * q) % N, n)

    def com(self, w: RPW) -> Point:
        """commitRPW sc g lin [h0, h1] nrm gs (Binary.hs:148)"""
        return self.backend.commit([w.sc] + list(w.lin) + list(w.nrm), [self.g] + self.hs[:len(w.lin)] + self.gs[:len(w.nrm)])


def setup(backend: Backend, points: Sequence[Point], conserve: bool, rds: Sequence[BinRangeData], net_public: int, flavour: str = "NL") -> SetupBRP:
    """setupBRP (Binary.hs:143-156): points = [h, g, h0, h1] ++ gs"""
    nrm_len = sum(len(rd.base_coeffs) for rd in rds)
    if len(points) < 4 + nrm_len:
        raise ValueError("not enough basis points")
    rounds, final = optimal_witness_size(nrm_len, 2, flavour)
    return SetupBRP(nrm_len, list(rds), net_public, conserve, points[1], list(points[2:4]), list(points[4:4 + nrm_len]), rounds, final, backend, flavour)


def make_public_consts(cons: bool, net_pub: int, x: int, q0: int, q0_inv: int, rds: Sequence[BinRangeData]) -> RPW:
    """makePublicConsts (Binary.hs:73-98)"""
    x2s = powers1(x * x % N, len(rds))
    bss = [xi * b % N for xi, rd in zip(x2s, rds) if not rd.is_assumed for b in rd.base_coeffs]
    mins = [0 if rd.is_assumed else rd.lo % N for rd in rds]
    net = (-x * net_pub) % N if cons else 0
    sc = -2 * (net + sum(a * b for a, b in zip(mins, x2s))) % N
    half = inv(2)
    q2, q2i, nrm = q0, q0_inv, []
    for bx in bss:
        p = (bx * q2i - half) % N
        sc = (sc + q2 * p * p) % N
        nrm.append(p)
        q2, q2i = q2 * q0 % N, q2i * q0_inv % N
    return RPW(sc, [], nrm)


def witness(st: SetupBRP, inputs: Sequence[Tuple[int, int]]):
    """witnessBRP (Binary.hs:158-166): inputs are (amount, blinding); returns (inputs, digits)"""
    if len(inputs) != len(st.rds):
        raise ValueError("Different number of values and ranges")
    v_sum = sum((-v if rd.is_output else v) for (v, _), rd in zip(inputs, st.rds))
    if not (st.conserve and (st.net_public + v_sum) % N == 0):
        raise ValueError("binary witness needs a conserved schema whose amounts balance (Binary.hs:162-164)")
    ds = [d for (v, _), rd in zip(inputs, st.rds) for d in make_digits(rd, v)]
    return [(v % N, bl % N) for v, bl in inputs], ds


def _init_terms(st: SetupBRP, coms: Sequence[Point], x: int, t: int) -> List[Tuple[int, Point]]:
    """openWith of TranscriptBRP (Binary.hs:107-110)"""
    ic = input_coeffs(st.conserve, [rd.is_output for rd in st.rds], [rd.is_assumed for rd in st.rds], x)
    return [(2 * t * t * c % N, p) for c, p in zip(ic, coms[2:])] + [(1, coms[0]), (t % N, coms[1])]


def prove_rp(st: SetupBRP, wit, tr: Transcript):
    """proveBRPM (Binary.hs:169-204)"""
    ns, ds = wit
    n_wits = [RPW(v, [bl], []) for v, bl in ns]                       # scalarRPW' (Internal.hs:56-57)
    n_coms = st.backend.commit_rows([[w.sc] + w.lin for w in n_wits], [st.g] + st.hs[:1])    # = [st.com(w) ...], one launch
    s_bl, l_bl0 = tr.random(), tr.random()
    d_wit = RPW(s_bl, [l_bl0, 0], list(ds)); d_com = st.com(d_wit)
    q, x, r = tr.oracle([d_com] + n_coms, 3)
    r_inv = inv(r)
    q0 = st.q_powers(q, 1)[0]
    q0_inv = inv(q0)
    pub = make_public_consts(st.conserve, st.net_public, x, q0, q0_inv, st.rds)
    bls_nrm = [tr.random() for _ in range(st.nrm_len)]
    bl_bl = tr.random()
    dn = (d_wit + pub).nrm
    ws = st.q_powers(q, max(len(bls_nrm), len(dn)))
    wdot = lambda a, b: sum(w * u * v for w, u, v in zip(ws, a, b)) % N      # weightedDotZip (Utils.hs:211-215)
    bl0_sc, bl1_sc = wdot(bls_nrm, bls_nrm), 2 * wdot(bls_nrm, dn) % N      # makePolyTerms (Internal.hs:69-80)
    bl_wit = RPW(bl0_sc, [bl_bl, r_inv * (s_bl - bl1_sc) % N], bls_nrm); bl_com = st.com(bl_wit)
    t = tr.oracle([bl_com], 1)[0]
    coms = [bl_com, d_com] + n_coms
    pub1 = RPW(t * pub.sc % N, [], pub.nrm)
    acc = RPW()
    for c, nw in zip(input_coeffs(st.conserve, [rd.is_output for rd in st.rds], [rd.is_assumed for rd in st.rds], x), n_wits):
        acc = acc + nw.scale(c)
    wit1 = pub1 + d_wit + acc.scale(2 * t % N)
    bp_wit = bl_wit + wit1.scale(t)
    sbp = SetupBP(q, [0, r * t % N], pub1.scale(t), _init_terms(st, coms, x, t), st.rounds)
    return coms, sbp, bp_wit


def verify_rp(st: SetupBRP, coms: Sequence[Point], tr: Transcript) -> SetupBP:
    """verifyBRPM (Binary.hs:206-222)"""
    if len(coms) != 2 + len(st.rds):
        raise ValueError("wrong number of range-proof commitments")
    q, x, r = tr.oracle([coms[1]] + list(coms[2:]), 3)
    q0 = st.q_powers(q, 1)[0]
    t = tr.oracle([coms[0]], 1)[0]
    pub = make_public_consts(st.conserve, st.net_public, x, q0, inv(q0), st.rds)
    return SetupBP(q, [0, r * t % N], RPW(t * pub.sc % N, [], pub.nrm).scale(t), _init_terms(st, coms, x, t), st.rounds)


def prove(st: SetupBRP, wit, oracle: OracleN, rand: RandFn) -> RangeProof:
    tr = Transcript(oracle, rand)
    coms, sbp, w = prove_rp(st, wit, tr)
    resps, nw, lw = st.backend.prove_bp(st.flavour, sbp.rounds, w.sc, st.g, sbp.q, sbp.cs, w.nrm, st.gs, w.lin, st.hs, lambda xs: tr.oracle(xs, 1)[0])
    return RangeProof(coms, resps, nw, lw)


def verify(st: SetupBRP, proof: RangeProof, oracle: OracleN) -> bool:
    if len(proof.responses) != st.rounds or (len(proof.wit_nrm), len(proof.wit_lin)) != st.final_lens or len(proof.coms) != 2 + len(st.rds):
        return False
    tr = Transcript(oracle)
    sbp = verify_rp(st, proof.coms, tr)
    es: List[int] = []
    for a, b in reversed(proof.responses):
        es.insert(0, tr.oracle([a, b], 1)[0])
    pad = lambda xs, n: list(xs) + [0] * (n - len(xs))
    return st.backend.verify_bp(st.flavour, sbp.q, sbp.pub.sc, st.g, pad(sbp.pub.nrm, st.nrm_len), st.gs, sbp.cs, [0, 0], st.hs, es, list(proof.responses),
                                list(proof.wit_nrm), list(proof.wit_lin), sbp.init_terms)


def verifier_challenges(st: SetupBRP, proof: RangeProof, oracle: OracleN) -> Optional[Tuple[List[int], List[int]]]:
    """the oracle calls of verifyBRPM (Binary.hs:209, :213) and of verifyBPM (Bulletproof.hs:374): ((q, x, r, t), [e_k ... e_1])"""
    if len(proof.responses) != st.rounds or (len(proof.wit_nrm), len(proof.wit_lin)) != st.final_lens or len(proof.coms) != 2 + len(st.rds):
        return None
    tr = Transcript(oracle)
    q, x, r = tr.oracle([proof.coms[1]] + list(proof.coms[2:]), 3)
    t = tr.oracle([proof.coms[0]], 1)[0]
    es: List[int] = []
    for a, b in reversed(proof.responses):
        es.insert(0, tr.oracle([a, b], 1)[0])
    return [q, x, r, t], es


class NativeBinaryRangeProofs(NativeRangeProofs):
    """One RangeProof.Binary setup registered with the library (bppp_rp_create_binary): the same handle type and entry points as the typed
    reciprocal proofs — verify_batch* decode the reference's files, hash every transcript and decide the batch with one MSM
    (verifyBRPM + verifyBPM); prove_batch is proveBRPM + proveBPM in lockstep (norm-linear argument)."""

    def __init__(self, gpu, st: SetupBRP, oracle_tag: bytes = b"", h: Point = None):
        import ctypes as C
        from .capi import RP_ASSUMED, RP_OUTPUT, RpRange, RpShape, int_to_limbs, points_to_array
        self.gpu, self.st, self.h = gpu, st, None
        rng = (RpRange * len(st.rds))()
        for r, rd in zip(rng, st.rds):
            r.base = 2
            r.flags = (RP_OUTPUT if rd.is_output else 0) | (RP_ASSUMED if rd.is_assumed else 0)
            r.min[:] = [int(v) for v in int_to_limbs(rd.lo % 2**256)]
            r.max[:] = [int(v) for v in int_to_limbs(rd.hi % 2**256)]
        pts = points_to_array([h if h is not None else st.g, st.g] + list(st.hs) + list(st.gs))
        net = int_to_limbs(st.net_public % 2**256)
        hnd = C.c_void_p()
        rc = gpu.lib.bppp_rp_create_binary(gpu.h, 0 if st.flavour == "NL" else 1, int(st.conserve), C.cast(rng, C.c_void_p), len(st.rds), C.c_void_p(net.ctypes.data),
                                           C.c_void_p(pts.ctypes.data), pts.shape[0], oracle_tag if oracle_tag else None, C.byref(hnd))
        gpu._check(rc, "bppp_rp_create_binary")
        self.h = hnd
        gpu._adopt(self)
        shp = RpShape()
        gpu._check(gpu.lib.bppp_rp_info(self.h, C.byref(shp)), "bppp_rp_info")
        self.shape = {n: int(getattr(shp, n)) for n, _ in RpShape._fields_}
        if (self.shape["norm_len"], self.shape["lin_len"], self.shape["rounds"], (self.shape["final_norm"], self.shape["final_lin"])) != \
                (st.nrm_len, 2, st.rounds, tuple(st.final_lens)):
            raise RuntimeError("native binary setup disagrees with the host setup: %r" % (self.shape,))

    def prove_batch(self, inputs: Sequence[Sequence[Tuple[int, int]]], rand_prefixes: Sequence[bytes]) -> List[Tuple[bytes, bytes]]:
        """bppp_rp_prove_batch on a binary setup: inputs[b] = [(amount, blinding) per range]"""
        return super().prove_batch([[(v, 0, bl) for v, bl in row] for row in inputs], rand_prefixes)

    def split_challenges(self, flat: Sequence[int]):
        return list(flat[:4]), list(flat[4:])


def setup_from_schema(backend: Backend, schema: dict, points: Optional[Sequence[Point]] = None) -> SetupBRP:
    """The binary branch of the CLI's schema handling (app/Parse.hs:126-158, app/Main.hs:291-316)."""
    from .rangeproof import basis_points
    if not schema.get("binary", False):
        raise ValueError("not a binary schema")
    if schema.get("typed", False):
        raise ValueError("Can't make typed binary proof")
    arg = str(schema.get("argument", "IP")).lower()
    flavour = {"ip": "IP", "innerproduct": "IP", "nl": "NL", "normlinear": "NL"}.get(arg)
    if flavour is None:
        raise ValueError("Unsupported Argument: " + arg)
    rds: List[BinRangeData] = []
    for r in schema["ranges"]:
        if r.get("base", 2) != 2 or r.get("isShared", False):
            raise ValueError("Invalid base / shared digits for binary range proof")
        rd = make_range_data(int(r.get("min", 0)), int(r.get("max", 2**64)), bool(r.get("isOutput", False)), bool(r.get("isAssumed", False)))
        if rd is None:
            raise ValueError("Invalid range: %r" % (r,))
        rds += [rd] * int(r.get("count", 1))
    pubs = schema.get("public", [])
    if any(int(p.get("type", 0)) != 0 for p in pubs):
        raise ValueError("Cannot have type of public value in binary proof")
    net = sum((-int(p["amount"]) if p.get("isOutput", False) else int(p["amount"])) for p in pubs)
    if points is None:
        points = basis_points(str(schema["basisSeed"]).encode(), 4 + sum(len(rd.base_coeffs) for rd in rds))
    return setup(backend, points, bool(schema.get("conserved", False)), rds, net, flavour)
