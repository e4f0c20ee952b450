"""Typed reciprocal range proofs (RangeProof.TypedReciprocal + RangeProof.Internal + the generic RangeProof wrapper of the
reference) on top of the GPU norm-linear argument — SURVEY.md section 8(f) rank 1.

This is HOST protocol logic (phase 1-3 commitments' scalars, blinding algebra, public constants); everything that touches
curve points goes through a `Backend`: `commit` (the reference's `commitRPW` = one Pedersen MSM, src/RangeProof/Internal.hs:45-50)
and the norm-linear argument's prover / verifier (src/Bulletproof.hs:346-378).  `GpuBackend` is the product backend (C ABI:
bppp_msm, bppp_nl_*); there is no CPU backend in the package — the tests inject one built from `oracle/` to check that the GPU
path yields the same transcript bit for bit.

Names follow the reference so the call sites read like src/RangeProof/TypedReciprocal.hs.  Both argument flavours are wired:
"NL" (Bulletproof.NormArgument; `examples/{32by64,64by64,96by64,128by64}`) and "IP" (Bulletproof.InnerProductArgument, the
schema default, app/Parse.hs:100; `examples/{32bit,64bit,rec_test}`).  The reference cannot be run here (no GHC), so proofs are
pinned by algebraic closure (prove -> verify accepts; any tampering rejects) and by the shapes of SURVEY.md Appendix B, not by
reference-generated vectors: parity of this layer with the Haskell implementation is UNPINNED.
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141      # group order: the scalar field `s`
Point = Optional[Tuple[int, int]]
OracleN = Callable[[List[Point], int], List[int]]     # whole transcript (newest first), count -> that many challenges
RandFn = Callable[[int], int]                         # counter -> scalar (ZKPT's `h n`, src/ZKP.hs:88-92)


def decode_field(b: bytes, modulus: int) -> int:
    """`decode` through Binary (Prime p) (src/Encoding.hs:75-79): four big-endian 64-bit words, LEAST-significant word first,
    reduced by toP.  This is how the reference turns a SHA-256 digest into a field element (`hash = decode . SHA.hash`,
    app/Main.hs:64-65) and how it reads scalars and coordinates from files."""
    return sum(int.from_bytes(b[8 * i:8 * i + 8], "big") << (64 * i) for i in range(4)) % modulus


# ----------------------------------------------------------------------------- small helpers (src/Utils.hs)
def inv(a: int) -> int:
    a %= N
    return pow(a, N - 2, N) if a else 0


def batch_inverse(xs: Sequence[int]) -> List[int]:
    """batchInverse (src/Data/Field/BatchInverse.hs:18-39): Montgomery's trick, 0 -> 0."""
    pre, acc = [], 1
    for x in xs:
        pre.append(acc)
        if x % N:
            acc = acc * x % N
    y = inv(acc)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        if xs[i] % N:
            out[i] = y * pre[i] % N
            y = y * xs[i] % N
    return out


def powers1(a: int, n: int) -> List[int]:
    """first n of powers' a = [a, a^2, ...] (Utils.hs:107-108)"""
    out, c = [], 1
    for _ in range(n):
        c = c * a % N
        out.append(c)
    return out


def integer_log(b: int, n: int) -> int:
    """integerLog (Utils.hs:78-79)"""
    r = 0
    while n >= b:
        n //= b
        r += 1
    return r


def pad_right(n: int, z, xs: Sequence) -> list:
    return (list(xs) + [z] * n)[:n]


def insert_at(n: int, x, xs: Sequence) -> list:
    xs = list(xs)
    return xs[:n] + [x] + xs[n:]


def remove_at(n: int, xs: Sequence) -> list:
    xs = list(xs)
    return xs[:n] + xs[n + 1:]


# ----------------------------------------------------------------------------- RPWitness (RangeProof/Internal.hs:20-43)
@dataclass
class RPW:
    sc: int = 0
    lin: List[int] = field(default_factory=list)
    nrm: List[int] = field(default_factory=list)

    def __add__(self, o: "RPW") -> "RPW":        # (^+^) = zipWithDef'' (+) 0 0
        z = lambda a, b: [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % N for i in range(max(len(a), len(b)))]
        return RPW((self.sc + o.sc) % N, z(self.lin, o.lin), z(self.nrm, o.nrm))

    def scale(self, s: int) -> "RPW":            # (*^)
        return RPW(self.sc * s % N, [x * s % N for x in self.lin], [x * s % N for x in self.nrm])


# ----------------------------------------------------------------------------- ranges and digits (TypedReciprocal.hs:77-131)
@dataclass
class RangeData:
    base: int
    lo: int
    hi: int
    is_shared: bool
    is_output: bool
    is_assumed: bool
    has_bit: bool
    base_coeffs: List[int]


def make_range_data(base: int, lo: int, hi: int, is_shared: bool = False, is_output: bool = False, is_assumed: bool = False) -> Optional[RangeData]:
    """makeRangeData (TypedReciprocal.hs:103-120) for the field of characteristic N; None if the range is invalid."""
    if not (hi > lo and base > 1 and hi - lo < N):
        return None
    b, w = base, hi - lo
    n1 = integer_log(b, w - 1)
    has_bit = ((w - 1) % (b - 1)) != 0
    tail = [b ** (n1 - i) for i in range(1, n1 + 1)]
    if not has_bit:
        bs = [(w - b ** n1) // (b - 1)] + tail
    elif w < 2 * b ** n1:
        bs = [w - b ** n1] + tail
    else:
        bn1 = 1 + w // (2 * (b - 1)) - (b ** n1 - 1) // (b - 1)
        bs = [w - bn1 * (b - 1) - b ** n1, bn1] + tail
    return RangeData(b, lo, hi, is_shared, is_output, is_assumed, has_bit, [] if is_assumed else bs)


def digits(rd: RangeData, n: int) -> List[int]:
    """digits (TypedReciprocal.hs:125-127): greedy mixed-radix digits; the first is binary when has_bit."""
    out = []
    for i, coeff in enumerate(rd.base_coeffs):
        radix = 2 if (rd.has_bit and i == 0) else rd.base
        d = min(radix - 1, n // coeff)
        n -= d * coeff
        out.append(d)
    return out


def counts(xs: Sequence[int], ys: Sequence[int]) -> List[int]:
    m: Dict[int, int] = {}
    for y in ys:
        m[y] = m.get(y, 0) + 1
    return [m.get(x, 0) for x in xs]


# Phase1 (TypedReciprocal.hs:56-60): ("inline", idx, base, b, d, m, s) | ("shared", idx, base, b, d) | ("typing", idx, io, ia, v, t);
# private fields (d, m, v, t) are None on the verifier's side
def make_phase1s(ind: int, rd: RangeData, n: Optional[int]):
    """makePhase1s / makePhase1sVer (TypedReciprocal.hs:133-169): (phase-1 records, shared multiplicities or None)."""
    if rd.is_assumed:
        return [], None
    ver = n is None
    n_adj = 0 if ver else n - rd.lo
    if not (0 <= n_adj < rd.hi - rd.lo):
        raise ValueError("value outside its range")
    b = rd.base
    ds = digits(rd, n_adj)
    ms = ([ds[0]] + counts(range(1, b), ds[1:])) if rd.has_bit else counts(range(1, b), ds)
    ns = ([1] if rd.has_bit else []) + list(range(1, b))
    radix = lambda i: 2 if (rd.has_bit and i == 0) else b
    priv = (lambda v: None) if ver else (lambda v: v % N)
    if rd.is_shared:
        ph1s = [("shared", ind, radix(i), rd.base_coeffs[i] % N, priv(ds[i])) for i in range(min(len(rd.base_coeffs), len(ds)))]
        return ph1s, (None if ver else [m % N for m in ms])
    ln = max(len(rd.base_coeffs), len(ds), len(ms), len(ns))
    bs_, ds_, ms_, ns_ = (pad_right(ln, 0, w) for w in (rd.base_coeffs, ds, ms, ns))
    return [("inline", ind, radix(i), bs_[i] % N, priv(ds_[i]), priv(ms_[i]), ns_[i] % N) for i in range(ln)], None


def get_ds_ms(ph1s) -> Tuple[List[int], List[int]]:
    """getDsMs (TypedReciprocal.hs:74-80)"""
    ds, ms = [], []
    for p in ph1s:
        if p[0] == "inline":
            ds.append(p[4]); ms.append(p[5])
        elif p[0] == "shared":
            ds.append(p[4]); ms.append(0)
        else:
            ds.append(p[5]); ms.append(0)
    return ds, ms


# Phase2 (TypedReciprocal.hs:180-181)
@dataclass
class Ph2:
    is_t: bool
    d: Optional[int]
    m: Optional[int]
    u: int
    v: int
    r: Optional[int]
    c: int


def make_phase2s(e: int, e_inv: int, x: int, base_map: Dict[int, int], ph1s, private: bool) -> List[Ph2]:
    """makePhase2s (TypedReciprocal.hs:185-205).  private=False is the verifier's `b ~ ()` instantiation."""
    dens, ss, ps, vs, rec = [], [], [], [], []
    xpow: Dict[int, int] = {}
    for p in ph1s:
        xi = xpow.get(p[1])
        if xi is None:
            xi = xpow[p[1]] = pow(x, 2 * (p[1] + 1), N)
        if p[0] == "typing":
            _, _, io, ia, v, t = p
            x2 = (-x) % N if io else x % N
            dens.append((e + t) % N if private else 0); ss.append(0); ps.append(v if private else 0); vs.append(x2)
            rec.append((True, t, 0 if private else None, 0 if ia else xi, x2))
        elif p[0] == "inline":
            _, _, base, b, d, m, s = p
            x2 = base_map[base]
            dens.append((e + d) % N if private else 0); ss.append(0 if s == 0 else (e + s) % N); ps.append(1); vs.append(x2)
            rec.append((False, d, m, xi * b % N, x2))
        else:
            _, _, base, b, d = p
            x2 = base_map[base]
            dens.append((e + d) % N if private else 0); ss.append(0); ps.append(1); vs.append(x2)
            rec.append((False, d, 0 if private else None, xi * b % N, x2))
    rs = [a * b % N for a, b in zip(ps, batch_inverse(dens))] if private else [None] * len(ph1s)
    cs = [v * (0 if s == 0 else (e_inv - s) % N) % N for v, s in zip(vs, batch_inverse(ss))]
    return [Ph2(is_t, d, m, u, v, r, c) for (is_t, d, m, u, v), r, c in zip(rec, rs, cs)]


def make_shared_coeffs(e: int, e_inv: int, m_bases: Sequence[int], base_map: Dict[int, int]) -> List[int]:
    """makeSharedCoeffs (TypedReciprocal.hs:213-216)"""
    xs, ss = [], []
    for b in m_bases:
        for s in range(1, b):
            xs.append(base_map[b]); ss.append((e + s) % N)
    return [x * ((e_inv - si) % N) % N for x, si in zip(xs, batch_inverse(ss))]


def make_error_terms(e: int, xp: int, shared_cs: Sequence[int], bls_ms: Sequence[int], ph3s) -> List[int]:
    """makeErrorTerms (TypedReciprocal.hs:226-243); ph3s = [(Ph2, q2, bl)]"""
    tot = [0, 0, 0, 2 * sum(a * b for a, b in zip(shared_cs, bls_ms)) % N, 0, 0]
    for p, q2, bl in ph3s:
        d, m, u, v, r, c = p.d, p.m, p.u, p.v, p.r, p.c
        rC = xp * (u + q2) % N if p.is_t else u
        dC = (v + q2 * e) % N
        errs = [q2 * bl * bl,
                2 * q2 * m * bl,
                q2 * m * m + 2 * bl * (q2 * d + dC),
                2 * (bl * (q2 * r + rC) + m * (q2 * d + dC)),
                (q2 * d * d + 2 * d * dC) + 2 * (bl * c + m * (q2 * r + rC)),
                (q2 * r * r + 2 * r * rC) + 2 * c * d]
        tot = [(a + b) % N for a, b in zip(tot, errs)]
    return tot


def input_coeffs(has_types: bool, assumed: Sequence[bool], x: int, q0: int) -> List[int]:
    """inputCoeffs (TypedReciprocal.hs:325-328)"""
    xp = [0 if a else p for a, p in zip(assumed, powers1(x * x % N, len(assumed)))]
    return [(a + b) % N for a, b in zip(powers1(q0, len(assumed)), xp)] if has_types else xp


def make_bp_coeffs(has_types: bool, xp: int, r0: int, r1: int, t: int, cs: Sequence[int]) -> List[int]:
    """makeBpCoeffs (TypedReciprocal.hs:391-396)"""
    rs = r0 * r1 % N
    tp = lambda k: pow(t, k, N)
    return [(-xp) % N if has_types else 0, rs * t % N, rs * tp(2) % N, rs * tp(3) % N, r0 * tp(4) % N, rs * tp(6) % N] + \
           [2 * tp(3) * c % N for c in cs]


def make_public_consts(e: int, e_inv: int, x: int, xp: int, q0: int, q0_inv: int, t: int, has_types: bool, rds: Sequence[RangeData],
                       pub_vt: Sequence[Tuple[bool, int, int]], ph2s: Sequence[Ph2]) -> RPW:
    """makePublicConsts (TypedReciprocal.hs:246-274); pub_vt entries are (isOutput, type, amount) as destructured at :258."""
    t2, t3, t4, t5 = (pow(t, k, N) for k in (2, 3, 4, 5))
    mins = [0 if rd.is_assumed else rd.lo % N for rd in rds]
    pub_rs = batch_inverse([(e + ty) % N for _, ty, _ in pub_vt])
    pub_sum = sum(((-r * v) if is_out else (r * v)) for (is_out, _, v), r in zip(pub_vt, pub_rs)) % N
    z = -2 * t5 * sum(a * b for a, b in zip(mins, powers1(x * x % N, len(mins))))
    if has_types:
        z -= 2 * t5 * x * pub_sum
    ts0, ts1 = [], []
    for p, q2, qi2 in zip(ph2s, powers1(q0, len(ph2s)), powers1(q0_inv, len(ph2s))):
        if p.is_t:
            rC, p2C = xp * (qi2 * p.u + 1) % N, 0
        else:
            rC, p2C = qi2 * p.u % N, (2 * q2 + 2 * e_inv * p.v) % N
        pv = (t2 * (e + qi2 * p.v) + t3 * rC + t4 * (qi2 * p.c)) % N
        ts0.append((q2 * pv * pv + t5 * p2C) % N)
        ts1.append(pv)
    return RPW((z + sum(ts0)) % N, [], ts1)


# ----------------------------------------------------------------------------- blinding (RangeProof/Internal.hs:118-196)
def blind_witness(n: int, k: int, ls: Sequence[int], ns: Sequence[int], rnd: Callable[[], int]) -> RPW:
    """blindWitness (:130-139)"""
    n_bls = 2 * n - 1 if k == 1 else 2 * n - k + 1
    bls = pad_right(2 * n + 1, 0, insert_at(2 * n - k, 0, [rnd() for _ in range(n_bls)]))
    return RPW(bls[0], bls[1:] + list(ls), list(ns))


def blind_err_witness(n: int, es: Sequence[int], ls: Sequence[int], ns: Sequence[int], rnd: Callable[[], int]) -> RPW:
    """blindErrWitness (:142-149)"""
    bls = pad_right(2 * n + 1, 0, insert_at(n, 0, [rnd() for _ in range(n + 1)]) + list(es))
    return RPW(bls[0], bls[1:] + list(ls), list(ns))


def _scale_errs(n: int, s: int, xs: Sequence[int]) -> List[int]:
    """scaleErrs (:118-121)"""
    xs = list(xs)
    ys, zs = xs[:n + 1], xs[n + 1:]
    a, b = zs[:n - 2], zs[n - 2:]
    return ys + [s * v % N for v in a] + b


def _sum_diagonals(table: Sequence[Sequence[int]]) -> List[int]:
    """sumDiagonals (:104-111)"""
    m: Dict[int, int] = {}
    for a, row in enumerate(table):
        for b, v in enumerate(row):
            m[a + b] = (m.get(a + b, 0) + v) % N
    return [m[k] for k in sorted(m)]


def blind_blinding_term(bl_bls: RPW, tC: int, r0: int, r0_inv: int, r1: int, r1_inv: int, errs: Sequence[int], wits: Sequence[RPW], input_bl: int) -> RPW:
    """blindBlindingTerm (:154-196)"""
    assert bl_bls.sc == 0
    blT, bls_lin, bls_nrm = bl_bls.lin[0], bl_bls.lin[1:], bl_bls.nrm
    rs_inv = r0_inv * r1_inv % N
    n = len(wits)
    wits1, wit_err = wits[:n - 1], wits[n - 1]
    wit_err_row = [wit_err.sc] + pad_right(2 * n, 0, wit_err.lin[:n + 1])
    wit_rows = [[w.sc] + w.lin[:2 * n] for w in wits1]
    wit_rows1 = [[r[0], r[1]] + [(-v) % N for v in r[2:]] for r in wit_rows + [wit_err_row]]
    errs1 = [(-v) % N for v in [(errs[0] - tC * blT) % N] + [rs_inv * v % N for v in errs[1:]]]
    add_consts = lambda a, b, r: [(a * r[0] + b * r[1]) % N] + r[2:]
    table = [insert_at(2 * n - 1, 0, row) for row in
             [errs1] + [_scale_errs(n, r1_inv, add_consts(rs_inv, rs_inv * tC % N, r)) for r in wit_rows1]]
    bl_errs = _scale_errs(n, r1, remove_at(2 * n - 1, _sum_diagonals(table))[:2 * n])
    bl_errs[-1] = (bl_errs[-1] - 2 * input_bl) % N
    return RPW((-bl_errs[0]) % N, [blT] + bl_errs[1:] + list(bls_lin), list(bls_nrm))


# ----------------------------------------------------------------------------- the curve side: backends
class Backend:
    """What the protocol needs from the curve: the Pedersen MSM and the norm-linear argument."""

    def commit(self, scalars: Sequence[int], points: Sequence[Point]) -> Point:
        raise NotImplementedError

    def commit_rows(self, rows: Sequence[Sequence[int]], points: Sequence[Point]) -> List[Point]:
        """one commitment per row of scalars, all over the same points (the input commitments of a proof); default: one MSM each"""
        return [self.commit(r, points) for r in rows]

    def prove_bp(self, flavour: str, n_rounds: int, sc: int, g: Point, q: int, cs, nrm, gs, lin, hs, oracle1: Callable[[List[Point]], int]):
        """proveBPM on PSV(sc, g, makeNormLinearBP q cs nrm gs lin hs) of the NL or IP flavour; returns (responses last round first,
        norm witness, linear witness) = getWitness of the final opening"""
        raise NotImplementedError

    def verify_bp(self, flavour: str, q: int, sp: int, g: Point, pub_nrm, gs, cs, pub_lin, hs, es, responses, wit_nrm, wit_lin, init_terms) -> bool:
        raise NotImplementedError


class GpuBackend(Backend):
    """The product backend: every group operation runs on the MI355X through the C ABI."""

    def __init__(self, gpu):
        self.gpu = gpu

    def commit(self, scalars, points):
        from .capi import points_to_array, scalars_to_array
        if not len(scalars):
            return None
        return self.gpu.msm(scalars_to_array([s % N for s in scalars]), points_to_array(list(points)))

    def commit_rows(self, rows, points):
        """all rows in ONE batched MSM over the shared points (bppp_msm_batch_device, shared_points = 1)"""
        import numpy as np
        from .capi import points_to_array, scalars_to_array
        if not rows:
            return []
        n = len(points)
        d_s = self.gpu.to_device(np.concatenate([scalars_to_array([s % N for s in r]) for r in rows]))
        d_p = self.gpu.to_device(points_to_array(list(points)))
        try:
            return self.gpu.msm_batch_device(d_s, d_p, n, len(rows), shared_points=True)
        finally:
            self.gpu.free(d_s); self.gpu.free(d_p)

    def prove_bp(self, flavour, n_rounds, sc, g, q, cs, nrm, gs, lin, hs, oracle1):
        from .bulletproof import NormLinearBP, NormLinearIP, proveBPM
        com = (NormLinearBP if flavour == "NL" else NormLinearIP)(self.gpu, sc, g, q, cs, nrm, gs, lin, hs)
        try:
            resps, _ = proveBPM(n_rounds, com, oracle1)
            wit = com.getWitness()
        finally:
            com.close()
        return resps, wit[0], wit[1]

    def verify_bp(self, flavour, q, sp, g, pub_nrm, gs, cs, pub_lin, hs, es, responses, wit_nrm, wit_lin, init_terms):
        from .bulletproof import verifyBPM, verifyBPM_IP
        pad = lambda xs, n: list(xs) + [0] * (n - len(xs))
        fn = verifyBPM if flavour == "NL" else verifyBPM_IP
        return fn(self.gpu, q, sp, g, pad(pub_nrm, len(gs)), gs, pad(cs, len(hs)), pad(pub_lin, len(hs)), hs, es, responses, wit_nrm, wit_lin, init_terms)


# ----------------------------------------------------------------------------- transcript (src/ZKP.hs:73-101)
class Transcript:
    """ZKPT's state: all commitments so far (newest first) and the random counter."""

    def __init__(self, oracle: OracleN, rand: Optional[RandFn] = None):
        self.fn, self.rand_fn, self.cs, self.n = oracle, rand, [], 0

    def oracle(self, xs: Sequence[Point], count: int = 1) -> List[int]:
        self.cs = list(xs) + self.cs
        return [v % N for v in self.fn(self.cs, count)]

    def random(self) -> int:
        if self.rand_fn is None:
            raise RuntimeError("No Random in Verifier")        # app/Main.hs:202
        v = self.rand_fn(self.n) % N
        self.n += 1
        return v


def sha256_oracle(tag: bytes = b"") -> OracleN:
    """shaOracle (app/Main.hs:75-80): challenge n = hash (show n <> show (length ps) <> foldMap (coords . toA) ps) with
    coords (A x y) = show x <> show y and hash = decode . SHA-256 — the digest read as a field element by Binary (Prime p)
    (decode_field).  `show` of a field element is taken to be its plain decimal integer (as FastPrime's instance prints it,
    src/Data/Field/Galois/FastPrime.hs:129-130); that text format of the third-party `Prime` type cannot be confirmed offline
    (SURVEY.md 8c), so the hash INPUT is parity-unpinned while the digest decode follows the source.  `tag` (default empty =
    the reference's input) is an optional domain-separation prefix for tests that want distinct oracles.  The native
    counterpart (same bytes in, same challenge out) is csrc/sha256.hip.h + bppp_rp_* (include/bppp.h)."""
    enc: Dict[Point, bytes] = {}

    def one(p: Point) -> bytes:
        b = enc.get(p)
        if b is None:
            b = enc[p] = b"inf" if p is None else str(p[0]).encode() + str(p[1]).encode()
        return b

    def fn(cs: List[Point], count: int) -> List[int]:
        body = b"".join(one(p) for p in cs)
        return [decode_field(hashlib.sha256(tag + str(n).encode() + str(len(cs)).encode() + body).digest(), N) for n in range(1, count + 1)]
    return fn


def hash_to_scalar(prefix: bytes) -> RandFn:
    """hashToScalar rn . show (app/Main.hs:83-84, :189) — the prover's deterministic randomness: hash (prefix <> show n), the
    digest decoded by Binary (Prime p) as everywhere else (decode_field)."""
    return lambda n: decode_field(hashlib.sha256(prefix + str(n).encode()).digest(), N)


# ----------------------------------------------------------------------------- setup (TypedReciprocal.hs:332-359)
@dataclass
class SetupTRRP:
    has_types: bool
    m_bases: List[int]
    sorted_bases: List[int]
    nrm_len: int
    lin_len: int
    pub_vt: List[Tuple[bool, int, int]]
    rds: List[RangeData]
    g: Point
    hs: List[Point]
    gs: List[Point]
    rounds: int
    final_lens: Tuple[int, int]
    backend: Backend
    flavour: str = "NL"

    def base_map(self, x: int) -> Dict[int, int]:
        """makeBaseMap: sortedBases zipped with x^3, x^5, ... (powers'' (x^3) (x^2), :349)"""
        out, c = {}, pow(x, 3, N)
        for b in self.sorted_bases:
            out[b] = c
            c = c * x % N * x % N
        return out

    def q_powers(self, q: int, n: int) -> List[int]:
        """qPowers': powers' (q^2) for the NL Norm (Bulletproof/NormArgument.hs:148), powers' (-q^2) for the IP one
        (Bulletproof/InnerProductArgument.hs:231)"""
        return powers1(q * q % N if self.flavour == "NL" else (-q * q) % N, n)

    def com(self, w: RPW) -> Point:
        """commitRPW sc g lin hs nrm gs (RangeProof/Internal.hs:45-50)"""
        assert len(w.lin) <= len(self.hs) and len(w.nrm) <= len(self.gs)
        return self.backend.commit([w.sc] + list(w.lin) + list(w.nrm), [self.g] + self.hs[:len(w.lin)] + self.gs[:len(w.nrm)])


def round_reduce(n: int) -> int:
    return n // 2 + n % 2


def number_rounds_reduce(n: int) -> Tuple[int, int]:
    """numberRoundsReduce (src/Bulletproof.hs:300-304)"""
    r = 0
    while n >= 5:
        n = round_reduce(n)
        r += 1
    return r, n


def optimal_witness_size(n_len: int, l_len: int, flavour: str = "NL") -> Tuple[int, Tuple[int, int]]:
    """optimalWitnessSize of NormLinear: NL flavour Bulletproof/NormArgument.hs:165-178; IP flavour (the norm vector is paired up
    first and reduced to <= 2 pairs) Bulletproof/InnerProductArgument.hs:253-267 with numberRoundsReduce' (Bulletproof.hs:307-308)"""
    if flavour == "NL":
        nR, n1 = number_rounds_reduce(n_len)
    else:
        nR, n1 = number_rounds_reduce((n_len + n_len % 2) // 2)
        if n1 > 2:
            nR, n1 = nR + 1, round_reduce(n1)
    lR, l1 = number_rounds_reduce(l_len)
    r = max(nR, lR)
    for _ in range(r - nR):
        n1 = round_reduce(n1)
    for _ in range(r - lR):
        l1 = round_reduce(l1)
    w = 1 if flavour == "NL" else 2
    if w * n1 + l1 > 5:
        return r + 1, (w * round_reduce(n1), round_reduce(l1))
    return r, (w * n1, l1)


def setup(backend: Backend, points: Sequence[Point], has_types: bool, pub_vt: Sequence[Tuple[bool, int, int]], rds: Sequence[RangeData],
          flavour: str = "NL") -> SetupTRRP:
    """setup (TypedReciprocal.hs:332-359): points = h : g : ps (h is not used by the proof, as in the reference)."""
    live = [rd for rd in rds if not rd.is_assumed]
    any_has_bit = any(rd.has_bit for rd in live)
    any_shared_has_bit = any(rd.has_bit and rd.is_shared for rd in live)
    pairs = sorted(((rd.is_shared, rd.base) for rd in live), key=lambda p: p[1])
    m_bases = sorted(set(([2] if any_shared_has_bit else []) + [b for s, b in pairs if s]))
    sorted_bases = sorted(set(([2] if any_has_bit else []) + [b for _, b in pairs]))
    nrm_len = sum(len(rd.base_coeffs) + (1 if has_types else 0) for rd in rds)
    lin_len = 6 + sum(b - 1 for b in m_bases)
    ps = list(points[2:])
    if len(ps) < lin_len + nrm_len:
        raise ValueError("not enough basis points")
    if flavour not in ("NL", "IP"):
        raise ValueError("argument flavour must be NL or IP")
    rounds, final = optimal_witness_size(nrm_len, lin_len, flavour)
    return SetupTRRP(has_types, m_bases, sorted_bases, nrm_len, lin_len, list(pub_vt), list(rds), points[1], ps[:lin_len], ps[lin_len:lin_len + nrm_len],
                     rounds, final, backend, flavour)


# ----------------------------------------------------------------------------- witness (TypedReciprocal.hs:361-389)
@dataclass
class WitnessTRRP:
    inputs: List[Tuple[int, int, int]]       # (amount, type, blinding) per input — the PedersenScalarPair's scalars
    ph1s: list
    base_mss: List[Tuple[int, List[int]]]


def witness(st: SetupTRRP, inputs: Sequence[Tuple[int, int, int]]) -> WitnessTRRP:
    """witnessTRRP (:372-389): inputs are (amount, type, blinding)."""
    if len(inputs) != len(st.rds):
        raise ValueError("Different number of values and ranges")
    if st.has_types:
        sums: Dict[int, int] = {}
        for io, ty, v in st.pub_vt:
            sums[ty % N] = (sums.get(ty % N, 0) + (-v if io else v)) % N
        for (v, ty, _), rd in zip(inputs, st.rds):
            sums[ty % N] = (sums.get(ty % N, 0) + (-v if rd.is_output else v)) % N
        if any(sums.values()):
            raise ValueError("amounts of some type do not balance")
    ph1ss, mss = [], []
    for i, ((v, _, _), rd) in enumerate(zip(inputs, st.rds)):
        p, m = make_phase1s(i, rd, v)
        ph1ss.append(p); mss.append(m)
    types = [("typing", i, rd.is_output, rd.is_assumed, v % N, ty % N) for i, ((v, ty, _), rd) in enumerate(zip(inputs, st.rds))]
    ph1s = (types if st.has_types else []) + [p for ps_ in ph1ss for p in ps_]
    acc: Dict[int, List[int]] = {}
    add = lambda b, ms: acc.__setitem__(b, [(x + y) % N for x, y in zip(acc[b], ms)] if b in acc else list(ms))
    for rd, ms in zip(st.rds, mss):                      # baseMss (:363-370)
        if ms is None:
            continue
        if rd.has_bit:
            add(2, [ms[0]]); add(rd.base, ms[1:])
        else:
            add(rd.base, ms)
    return WitnessTRRP([(v % N, ty % N, bl % N) for v, ty, bl in inputs], ph1s, sorted(acc.items()))


# ----------------------------------------------------------------------------- proof object and the two protocol halves
@dataclass
class RangeProof:
    """RP coms (PBP responses opening) (src/RangeProof.hs:90): coms = blCom : rCom : dmCom : mCom : nComs."""
    coms: List[Point]
    responses: List[Tuple[Point, Point]]      # last round first
    wit_nrm: List[int]
    wit_lin: List[int]


@dataclass
class SetupBP:
    """SBP basis initCom publicCs rounds (src/Bulletproof.hs:327-328), flattened to what verifyBPM consumes."""
    q: int
    cs: List[int]
    pub: RPW
    init_terms: List[Tuple[int, Point]]
    rounds: int


def _init_terms(st: SetupTRRP, coms: Sequence[Point], x: int, q0: int, t: int) -> List[Tuple[int, Point]]:
    """openWith of TranscriptTRRP (TypedReciprocal.hs:293-297)"""
    bl, r, dm, m = coms[:4]
    tp = lambda k: pow(t, k, N)
    ss = [2 * tp(5) * c % N for c in input_coeffs(st.has_types, [rd.is_assumed for rd in st.rds], x, q0)]
    return list(zip(ss, coms[4:])) + [(1, bl), (t % N, m), (tp(2), dm), (tp(3), r)]


def prove_rp(st: SetupTRRP, w: WitnessTRRP, tr: Transcript) -> Tuple[List[Point], SetupBP, RPW]:
    """proveTRRPM (TypedReciprocal.hs:399-446)"""
    num_terms = 3
    is_as = [rd.is_assumed for rd in st.rds]
    m_bases = [b for b, _ in w.base_mss]
    if m_bases != st.m_bases:
        raise ValueError("witness does not cover the setup's shared bases")
    ms_shared = [m for _, ms in w.base_mss for m in ms]
    ds, ms_inline = get_ds_ms(w.ph1s)

    n_wits = [RPW(v, [ty, bl], []) for v, ty, bl in w.inputs]                    # scalarPairRPW' (Internal.hs:59-60)
    n_coms = st.backend.commit_rows([[nw.sc] + nw.lin for nw in n_wits], [st.g] + st.hs[:2])   # = [st.com(nw) ...], one launch
    dm_wit = blind_witness(num_terms, 2, ms_shared, ds, tr.random); dm_com = st.com(dm_wit)
    m_wit = blind_witness(num_terms, 1, [], ms_inline, tr.random); m_com = st.com(m_wit)

    e, x, r0 = tr.oracle([dm_com, m_com] + n_coms, 3)
    e_inv, r0_inv = batch_inverse([e, r0])

    base_map = st.base_map(x)
    ph2s = make_phase2s(e, e_inv, x, base_map, w.ph1s, private=True)
    err7 = r0_inv * (-sum(2 * p.r * p.c for p in ph2s)) % N                      # err7Term (:209-211)
    r_wit = blind_err_witness(num_terms, [err7], [], [p.r for p in ph2s], tr.random); r_com = st.com(r_wit)

    q, xp, r1 = tr.oracle([r_com], 3)
    q0 = st.q_powers(q, 1)[0]
    q_inv, q0_inv, r1_inv = batch_inverse([q, q0, r1])
    shared_cs = make_shared_coeffs(e, e_inv, m_bases, base_map)
    tC = xp if st.has_types else 0

    bls_lin = [tr.random() for _ in range(st.lin_len - 5)]
    bls_nrm = [tr.random() for _ in range(st.nrm_len)]
    bls_ms = bls_lin[1:]

    n_wit_sum = RPW()
    for c, nw in zip(input_coeffs(st.has_types, is_as, x, q0), n_wits):
        n_wit_sum = n_wit_sum + nw.scale(c)
    input_bl = n_wit_sum.lin[1]
    ph3s = list(zip(ph2s, st.q_powers(q, len(ph2s)), bls_nrm))                  # the q^-2i of Ph3 are not used by makeErrorTerms
    errs = make_error_terms(e, xp, shared_cs, bls_ms, ph3s)
    bl_wit = blind_blinding_term(RPW(0, bls_lin, bls_nrm), tC, r0, r0_inv, r1, r1_inv, errs, [m_wit, dm_wit, r_wit], input_bl)
    bl_com = st.com(bl_wit)
    t = tr.oracle([bl_com], 1)[0]

    pub = make_public_consts(e, e_inv, x, xp, q0, q0_inv, t, st.has_types, st.rds, st.pub_vt, ph2s)
    tp = lambda k: pow(t, k, N)
    wit = pub + bl_wit + m_wit.scale(t) + dm_wit.scale(tp(2)) + r_wit.scale(tp(3)) + n_wit_sum.scale(2 * tp(5) % N)
    coms = [bl_com, r_com, dm_com, m_com] + n_coms
    cs = make_bp_coeffs(st.has_types, xp, r0, r1, t, shared_cs)
    return coms, SetupBP(q, cs, pub, _init_terms(st, coms, x, q0, t), st.rounds), wit


def verify_rp(st: SetupTRRP, coms: Sequence[Point], tr: Transcript) -> SetupBP:
    """verifyTRRPM (TypedReciprocal.hs:449-467)"""
    if len(coms) != 4 + len(st.rds):
        raise ValueError("wrong number of range-proof commitments")
    ph1ss = [make_phase1s(i, rd, None)[0] for i, rd in enumerate(st.rds)]
    types = [("typing", i, rd.is_output, rd.is_assumed, None, None) for i, rd in enumerate(st.rds)]
    ph1s = (types if st.has_types else []) + [p for ps_ in ph1ss for p in ps_]
    bl_com, r_com, dm_com, m_com = coms[:4]
    e, x, r0 = tr.oracle([dm_com, m_com] + list(coms[4:]), 3)
    q, xp, r1 = tr.oracle([r_com], 3)
    q0 = st.q_powers(q, 1)[0]
    t = tr.oracle([bl_com], 1)[0]
    e_inv, _, q0_inv = batch_inverse([e, q, q0])
    base_map = st.base_map(x)
    ph2s = make_phase2s(e, e_inv, x, base_map, ph1s, private=False)
    pub = make_public_consts(e, e_inv, x, xp, q0, q0_inv, t, st.has_types, st.rds, st.pub_vt, ph2s)
    cs = make_bp_coeffs(st.has_types, xp, r0, r1, t, make_shared_coeffs(e, e_inv, st.m_bases, base_map))
    return SetupBP(q, cs, pub, _init_terms(st, coms, x, q0, t), st.rounds)


# ----------------------------------------------------------------------------- ZKP instance of RangeProof (src/RangeProof.hs:93-105)
def prove(st: SetupTRRP, w: WitnessTRRP, oracle: OracleN, rand: RandFn) -> RangeProof:
    """proveM: proveRP, then the norm-linear argument on the combined witness (proveBPM, src/Bulletproof.hs:357-359)."""
    tr = Transcript(oracle, rand)
    coms, sbp, wit = prove_rp(st, w, tr)
    resps, nw, lw = st.backend.prove_bp(st.flavour, sbp.rounds, wit.sc, st.g, sbp.q, sbp.cs, wit.nrm, st.gs, wit.lin, st.hs, lambda xs: tr.oracle(xs, 1)[0])
    return RangeProof(coms, resps, nw, lw)


def verify_inputs(st: SetupTRRP, proof: RangeProof, oracle: OracleN) -> Optional[dict]:
    """verifyRP plus the challenge derivation of verifyBPM (src/Bulletproof.hs:374): everything the (batch) verifier's MSM needs.
    None if the proof is malformed (wrong lengths)."""
    if len(proof.responses) != st.rounds or (len(proof.wit_nrm), len(proof.wit_lin)) != st.final_lens or len(proof.coms) != 4 + len(st.rds):
        return None
    tr = Transcript(oracle)
    sbp = verify_rp(st, proof.coms, tr)
    es: List[int] = []
    for a, b in reversed(proof.responses):            # foldrM: first round first, consed => last round first
        es.insert(0, tr.oracle([a, b], 1)[0])
    pad = lambda xs, n: list(xs) + [0] * (n - len(xs))
    return {"q": sbp.q, "sp": sbp.pub.sc, "pub_norm": pad(sbp.pub.nrm, st.nrm_len), "pub_lin_c": pad(sbp.cs, st.lin_len), "pub_lin_x": pad(sbp.pub.lin, st.lin_len),
            "es": es, "responses": list(proof.responses), "wit_norm": list(proof.wit_nrm), "wit_lin": list(proof.wit_lin), "init_terms": sbp.init_terms}


def verify(st: SetupTRRP, proof: RangeProof, oracle: OracleN) -> bool:
    """verifyM (src/RangeProof.hs:103-105; src/Bulletproof.hs:343, :370-378)"""
    v = verify_inputs(st, proof, oracle)
    if v is None:
        return False
    return st.backend.verify_bp(st.flavour, v["q"], v["sp"], st.g, v["pub_norm"], st.gs, v["pub_lin_c"], v["pub_lin_x"], st.hs, v["es"], v["responses"], v["wit_norm"],
                                v["wit_lin"], v["init_terms"])


# ----------------------------------------------------------------------------- device-side verifier scalars (csrc/trrp.hip)
def verifier_challenges(st: SetupTRRP, proof: RangeProof, oracle: OracleN) -> Optional[Tuple[List[int], List[int]]]:
    """The oracle calls of verifyTRRPM (TypedReciprocal.hs:459-462) and of verifyBPM (Bulletproof.hs:374), nothing else:
    ((e, x, r0, q, x', r1, t), [e_k ... e_1])  — the host's whole share of a device-derived verification."""
    if len(proof.responses) != st.rounds or (len(proof.wit_nrm), len(proof.wit_lin)) != st.final_lens or len(proof.coms) != 4 + len(st.rds):
        return None
    tr = Transcript(oracle)
    bl_com, r_com, dm_com, m_com = proof.coms[:4]
    e, x, r0 = tr.oracle([dm_com, m_com] + list(proof.coms[4:]), 3)
    q, xp, r1 = tr.oracle([r_com], 3)
    t = tr.oracle([bl_com], 1)[0]
    es: List[int] = []
    for a, b in reversed(proof.responses):
        es.insert(0, tr.oracle([a, b], 1)[0])
    return [e, x, r0, q, xp, r1, t], es


class DeviceVerifierTables:
    """The static structure of one setup uploaded for bppp_trrp_public_device: from then on a proof's public scalars
    (makePublicConsts, makeBpCoeffs, the initCom scalars) are computed on the GPU from its seven challenges."""

    def __init__(self, gpu, st: SetupTRRP):
        import ctypes as C
        import numpy as np
        from .capi import _ptr, scalars_to_array
        self.gpu, self.st, self.h = gpu, st, None
        ph1ss = [make_phase1s(i, rd, None)[0] for i, rd in enumerate(st.rds)]
        types = [("typing", i, rd.is_output, rd.is_assumed, None, None) for i, rd in enumerate(st.rds)]
        ph1s = (types if st.has_types else []) + [p for ps_ in ph1ss for p in ps_]
        if len(ph1s) != st.nrm_len:
            raise ValueError("phase-1 layout does not match the setup's norm length")
        slot_of = {b: k for k, b in enumerate(st.sorted_bases)}
        syms: List[int] = []
        sym_idx: Dict[int, int] = {}

        def sym(v: int) -> int:
            v %= N
            if v not in sym_idx:
                sym_idx[v] = len(syms)
                syms.append(v)
            return sym_idx[v]
        kind, rng_, slot, psym, coeff = [], [], [], [], []
        for p in ph1s:
            if p[0] == "typing":
                kind.append(0 | (0x100 if p[2] else 0) | (0x200 if p[3] else 0)); rng_.append(p[1]); slot.append(0); psym.append(0xFFFFFFFF); coeff.append(0)
            elif p[0] == "inline":
                kind.append(1); rng_.append(p[1]); slot.append(slot_of[p[2]]); psym.append(sym(p[6]) if p[6] else 0xFFFFFFFF); coeff.append(p[3])
            else:
                kind.append(2); rng_.append(p[1]); slot.append(slot_of[p[2]]); psym.append(0xFFFFFFFF); coeff.append(p[3])
        cs_slot, cs_sym = [], []
        for b in st.m_bases:
            for s_ in range(1, b):
                cs_slot.append(slot_of[b]); cs_sym.append(sym(s_))
        if 6 + len(cs_slot) != st.lin_len:
            raise ValueError("shared-base layout does not match the setup's linear length")
        pub_out = [1 if io else 0 for io, _, _ in st.pub_vt]
        pub_sym = [sym(ty) for _, ty, _ in st.pub_vt]
        pub_amt = [v % N for _, _, v in st.pub_vt]
        u32 = lambda xs: np.ascontiguousarray(np.array(list(xs) or [0], dtype=np.uint32))
        sc = lambda xs: scalars_to_array(list(xs) or [0])
        self._keep = [u32(kind), u32(rng_), u32(slot), u32(psym), sc(coeff), sc([rd.lo % N for rd in st.rds]), u32([1 if rd.is_assumed else 0 for rd in st.rds]),
                      sc(syms), u32(cs_slot), u32(cs_sym), u32(pub_out), sc(pub_amt), u32(pub_sym)]
        k = self._keep
        h = C.c_void_p()
        rc = gpu.lib.bppp_trrp_create(gpu.h, 0 if st.flavour == "NL" else 1, int(st.has_types), st.nrm_len, st.lin_len, len(st.rds), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]),
                                      _ptr(k[3]), _ptr(k[4]), _ptr(k[5]), _ptr(k[6]), len(syms), _ptr(k[7]), _ptr(k[8]), _ptr(k[9]), len(st.pub_vt), _ptr(k[10]),
                                      _ptr(k[11]), _ptr(k[12]), C.byref(h))
        gpu._check(rc, "bppp_trrp_create")
        self.h = h
        gpu._adopt(self)
        self.ninit = 4 + len(st.rds)

    def close(self):
        if self.h:
            self.gpu.lib.bppp_trrp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def public_device(self, batch: int, d_challenges: int, d_q: int, d_sp: int, d_pub_norm: int, d_pub_lin_c: int, d_init_scalars: int):
        """device pointers in, device arrays out (see include/bppp.h); asynchronous on the context's stream"""
        from .capi import _ptr
        self.gpu._check(self.gpu.lib.bppp_trrp_public_device(self.h, batch, _ptr(d_challenges), _ptr(d_q), _ptr(d_sp), _ptr(d_pub_norm), _ptr(d_pub_lin_c),
                                                             _ptr(d_init_scalars)), "bppp_trrp_public_device")

    def public(self, challenges: Sequence[Sequence[int]]) -> List[dict]:
        """host convenience (tests): the derived scalars of each proof as Python integers"""
        import numpy as np
        from .capi import array_to_scalars, scalars_to_array
        B, st, gpu = len(challenges), self.st, self.gpu
        d_ch = gpu.to_device(np.concatenate([scalars_to_array([c % N for c in ch]) for ch in challenges]))
        sizes = {"q": B, "sp": B, "pn": B * st.nrm_len, "cs": B * st.lin_len, "init": B * self.ninit}
        bufs = {k: gpu.to_device(np.zeros((n, 4), dtype=np.uint64)) for k, n in sizes.items()}
        try:
            self.public_device(B, d_ch, bufs["q"], bufs["sp"], bufs["pn"], bufs["cs"], bufs["init"])
            host = {k: array_to_scalars(gpu.download(bufs[k], (n, 4))) for k, n in sizes.items()}
        finally:
            gpu.free(d_ch)
            for p in bufs.values():
                gpu.free(p)
        cut = lambda xs, n, b: xs[b * n:(b + 1) * n]
        return [{"q": host["q"][b], "sp": host["sp"][b], "pub_norm": cut(host["pn"], st.nrm_len, b), "pub_lin_c": cut(host["cs"], st.lin_len, b),
                 "init_scalars": cut(host["init"], self.ninit, b)} for b in range(B)]


# ----------------------------------------------------------------------------- the native range-proof layer (csrc/rp.hip)
class NativeRangeProofs:
    """One setup registered with the library (bppp_rp_create): ranges, layout and basis live on the device; batches of ENCODED
    proofs (the reference's commitments / proof files, bulletproofspp_amd.encoding) are verified end to end there — decoding,
    all SHA-256 transcript hashing (the CLI's shaOracle), verifyTRRPM's scalars, challenge expansion and one combined MSM.
    Both argument flavours verify and prove (bppp_rp_prove_batch: the norm-linear argument through csrc/nlb.hip, the inner-product one
    through csrc/ipb.hip once the handle has its comb table, csrc/rpprove.hip's host-algebra ip_argument_lockstep before that)."""

    def __init__(self, gpu, st: SetupTRRP, oracle_tag: bytes = b"", h: Point = None):
        import ctypes as C
        from .capi import RP_ASSUMED, RP_OUTPUT, RP_SHARED, RpPublic, RpRange, RpShape, int_to_limbs, points_to_array
        self.gpu, self.st, self.h = gpu, st, None
        rng = (RpRange * len(st.rds))()
        for r, rd in zip(rng, st.rds):
            r.base = rd.base
            r.flags = (RP_SHARED if rd.is_shared else 0) | (RP_OUTPUT if rd.is_output else 0) | (RP_ASSUMED if rd.is_assumed else 0)
            r.min[:] = [int(v) for v in int_to_limbs(rd.lo % 2**256)]     # two's complement: a minimum may be negative (examples/rec_test)
            r.max[:] = [int(v) for v in int_to_limbs(rd.hi % 2**256)]
        pubs = (RpPublic * max(len(st.pub_vt), 1))()
        for p_, (io, ty, v) in zip(pubs, st.pub_vt):
            p_.is_output = 1 if io else 0
            p_.type[:] = [int(x) for x in int_to_limbs(ty % N)]
            p_.amount[:] = [int(x) for x in int_to_limbs(v % N)]
        pts = points_to_array([h if h is not None else st.g, st.g] + list(st.hs) + list(st.gs))
        hnd = C.c_void_p()
        rc = gpu.lib.bppp_rp_create(gpu.h, 0 if st.flavour == "NL" else 1, int(st.has_types), C.cast(rng, C.c_void_p), len(st.rds), C.cast(pubs, C.c_void_p), len(st.pub_vt),
                                    C.c_void_p(pts.ctypes.data), pts.shape[0], oracle_tag if oracle_tag else None, C.byref(hnd))
        gpu._check(rc, "bppp_rp_create")
        self.h = hnd
        gpu._adopt(self)
        shp = RpShape()
        gpu._check(gpu.lib.bppp_rp_info(self.h, C.byref(shp)), "bppp_rp_info")
        self.shape = {n: int(getattr(shp, n)) for n, _ in RpShape._fields_}
        if (self.shape["norm_len"], self.shape["lin_len"], self.shape["rounds"], (self.shape["final_norm"], self.shape["final_lin"])) != \
                (st.nrm_len, st.lin_len, st.rounds, tuple(st.final_lens)):
            raise RuntimeError("native setup disagrees with the host setup: %r" % (self.shape,))

    def close(self):
        if self.h:
            self.gpu.lib.bppp_rp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        """bppp_rp_set_option: comb_min, comb_budget, comb_bits, split_min, host_oracle_max, fold_points, host_algebra, timing"""
        from .capi import RP_OPTIONS
        self.gpu._check(self.gpu.lib.bppp_rp_set_option(self.h, RP_OPTIONS[name], int(value)), "bppp_rp_set_option")

    def prove_batch(self, inputs: Sequence[Sequence[Tuple[int, int, int]]], rand_prefixes: Sequence[bytes]) -> List[Tuple[bytes, bytes]]:
        """bppp_rp_prove_batch: inputs[b] = [(amount, type, blinding) per range]; rand_prefixes[b] = the hashToScalar prefix of
        proof b (all of one length).  Returns [(commitments file, proof file)] — the bytes encoding.encode_proof(prove(...)) gives."""
        import ctypes as C
        import numpy as np
        from .capi import scalars_to_array
        B, nr = len(inputs), len(self.st.rds)
        if B == 0:
            return []
        if len(rand_prefixes) != B or len({len(p_) for p_ in rand_prefixes}) != 1 or any(len(row) != nr for row in inputs):
            raise ValueError("one equal-length randomness prefix per proof and one (amount, type, blinding) per range are required")
        amt = scalars_to_array([v % 2**256 for row in inputs for v, _, _ in row])
        typ = scalars_to_array([t % N for row in inputs for _, t, _ in row])
        bld = scalars_to_array([b_ % N for row in inputs for _, _, b_ in row])
        plen = len(rand_prefixes[0])
        pre = np.frombuffer(b"".join(rand_prefixes) or b"\0", dtype=np.uint8)
        cf = np.zeros(B * self.shape["coms_bytes"], dtype=np.uint8)
        pf = np.zeros(B * self.shape["proof_bytes"], dtype=np.uint8)
        vp = lambda a: C.c_void_p(a.ctypes.data)
        rc = self.gpu.lib.bppp_rp_prove_batch(self.h, B, vp(amt), vp(typ), vp(bld), vp(pre), plen, vp(cf), vp(pf))
        self.gpu._check(rc, "bppp_rp_prove_batch")
        cb, pb = self.shape["coms_bytes"], self.shape["proof_bytes"]
        return [(cf[b * cb:(b + 1) * cb].tobytes(), pf[b * pb:(b + 1) * pb].tobytes()) for b in range(B)]

    def verify_batch(self, coms_files: Sequence[bytes], proof_files: Sequence[bytes], seed: Optional[bytes] = None, want_status: bool = False,
                     want_challenges: bool = False):
        """bppp_rp_verify_batch on host byte strings: returns accept, or (accept, status list, challenges per proof) as asked.
        `seed` is the verifier's randomness behind the batch weights: fresh from os.urandom unless given (fixed seeds are for tests)."""
        import ctypes as C
        import numpy as np
        if seed is None:
            seed = os.urandom(32)
        B = len(proof_files)
        if len(coms_files) != B or len(seed) != 32:
            raise ValueError("one commitments file per proof and a 32-byte seed are required")
        if any(len(c) != self.shape["coms_bytes"] for c in coms_files) or any(len(p_) != self.shape["proof_bytes"] for p_ in proof_files):
            return (False, [2] * B, None) if (want_status or want_challenges) else False       # wrong length: malformed, as decodeProof' returns Nothing
        cb, pb = np.frombuffer(b"".join(coms_files), dtype=np.uint8), np.frombuffer(b"".join(proof_files), dtype=np.uint8)
        return self._verify(self.gpu.lib.bppp_rp_verify_batch, B, C.c_void_p(cb.ctypes.data), C.c_void_p(pb.ctypes.data), seed, want_status, want_challenges, (cb, pb))

    def verify_batch_device(self, batch: int, d_coms: int, d_proofs: int, seed: Optional[bytes] = None, want_status: bool = False, want_challenges: bool = False):
        import ctypes as C
        if seed is None:
            seed = os.urandom(32)
        return self._verify(self.gpu.lib.bppp_rp_verify_batch_device, batch, C.c_void_p(d_coms), C.c_void_p(d_proofs), seed, want_status, want_challenges, None)

    def verify_batch_device_point(self, batch: int, d_coms: int, d_proofs: int, seed: bytes, index_offset: int = 0) -> Tuple[bool, Point]:
        """bppp_rp_verify_shard_device: (accept, the combined point) — the partial result of one rank when the job is sharded
        proof-per-GPU; this rank holds proofs [index_offset, index_offset + batch) of the job, every rank passes the same seed."""
        import ctypes as C
        import numpy as np
        from .capi import array_to_point
        acc, out = C.c_int(0), np.zeros(8, dtype=np.uint64)
        sd = np.frombuffer(seed, dtype=np.uint8)
        rc = self.gpu.lib.bppp_rp_verify_shard_device(self.h, batch, index_offset, C.c_void_p(d_coms), C.c_void_p(d_proofs), C.c_void_p(sd.ctypes.data),
                                                      C.byref(acc), None, None, C.c_void_p(out.ctypes.data))
        self.gpu._check(rc, "bppp_rp_verify_shard_device")
        return bool(acc.value), array_to_point(out)

    def _verify(self, fn, B, pc, pp, seed, want_status, want_challenges, keep):
        import ctypes as C
        import numpy as np
        from .capi import array_to_scalars
        acc = C.c_int(0)
        status = np.zeros(max(B, 1), dtype=np.uint32) if want_status else None
        nch = self.shape["challenges_per_proof"]
        chal = np.zeros((max(B, 1) * nch, 4), dtype=np.uint64) if want_challenges else None
        sd = np.frombuffer(seed, dtype=np.uint8)
        rc = fn(self.h, B, pc, pp, C.c_void_p(sd.ctypes.data), C.byref(acc), C.c_void_p(status.ctypes.data) if want_status else None,
                C.c_void_p(chal.ctypes.data) if want_challenges else None, None)
        self.gpu._check(rc, "bppp_rp_verify_batch")
        if not (want_status or want_challenges):
            return bool(acc.value)
        chs = None
        if want_challenges:
            flat = array_to_scalars(chal)
            lead = nch - self.shape["rounds"]               # 7 range-proof challenges (typed reciprocal) or 4 (binary), then one per round
            chs = [(flat[b * nch:b * nch + lead], flat[b * nch + lead:(b + 1) * nch]) for b in range(B)]
        return bool(acc.value), ([int(v) for v in status[:B]] if want_status else None), chs


# ----------------------------------------------------------------------------- schema files (app/Parse.hs, app/Main.hs)
def approx_log_w(n: int) -> int:
    """approxLogW (app/Parse.hs:202-206): the default base for a range of width n"""
    l = integer_log(2, n)
    return l // integer_log(2, l)


def setup_from_schema(backend: Backend, schema: dict, points: Optional[Sequence[Point]] = None) -> SetupTRRP:
    """The reciprocal branch of the CLI's schema handling (app/Parse.hs:100-186, app/Main.hs:262-285): defaults argument = IP,
    count = 1, min = 0, max = 2^64, base = approxLogW (max - min), flags False; typed or conserved => hasTypes.  `points`
    defaults to the try-and-increment stream over schema["basisSeed"] (getPoints, app/Main.hs:68-72; even-y root: this build's
    documented choice).  Binary schemas ("binary": true) are a different protocol (RangeProof.Binary): they go through
    bulletproofspp_amd.rangeproof_binary.setup_from_schema and are refused here."""
    if schema.get("binary", False):
        raise ValueError("a binary schema (RangeProof.Binary): use bulletproofspp_amd.rangeproof_binary.setup_from_schema")
    arg = str(schema.get("argument", "IP")).lower()
    flavour = {"ip": "IP", "innerproduct": "IP", "nl": "NL", "normlinear": "NL"}.get(arg)
    if flavour is None:
        raise ValueError("Unsupported Argument: " + arg)
    has_types = bool(schema.get("typed", False)) or bool(schema.get("conserved", False))
    rds: List[RangeData] = []
    for r in schema["ranges"]:
        lo, hi = int(r.get("min", 0)), int(r.get("max", 2**64))
        base = int(r["base"]) if "base" in r else approx_log_w(hi - lo)
        rd = make_range_data(base, lo, hi, bool(r.get("isShared", False)), bool(r.get("isOutput", False)), bool(r.get("isAssumed", False)))
        if rd is None:
            raise ValueError("Invalid range: %r" % (r,))
        rds += [rd] * int(r.get("count", 1))
    pubs = []
    for pb in schema.get("public", []):
        if pb.get("blind") is not None:
            raise ValueError("Cannot have blinding on public value")
        pubs.append((bool(pb.get("isOutput", False)), int(pb.get("type", 0)), int(pb["amount"])))
    if points is None:
        if "basisSeed" not in schema:
            raise ValueError("no basis: pass points or give basisSeed")
        nrm_len = sum(len(rd.base_coeffs) + (1 if has_types else 0) for rd in rds)
        points = basis_points(str(schema["basisSeed"]).encode(), 2 + nrm_len + 6 + sum(rd.base for rd in rds if rd.is_shared) + 2)
    return setup(backend, points, has_types, pubs, rds, flavour)


def inputs_from_witness(witness_json: Sequence[dict], random_seed: bytes = b"default random seed") -> List[Tuple[int, int, int]]:
    """(amount, type, blinding) per entry; missing blindings come from hashToScalars ("Blinding " <> seed) numbered from 1
    (app/Main.hs:86-87, :252-253)"""
    gen = hash_to_scalar(b"Blinding " + random_seed)
    return [(int(w["amount"]), int(w.get("type", 0)), int(w["blind"]) if w.get("blind") is not None else gen(i + 1)) for i, w in enumerate(witness_json)]


def basis_points(seed: bytes, count: int) -> List[Point]:
    """getPoints (app/Main.hs:68-72): x = hash (seed <> show n) for n = 0, 1, ... — the digest decoded by Binary (Prime p)
    (decode_field, Encoding.hs:75-79) — kept when x^3 + 7 is a square (pointX); the root taken is the even one (the
    reference's `sr` choice cannot be confirmed offline)."""
    p = 2**256 - 2**32 - 977
    out, n = [], 0
    while len(out) < count:
        x = decode_field(hashlib.sha256(seed + str(n).encode()).digest(), p)
        n += 1
        rhs = (x * x * x + 7) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p != rhs:
            continue
        out.append((x, p - y if y & 1 else y))
    return out
