"""
pyoracle.py — pure-Python big-int restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (bulletproofspp_amd) never does.

PARITY STATUS: "parity unpinned" by the reference's own tests (it has none: SURVEY.md §4, §8c).
Pinned instead by the reference's constants, by OpenSSL libcrypto, and by agreement between this
module and the independent C restatement oracle/bppp_oracle.c (see that file's header).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Containers are the list instance of BPCollection (src/Bulletproof.hs:68-99), which is what the
shipped CLI uses (`type ArgColl = []`, app/Main.hs:101).

Conventions: scalars are ints in [0, N); coordinates ints in [0, P); an affine point is a tuple
(x, y) and the point at infinity (zeroV) is None.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

# secp256k1 (src/Data/Curve/Weierstrass/FastSECP256K1.hs:33,47,103-110,134-141)
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
B7 = 7
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE  # FastSECP256K1.hs:39
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72  # FastSECP256K1.hs:53

Point = Optional[Tuple[int, int]]


def inv_mod(a: int, m: int) -> int:
    """recip; 0 ↦ 0 as in batchInverse (src/Data/Field/BatchInverse.hs:18,23)."""
    a %= m
    return pow(a, m - 2, m) if a else 0


def batch_inverse(xs: Sequence[int], m: int) -> List[int]:
    """batchInverse (src/Data/Field/BatchInverse.hs:14-24): Montgomery trick, 0 ↦ 0."""
    pre, acc = [], 1
    for x in xs:
        pre.append(acc)
        if x % m:
            acc = acc * x % m
    y = inv_mod(acc, m)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        if xs[i] % m == 0:
            continue
        out[i] = y * pre[i] % m
        y = y * xs[i] % m
    return out


# ----------------------------------------------------------------------------- curve (pure python)
class PyEC:
    """Group law on y^2 = x^3 + 7 from the curve equation (the reference delegates to the
    un-vendored elliptic-curve-0.3.0; src/Commitment.hs:94-113)."""

    @staticmethod
    def on_curve(p: Point) -> bool:
        if p is None:
            return True
        x, y = p
        return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B7) % P == 0

    @staticmethod
    def neg(p: Point) -> Point:
        return None if p is None else (p[0], (-p[1]) % P)

    @staticmethod
    def add(p: Point, q: Point) -> Point:
        if p is None:
            return q
        if q is None:
            return p
        x1, y1 = p
        x2, y2 = q
        if x1 == x2:
            if (y1 + y2) % P == 0:
                return None
            lam = 3 * x1 * x1 * inv_mod(2 * y1, P) % P
        else:
            lam = (y2 - y1) * inv_mod(x2 - x1, P) % P
        x3 = (lam * lam - x1 - x2) % P
        return (x3, (lam * (x1 - x3) - y1) % P)

    # Jacobian helpers for the Straus loops (formula-independent result)
    @staticmethod
    def _jdbl(X, Y, Z):
        if Z == 0 or Y == 0:
            return (1, 1, 0)
        A = X * X % P
        Bq = Y * Y % P
        C = Bq * Bq % P
        D = 2 * ((X + Bq) ** 2 - A - C) % P
        E = 3 * A % P
        X3 = (E * E - 2 * D) % P
        return (X3, (E * (D - X3) - 8 * C) % P, 2 * Y * Z % P)

    @staticmethod
    def _jmadd(a: Point, X1, Y1, Z1):
        """nrmlAdd (src/Commitment.hs:128-144), completed for h = 0."""
        if a is None:
            return (X1, Y1, Z1)
        x2, y2 = a
        if Z1 == 0:
            return (x2, y2, 1)
        z1z1 = Z1 * Z1 % P
        u2 = x2 * z1z1 % P
        s2 = y2 * Z1 * z1z1 % P
        h = (u2 - X1) % P
        r0 = (s2 - Y1) % P
        if h == 0:
            return PyEC._jdbl(X1, Y1, Z1) if r0 == 0 else (1, 1, 0)
        hh = h * h % P
        i = 4 * hh % P
        j = h * i % P
        r = 2 * r0 % P
        v = X1 * i % P
        t = Y1 * j % P
        x3 = (r * r - j - 2 * v) % P
        return (x3, (r * (v - x3) - 2 * t) % P, ((Z1 + h) ** 2 - z1z1 - hh) % P)

    @staticmethod
    def _jaff(X, Y, Z) -> Point:
        if Z == 0:
            return None
        zi = inv_mod(Z, P)
        return (X * zi * zi % P, Y * zi * zi * zi % P)

    def straus(self, terms: Sequence[Tuple[int, Point]], rows: int) -> Point:
        """go len zeroV (src/Commitment.hs:334-335 / :352-353); terms are (non-negative magnitude, point)."""
        v = (1, 1, 0)
        for row in range(rows, 0, -1):
            v = self._jdbl(*v)
            for s, b in terms:
                if (s >> (row - 1)) & 1:
                    v = self._jmadd(b, *v)
        return self._jaff(*v)

    def inner_product(self, sgs: Sequence[Tuple[int, Point]]) -> Point:
        """innerProduct (src/Commitment.hs:325-335) + normalizeBasis (:364-367)."""
        terms = []
        for s, g in sgs:
            r = reduce_scalar(s)
            terms.append((-r, self.neg(g)) if r < 0 else (r, g))
        return self.straus(terms, 256)

    def pair_ip(self, s0: int, g0: Point, s1: int, g1: Point) -> Point:
        """projectivePairIP (src/Commitment.hs:343-353); s0, s1 are signed reduced scalars."""
        terms = [(-s, self.neg(g)) if s < 0 else (s, g) for s, g in ((s0, g0), (s1, g1))]
        return self.straus(terms, 129)

    def mul(self, s: int, p: Point) -> Point:
        return self.inner_product([(s % N, p)])


# ----------------------------------------------------------------------------- curve (C oracle via ctypes)
def _to_limbs(x: int, n: int = 4):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def _from_limbs(ls) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(ls))


def pt_to_limbs(p: Point):
    return [0] * 8 if p is None else _to_limbs(p[0]) + _to_limbs(p[1])


def pt_from_limbs(ls) -> Point:
    x, y = _from_limbs(ls[:4]), _from_limbs(ls[4:8])
    return None if x == 0 and y == 0 else (x, y)


def oracle_lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libbppp_oracle.so")


class CEC:
    """Same interface as PyEC, backed by oracle/bppp_oracle.c (fast enough for 2^16-term MSMs)."""

    def __init__(self, path: Optional[str] = None):
        self.lib = ctypes.CDLL(path or oracle_lib_path())
        self.U64 = ctypes.c_uint64

    def _arr(self, vals):
        return (self.U64 * len(vals))(*vals)

    def on_curve(self, p: Point) -> bool:
        return bool(self.lib.orc_on_curve(self._arr(pt_to_limbs(p))))

    def neg(self, p: Point) -> Point:
        return PyEC.neg(p)

    def add(self, p: Point, q: Point) -> Point:
        out = (self.U64 * 8)()
        self.lib.orc_point_add(self._arr(pt_to_limbs(p)), self._arr(pt_to_limbs(q)), out)
        return pt_from_limbs(out)

    def mul(self, s: int, p: Point) -> Point:
        out = (self.U64 * 8)()
        self.lib.orc_point_mul(self._arr(_to_limbs(s % N)), self._arr(pt_to_limbs(p)), out)
        return pt_from_limbs(out)

    def inner_product(self, sgs) -> Point:
        n = len(sgs)
        sc = (self.U64 * (4 * max(n, 1)))()
        pt = (self.U64 * (8 * max(n, 1)))()
        for i, (s, g) in enumerate(sgs):
            sc[4 * i:4 * i + 4] = _to_limbs(s % N)
            pt[8 * i:8 * i + 8] = pt_to_limbs(g)
        out = (self.U64 * 8)()
        self.lib.orc_inner_product(sc, pt, ctypes.c_size_t(n), out)
        return pt_from_limbs(out)

    def inner_product_raw(self, sc_buf, pt_buf, n: int) -> Point:
        """scalars / points already packed as contiguous uint64 buffers (numpy .ctypes or ctypes arrays)."""
        out = (self.U64 * 8)()
        self.lib.orc_inner_product(sc_buf, pt_buf, ctypes.c_size_t(n), out)
        return pt_from_limbs(out)

    def pair_ip(self, s0: int, g0: Point, s1: int, g1: Point) -> Point:
        out = (self.U64 * 8)()
        self.lib.orc_pair_ip(self._arr(_to_limbs(abs(s0), 3)), int(s0 < 0), self._arr(pt_to_limbs(g0)),
                             self._arr(_to_limbs(abs(s1), 3)), int(s1 < 0), self._arr(pt_to_limbs(g1)), out)
        return pt_from_limbs(out)

    def rational_reduce(self, x: int) -> Tuple[int, int]:
        am, bm = (self.U64 * 3)(), (self.U64 * 3)()
        an, bn = ctypes.c_int(0), ctypes.c_int(0)
        self.lib.orc_rational_reduce(self._arr(_to_limbs(x % N)), am, ctypes.byref(an), bm, ctypes.byref(bn))
        a, b = _from_limbs(am), _from_limbs(bm)
        return (-a if an.value else a, -b if bn.value else b)

    def lift_x(self, x: int) -> Point:
        out = (self.U64 * 8)()
        ok = self.lib.orc_lift_x(self._arr(_to_limbs(x)), out)
        return pt_from_limbs(out) if ok else None


# ----------------------------------------------------------------------------- SplitScalar (Prime p)
def reduce_scalar(x: int) -> int:
    """reduceScalar (src/Commitment.hs:276-279): signed representative in (-n/2, n/2]."""
    x %= N
    return -(N - x) if x > N - x else x


def _quot(a: int, b: int) -> int:
    """Haskell `quot`: truncation toward zero."""
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def rational_reduce_scalar(x: int) -> Tuple[int, int]:
    """rationalReduceScalar (src/Commitment.hs:242-255): the egcd list starts at its second
    argument (:252); first (r, s) with r^2 <= 2n (:247).  Invariant r ≡ s·x (mod n)."""
    prev = (N, 0)
    cur = (reduce_scalar(x), 1)
    while cur[0] * cur[0] > 2 * N:
        q = _quot(prev[0], cur[0])
        prev, cur = cur, (prev[0] - q * cur[0], prev[1] - q * cur[1])
    return cur


# ----------------------------------------------------------------------------- Utils folds
def powers(a: int, count: int, start: int = 1) -> List[int]:
    """powers / powers'' (src/Utils.hs:104-111), truncated to `count` terms."""
    out, v = [], start % N
    for _ in range(count):
        out.append(v)
        v = v * a % N
    return out


def powers1(a: int, count: int) -> List[int]:
    """powers' = tail . powers (src/Utils.hs:107-108)."""
    return powers(a, count, a)


def dot_zip(xs, ys) -> int:
    """dotZip (src/Utils.hs:209-210): zip truncates to the shorter."""
    return sum(x * y for x, y in zip(xs, ys)) % N


def weighted_dot_zip(ws, xs, ys) -> int:
    """weightedDotZip (src/Utils.hs:212-216)."""
    return sum(w * x * y for w, x, y in zip(ws, xs, ys)) % N


def chunks(n: int, xs: list) -> List[list]:
    """chunks (src/Utils.hs:222-224)."""
    return [xs[i:i + n] for i in range(0, len(xs), n)]


def round_reduce(n: int) -> int:
    """roundReduce (src/Bulletproof.hs:310-311)."""
    return n // 2 + n % 2


def number_rounds_reduce(n: int) -> Tuple[int, int]:
    """numberRoundsReduce (src/Bulletproof.hs:300-303)."""
    r = 0
    while n >= 5:
        n = round_reduce(n)
        r += 1
    return r, n


def round_reduce_by(n: int, k: int) -> int:
    for _ in range(k):
        n = round_reduce(n)
    return n


def optimal_witness_size_nl(n_len: int, l_len: int) -> Tuple[int, Tuple[int, int]]:
    """NormLinear optimalWitnessSize (src/Bulletproof/NormArgument.hs:165-178)."""
    nR, n1 = number_rounds_reduce(n_len)
    lR, l1 = number_rounds_reduce(l_len)
    r = max(nR, lR)
    n2 = round_reduce_by(n1, r - nR)
    l2 = round_reduce_by(l1, r - lR)
    if n2 + l2 > 5:
        return r + 1, (round_reduce(n2), round_reduce(l2))
    return r, (n2, l2)


def tensor(bs: Sequence[int], es: Sequence[int], qs_fn: Callable[[int], int]) -> List[int]:
    """tensor' for lists (src/Bulletproof.hs:94-95): foldr over es consuming qs from the left;
    qs_fn(k) is the k-th element of the (infinite) qs list."""
    ts = [1]
    k = 0
    for e in reversed(list(es)):
        q = qs_fn(k)
        k += 1
        ts = [q * t % N for t in ts] + [e * t % N for t in ts]
    return [b * t % N for b in bs for t in ts]


def tensor_vector(bs: Sequence[int], es: Sequence[int], qs: Sequence[int]) -> List[int]:
    """tensor' of the Data.Vector instance (src/Bulletproof.hs:114-122): V.generate (2^k * |bs|) multIndex with
    multIndex n = bs ! (n div 2^k) * product [if testBit n j then e else q | (j, e, q) <- zip3 [0..] (reverse es) qs]."""
    xs = list(zip(range(len(es)), reversed(list(es)), qs))
    l_exp = 1 << len(xs)
    out = []
    for n in range(l_exp * len(bs)):
        v = bs[n // l_exp]
        for j, e, q in xs:
            v = v * (e if (n >> j) & 1 else q) % N
        out.append(v)
    return out


def contract_vector(xs: Sequence[int], ys: Sequence[int]) -> List[int]:
    """contract' of the Data.Vector instance (src/Bulletproof.hs:147-159): one dot product per chunk of |xs| elements of ys, the
    last chunk possibly short (V.zipWith truncates)."""
    x_len, y_len = len(xs), len(ys)
    n, r = divmod(y_len, x_len)
    out = []
    for i in range(n + (1 if r else 0)):
        chunk = ys[y_len - r:] if i == n else ys[i * x_len:(i + 1) * x_len]
        out.append(sum(a * b for a, b in zip(xs, chunk)) % N)
    return out


def contract(xs: Sequence[int], ys: Sequence[int]) -> List[int]:
    """contract' for lists (src/Bulletproof.hs:97)."""
    return [dot_zip(xs, ch) for ch in chunks(len(xs), list(ys))]


def _halves(xs: list, default):
    """adjacent-pair traversal of foldMapHalves / mapHalves (src/Bulletproof.hs:77-90)."""
    for i in range(0, len(xs), 2):
        yield xs[i], (xs[i + 1] if i + 1 < len(xs) else default)


def zip_with_def2(f, x0, y0, xs, ys):
    """zipWithDef'' (src/Utils.hs:186-189)."""
    n = max(len(xs), len(ys))
    return [f(xs[i] if i < len(xs) else x0, ys[i] if i < len(ys) else y0) for i in range(n)]


def zip_with_def1(f, y0, xs, ys):
    """zipWithDef' (src/Utils.hs:182-184): length of xs."""
    return [f(xs[i], ys[i] if i < len(ys) else y0) for i in range(len(xs))]


# ----------------------------------------------------------------------------- Norm (NL flavour)
@dataclass
class Norm:
    """data Norm = N q qInv (BPF'' nrmlz [NF x g]) (src/Bulletproof/NormArgument.hs:86-99)."""
    q: int
    q_inv: int
    n: int
    body: List[Tuple[int, Point]]

    @staticmethod
    def make(q: int, ss: Sequence[int], gs: Sequence[Point]) -> "Norm":
        """makeNorm (:98-99)."""
        return Norm(q % N, inv_mod(q, N), 1, zip_with_def2(lambda s, g: (s % N, g), 0, None, list(ss), list(gs)))

    def open_terms(self):
        return [(x, g) for x, g in self.body]

    def eval_scalar(self) -> int:
        """evalScalar (:110-111)."""
        ss = [x for x, _ in self.body]
        return self.n * self.n * weighted_dot_zip(powers1(self.q * self.q % N, len(ss)), ss, ss) % N

    def make_scalars_coms(self):
        """makeScalarsComs (:113-118) via foldXR (:20-29)."""
        q, qi, n = self.q, self.q_inv, self.n
        q4 = pow(q, 4, N)
        s, sx, sr = 1, 0, 0
        xw, rw = [], []
        for (xl, gl), (xr, gr) in _halves(self.body, (0, None)):
            sx = (sx + s * xl * xr) % N
            sr = (sr + s * xr * xr) % N
            xw += [(q * xr % N, gl), (qi * xl % N, gr)]
            rw.append((xr, gr))
            s = s * q4 % N
        sX = 2 * n * n * pow(q, 3, N) * sx % N
        sR = n * n * q4 * sr % N
        return sX, Norm(q, qi, n, xw), sR, Norm(q, qi, n, rw)

    def get_witness(self) -> List[int]:
        """getWitness (:121)."""
        return [x * self.n % N for x, _ in self.body]

    def collapse(self, e: int, ec) -> "Norm":
        """collapse (:123-129)."""
        q, qi = self.q, self.q_inv
        a1, b1 = rational_reduce_scalar(e * qi % N)
        b0 = b1 % N
        b0i = inv_mod(b0, N)
        body = []
        for (xl, gl), (xr, gr) in _halves(self.body, (0, None)):
            body.append(((b0i * xl + e * q % N * b0i % N * xr) % N, ec.pair_ip(b1, gl, a1, gr)))
        return Norm(q * q % N, qi * qi % N, self.n * b0 % N * qi % N, body)

    @staticmethod
    def expand_challenges(es, wit: "Norm", pub: "Norm", basis: "Norm"):
        """expandChallenges (:131-145)."""
        vs = [wit.n * x % N for x, _ in wit.body]
        q = pub.q
        qF = q
        for _ in range(len(es)):
            qF = qF * qF % N
        sc = weighted_dot_zip(powers1(qF * qF % N, len(vs)), vs, vs)
        ts = tensor(vs, es, lambda k: pow(q, 2**k, N))
        pairs = list(zip(pub.body, basis.body))
        body = zip_with_def1(lambda pg, ep: ((pg[0][0] - ep) % N, pg[1][1]), 0, pairs, ts)
        return sc, Norm(1, 1, 1, body)


# ----------------------------------------------------------------------------- Linear (NL flavour)
@dataclass
class Linear:
    """newtype Linear = L (BPF'' nrmlz [LF c x g]) (src/Bulletproof/NormArgument.hs:34-48)."""
    n: int
    body: List[Tuple[int, int, Point]]

    @staticmethod
    def make(cs, ss, gs) -> "Linear":
        """makeLinear (:47-48)."""
        cx = zip_with_def2(lambda c, s: (c % N, s % N), 0, 0, list(cs), list(ss))
        return Linear(1, zip_with_def2(lambda c_x, g: (c_x[0], c_x[1], g), (0, 0), None, cx, list(gs)))

    def open_terms(self):
        return [(x, g) for _, x, g in self.body]

    def eval_scalar(self) -> int:
        """evalScalar (:53-54)."""
        return sum(c * x for c, x, _ in self.body) % N

    def make_scalars_coms(self):
        """makeScalarsComs (:56-59)."""
        sx, sr = 0, 0
        xw, rw = [], []
        for (cl, xl, gl), (cr, xr, gr) in _halves(self.body, (0, 0, None)):
            sx = (sx + cl * xr + cr * xl) % N
            sr = (sr + cr * xr) % N
            xw += [(cl, xr, gl), (cr, xl, gr)]
            rw.append((cr, xr, gr))
        return sx, Linear(self.n, xw), sr, Linear(self.n, rw)

    def get_witness(self) -> List[int]:
        """getWitness (:62)."""
        return [self.n * x % N for _, x, _ in self.body]

    def collapse(self, e: int, ec) -> "Linear":
        """collapse (:64-71)."""
        a1, b1 = rational_reduce_scalar(e)
        a0, b0 = a1 % N, b1 % N
        b0i = inv_mod(b0, N)
        body = []
        for (cl, xl, gl), (cr, xr, gr) in _halves(self.body, (0, 0, None)):
            body.append(((b0 * cl + a0 * cr) % N, (b0i * xl + e * b0i % N * xr) % N, ec.pair_ip(b1, gl, a1, gr)))
        return Linear(self.n * b0 % N, body)

    @staticmethod
    def expand_challenges(es, wit: "Linear", pub: "Linear", basis: "Linear"):
        """expandChallenges (:73-81)."""
        exp_es = tensor([1], es, lambda k: 1)
        cs1 = contract(exp_es, [c for c, _, _ in pub.body])
        vs = [wit.n * x % N for _, x, _ in wit.body]
        sc = dot_zip(cs1, vs)
        ts = tensor(vs, es, lambda k: 1)
        pairs = list(zip(pub.body, basis.body))
        body = zip_with_def1(lambda pg, ep: (pg[0][0], (pg[0][1] - ep) % N, pg[1][2]), 0, pairs, ts)
        return sc, Linear(1, body)


# ----------------------------------------------------------------------------- NormLinear = BPCompose Norm Linear
@dataclass
class NormLinear:
    """newtype NormLinear = NL (BPComp s norm linear) (NormArgument.hs:153-162; Bulletproof.hs:225-269)."""
    s: int
    norm: Norm
    lin: Linear

    @staticmethod
    def make(s, q, cs, nss, ngs, lss, lgs) -> "NormLinear":
        """makeNormLinearBP' (NormArgument.hs:162)."""
        return NormLinear(s % N, Norm.make(q, nss, ngs), Linear.make(cs, lss, lgs))

    def open_terms(self):
        """openWith (Bulletproof.hs:231-232): norm elements then linear elements."""
        return self.norm.open_terms() + self.lin.open_terms()

    @staticmethod
    def make_es(e: int) -> Tuple[int, int]:
        """makeEs (NormArgument.hs:109 via Bulletproof.hs:254)."""
        return e % N, (e * e - 1) % N

    def eval_scalar(self) -> int:
        """evalScalar (Bulletproof.hs:256)."""
        return self.s * (self.norm.eval_scalar() + self.lin.eval_scalar()) % N

    def make_scalars_coms(self):
        """makeScalarsComs (Bulletproof.hs:258-261): no scalarComp factor."""
        slA, wlA, srA, wrA = self.norm.make_scalars_coms()
        slB, wlB, srB, wrB = self.lin.make_scalars_coms()
        return (slA + slB) % N, NormLinear(self.s, wlA, wlB), (srA + srB) % N, NormLinear(self.s, wrA, wrB)

    def get_witness(self) -> List[int]:
        """getWitness (Bulletproof.hs:264)."""
        return [self.s * w % N for w in self.norm.get_witness() + self.lin.get_witness()]

    def collapse(self, e: int, ec) -> "NormLinear":
        """collapse (Bulletproof.hs:266)."""
        return NormLinear(self.s, self.norm.collapse(e, ec), self.lin.collapse(e, ec))

    @staticmethod
    def expand_challenges(es, wit: "NormLinear", pub: "NormLinear", basis: "NormLinear"):
        """expandChallenges (Bulletproof.hs:268-269): scalars add, bodies compose with pub's s."""
        sa, na = Norm.expand_challenges(es, wit.norm, pub.norm, basis.norm)
        sb, lb = Linear.expand_challenges(es, wit.lin, pub.lin, basis.lin)
        return (sa + sb) % N, NormLinear(pub.s, na, lb)


# ----------------------------------------------------------------------------- PSV, commit, prover, verifier
@dataclass
class PSV:
    """PedersenScalarVector (src/Commitment.hs:487-501): scalar·g + body."""
    sc: int
    g: Point
    body: NormLinear

    def open_terms(self):
        """openWith (Commitment.hs:499-501) with openToList's (:) fold (:412-413): body first, (s,g) last."""
        return self.body.open_terms() + [(self.sc % N, self.g)]


def commit(terms, ec) -> Point:
    """commit = innerProduct . openToList (src/Commitment.hs:416-417)."""
    return ec.inner_product(terms)


OracleFn = Callable[[List[Point]], int]


class Transcript:
    """The MonadZKP oracle of ZKPT (src/ZKP.hs:96-101): prepends the new commitments to the whole
    transcript and hashes all of it; `fn` maps that list to the first scalar (`head <$> oracle`)."""

    def __init__(self, fn: OracleFn):
        self.fn = fn
        self.cs: List[Point] = []

    def oracle(self, xs: List[Point]) -> int:
        self.cs = list(xs) + self.cs
        return self.fn(self.cs) % N


def prove_round(com: PSV, tr: Transcript, ec):
    """proveRoundM (src/Bulletproof.hs:346-355)."""
    c = com.body
    as_, a, bs_, b = c.make_scalars_coms()
    ac = commit(PSV(as_, com.g, a).open_terms(), ec)
    bc = commit(PSV(bs_, com.g, b).open_terms(), ec)
    e = tr.oracle([ac, bc])
    e0, e1 = c.make_es(e)
    sc1 = (com.sc + e0 * as_ + e1 * bs_) % N
    return PSV(sc1, com.g, c.collapse(e, ec)), (ac, bc), e


def prove_bp(n_rounds: int, com: PSV, tr: Transcript, ec):
    """proveBPM (src/Bulletproof.hs:357-359): responses come out LAST ROUND FIRST."""
    resps, es = [], []
    for _ in range(n_rounds):
        com, r, e = prove_round(com, tr, ec)
        resps.insert(0, r)
        es.insert(0, e)
    return com, resps, es


def verify_challenges(rs, tr: Transcript) -> List[int]:
    """the foldrM of verifyBPM (src/Bulletproof.hs:374): walks rs from the right (first round first)
    and conses, so es is ordered like rs (last round first)."""
    es: List[int] = []
    for a, b in reversed(rs):
        es.insert(0, tr.oracle([a, b]))
    return es


def verify_terms(init_terms, es, rs, pub: PSV, basis: PSV, wit_body: NormLinear):
    """The term list verifyBPM commits (src/Bulletproof.hs:375-377 with verifyWith :362-368):
    wit' ++ initCom ++ (e0·X, e1·R per response)."""
    sc, chs = NormLinear.expand_challenges(es, wit_body, pub.body, basis.body)
    wit1 = PSV((pub.sc - sc) % N, basis.g, chs)
    terms = wit1.open_terms() + list(init_terms)
    for e, (x, r) in zip(es, rs):
        e0, e1 = NormLinear.make_es(e)
        terms += [(e0, x), (e1, r)]
    return terms


def verify_bp(init_terms, rs, pub: PSV, basis: PSV, wit_body: NormLinear, tr: Transcript, ec) -> bool:
    """verifyBPM (src/Bulletproof.hs:370-378): zeroV == commit(...)."""
    es = verify_challenges(rs, tr)
    return commit(verify_terms(init_terms, es, rs, pub, basis, wit_body), ec) is None


# ----------------------------------------------------------------------------- deterministic stand-ins (harness)
def sha_oracle_fn(tag: bytes = b"bppp") -> OracleFn:
    """A deterministic injected oracle (the reference injects one: src/ZKP.hs:73-77).  The build's
    documented choice: SHA-256 over decimal coordinates, first challenge only."""
    import hashlib

    def fn(cs: List[Point]) -> int:
        h = hashlib.sha256()
        h.update(tag + b"1" + str(len(cs)).encode())
        for p in cs:
            if p is None:
                h.update(b"inf")
            else:
                h.update(str(p[0]).encode() + str(p[1]).encode())
        return int.from_bytes(h.digest(), "big") % N

    return fn


def hash_points(seed: bytes, count: int, ec: Optional[CEC] = None) -> List[Point]:
    """getPoints-style try-and-increment basis (app/Main.hs:68-72): x ← SHA-256(seed ‖ show n) mod p,
    accept when x^3+7 is a square; the build's documented root choice is the even y."""
    import hashlib
    out, n = [], 0
    while len(out) < count:
        x = int.from_bytes(hashlib.sha256(seed + str(n).encode()).digest(), "big") % P
        n += 1
        rhs = (x * x * x + B7) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P != rhs:
            continue
        if y & 1:
            y = P - y
        out.append((x, y))
    return out


# ============================================================================= inner-product flavour
# Restatement of src/Bulletproof/InnerProductArgument.hs (the default argument of the CLI, app/Parse.hs:100).
def number_rounds_reduce1(n: int) -> Tuple[int, int]:
    """numberRoundsReduce' (src/Bulletproof.hs:305-307): reduce to less than 3."""
    r, n1 = number_rounds_reduce(n)
    return (r + 1, round_reduce(n1)) if n1 > 2 else (r, n1)


def optimal_witness_size_ip(n_len: int, l_len: int) -> Tuple[int, Tuple[int, int]]:
    """IP-flavour NormLinear optimalWitnessSize (InnerProductArgument.hs:253-267)."""
    n_even = (n_len + (n_len % 2)) // 2
    nR, n1 = number_rounds_reduce1(n_even)
    lR, l1 = number_rounds_reduce(l_len)
    r = max(nR, lR)
    n2 = round_reduce_by(n1, r - nR)
    l2 = round_reduce_by(l1, r - lR)
    if 2 * n2 + l2 > 5:
        return r + 1, (2 * round_reduce(n2), round_reduce(l2))
    return r, (2 * n2, l2)


@dataclass
class InnerProduct:
    """data InnerProduct = IP s nrmlzY q qInv (BPF'' nrmlzX [IPF x g y h]) (InnerProductArgument.hs:32-49)."""
    s: int
    ny: int
    q: int
    q_inv: int
    nx: int
    body: List[Tuple[int, Point, int, Point]]

    @staticmethod
    def make(s, q, ss0, gs0, ss1, gs1) -> "InnerProduct":
        """makeIP (:47-49)."""
        n = max(len(ss0), len(gs0), len(ss1), len(gs1))
        g = lambda xs, i, z: xs[i] if i < len(xs) else z
        body = [(g(ss0, i, 0) % N, g(gs0, i, None), g(ss1, i, 0) % N, g(gs1, i, None)) for i in range(n)]
        return InnerProduct(s % N, 1, q % N, inv_mod(q, N), 1, body)

    def open_terms(self):
        """openWith (IPF x g y h) = x·g + y·h (:37-38); normalisation is not part of the commitment."""
        out = []
        for x, g, y, h in self.body:
            out += [(x, g), (y, h)]
        return out

    def eval_scalar(self) -> int:
        """evalScalar (:63-66)."""
        scs = [x * y % N for x, _, y, _ in self.body]
        return self.s * self.nx % N * self.ny % N * dot_zip(scs, powers1(self.q, len(scs))) % N

    @staticmethod
    def make_es(e: int) -> Tuple[int, int]:
        """makeEs (:68)."""
        return inv_mod(e, N), e % N

    def make_scalars_coms(self):
        """makeScalarsComs (:70-81) via foldLR (:17-26)."""
        q, qi = self.q, self.q_inv
        q2 = q * q % N
        s, sl, sr = 1, 0, 0
        wl, wr = [], []
        zero = (0, None, 0, None)
        for (xl, gl, yl, hl), (xr, gr, yr, hr) in _halves(self.body, zero):
            sl = (sl + s * xl * yr) % N
            sr = (sr + s * xr * yl) % N
            wl.append((qi * xl % N, gr, yr, hl))
            wr.append((q * xr % N, gl, yl, hr))
            s = s * q2 % N
        k = self.s * self.nx % N * self.ny % N
        sL = k * q % N * sl % N
        sR = k * q2 % N * sr % N
        mk = lambda t, b: InnerProduct(self.s, self.ny, q2, qi * qi % N, t * self.nx % N, b)
        return sL, mk(1, wl), sR, mk(qi, wr)

    def get_witness(self) -> List[int]:
        """getWitness (:83-84)."""
        out = []
        for x, _, y, _ in self.body:
            out += [self.nx * x % N, self.ny * y % N]
        return out

    def collapse(self, e: int, ec) -> "InnerProduct":
        """collapse (:86-101)."""
        q, qi = self.q, self.q_inv
        ei = inv_mod(e, N)
        a1, b1 = rational_reduce_scalar(qi * ei % N)
        b0 = b1 % N
        b0i = inv_mod(b0, N)
        c1, d1 = rational_reduce_scalar(e)
        d0 = d1 % N
        d0i = inv_mod(d0, N)
        body = []
        zero = (0, None, 0, None)
        for (xl, gl, yl, hl), (xr, gr, yr, hr) in _halves(self.body, zero):
            body.append((b0i * (xl + e * q % N * xr) % N, ec.pair_ip(b1, gl, a1, gr), d0i * (yl + ei * yr) % N, ec.pair_ip(d1, hl, c1, hr)))
        return InnerProduct(self.s, self.ny * d0 % N, q * q % N, qi * qi % N, self.nx * b0 % N * qi % N, body)

    @staticmethod
    def expand_challenges(es_y, wit: "InnerProduct", pub: "InnerProduct", basis: "InnerProduct"):
        """expandChallenges (:103-124)."""
        s, q = pub.s, pub.q
        qF = q
        for _ in range(len(es_y)):
            qF = qF * qF % N
        es_x = [inv_mod(e, N) for e in es_y]
        vx = [wit.nx * x % N for x, _, _, _ in wit.body]
        vy = [wit.ny * y % N for _, _, y, _ in wit.body]
        sc = s * weighted_dot_zip(powers1(qF, len(vx)), vx, vy) % N
        tx = tensor(vx, es_x, lambda k: pow(q, 2**k, N))
        ty = tensor(vy, es_y, lambda k: 1)
        pairs = list(zip(pub.body, basis.body))
        ts = list(zip(tx, ty))
        body = zip_with_def1(lambda pg, e_: ((pg[0][0] - e_[0]) % N, pg[1][1], (pg[0][2] - e_[1]) % N, pg[1][3]), (0, 0), pairs, ts)
        return sc, InnerProduct(s, 1, qF, inv_mod(qF, N), 1, body)


@dataclass
class LinearIP:
    """Linear of the IP flavour (InnerProductArgument.hs:132-181)."""
    n: int
    body: List[Tuple[int, int, Point]]

    @staticmethod
    def make(cs, ss, gs) -> "LinearIP":
        l = Linear.make(cs, ss, gs)
        return LinearIP(l.n, l.body)

    def open_terms(self):
        return [(x, g) for _, x, g in self.body]

    def eval_scalar(self) -> int:
        return sum(c * x for c, x, _ in self.body) % N

    def make_scalars_coms(self):
        """makeScalarsComs (:155-158): half-length openings."""
        sl, sr = 0, 0
        wl, wr = [], []
        for (cl, xl, gl), (cr, xr, gr) in _halves(self.body, (0, 0, None)):
            sl = (sl + cr * xl) % N
            sr = (sr + cl * xr) % N
            wl.append((cr, xl, gr))
            wr.append((cl, xr, gl))
        return sl, LinearIP(self.n, wl), sr, LinearIP(self.n, wr)

    def get_witness(self) -> List[int]:
        return [self.n * x % N for _, x, _ in self.body]

    def collapse(self, e: int, ec) -> "LinearIP":
        """collapse (:162-170): rationalReduceScalar of 1/e."""
        a1, b1 = rational_reduce_scalar(inv_mod(e, N))
        a0, b0 = a1 % N, b1 % N
        b0i = inv_mod(b0, N)
        body = []
        for (cl, xl, gl), (cr, xr, gr) in _halves(self.body, (0, 0, None)):
            body.append(((b0 * cl + a0 * cr) % N, (b0i * xl + e * b0i % N * xr) % N, ec.pair_ip(b1, gl, a1, gr)))
        return LinearIP(self.n * b0 % N, body)

    @staticmethod
    def expand_challenges(es1, wit: "LinearIP", pub: "LinearIP", basis: "LinearIP"):
        """expandChallenges (:172-181): challenges inverted first."""
        es = [inv_mod(e, N) for e in es1]
        sc, l = Linear.expand_challenges(es, Linear(wit.n, wit.body), Linear(pub.n, pub.body), Linear(basis.n, basis.body))
        return sc, LinearIP(l.n, l.body)


def ip_make_norm(r: int, ss: Sequence[int], gs: Sequence[Point], ec) -> InnerProduct:
    """makeNorm of the IP flavour (:194-206): pairs (s0,g0),(s1,g1) -> IPF x' g' y' h' with the basis change
    g' = g1 + r·g0, h' = g1 - r·g0; state IP 4 1 q q^-1 with q = r^4."""
    r %= N
    q = pow(r, 4, N)
    half, r2i = inv_mod(2, N), inv_mod(2 * r, N)
    items = zip_with_def2(lambda s, g: (s % N, g), 0, None, list(ss), list(gs))
    body = []
    for (s0, g0), (s1, g1) in _halves(items, (0, None)):
        p = ec.mul(r, g0)                                   # commit (CP r g0) (:204)
        body.append(((r2i * s0 + half * s1) % N, ec.add(g1, p), (-r2i * s0 + half * s1) % N, ec.add(g1, ec.neg(p))))
    return InnerProduct(4, 1, q, inv_mod(q, N), 1, body)


def ip_norm_get_witness(ip: InnerProduct) -> List[int]:
    """Norm.getWitness of the IP flavour (:222-223)."""
    out = []
    for x, _, y, _ in ip.body:
        out += [(ip.nx * x - ip.ny * y) % N, (ip.nx * x + ip.ny * y) % N]
    return out


@dataclass
class NormLinearIP:
    """NormLinear of the IP flavour (:239-267): BPCompose (Norm f) (Linear f)."""
    s: int
    norm: InnerProduct
    lin: LinearIP

    @staticmethod
    def make(s, r, cs, nss, ngs, lss, lgs, ec) -> "NormLinearIP":
        return NormLinearIP(s % N, ip_make_norm(r, nss, ngs, ec), LinearIP.make(cs, lss, lgs))

    def open_terms(self):
        return self.norm.open_terms() + self.lin.open_terms()

    @staticmethod
    def make_es(e: int):
        return InnerProduct.make_es(e)

    def eval_scalar(self) -> int:
        return self.s * (self.norm.eval_scalar() + self.lin.eval_scalar()) % N

    def make_scalars_coms(self):
        a = self.norm.make_scalars_coms()
        b = self.lin.make_scalars_coms()
        return (a[0] + b[0]) % N, NormLinearIP(self.s, a[1], b[1]), (a[2] + b[2]) % N, NormLinearIP(self.s, a[3], b[3])

    def get_witness(self) -> List[int]:
        return [self.s * w % N for w in ip_norm_get_witness(self.norm) + self.lin.get_witness()]

    def collapse(self, e: int, ec) -> "NormLinearIP":
        return NormLinearIP(self.s, self.norm.collapse(e, ec), self.lin.collapse(e, ec))

    @staticmethod
    def expand_challenges(es, wit: "NormLinearIP", pub: "NormLinearIP", basis: "NormLinearIP"):
        sa, na = InnerProduct.expand_challenges(es, wit.norm, pub.norm, basis.norm)
        sb, lb = LinearIP.expand_challenges(es, wit.lin, pub.lin, basis.lin)
        return (sa + sb) % N, NormLinearIP(pub.s, na, lb)


def verify_terms_generic(cls, init_terms, es, rs, pub: PSV, basis: PSV, wit_body):
    """verifyBPM's term list (Bulletproof.hs:375-377, :362-368) for any BPOpening class with expand_challenges / make_es."""
    sc, chs = cls.expand_challenges(es, wit_body, pub.body, basis.body)
    wit1 = PSV((pub.sc - sc) % N, basis.g, chs)
    terms = wit1.open_terms() + list(init_terms)
    for e, (x, r) in zip(es, rs):
        e0, e1 = cls.make_es(e)
        terms += [(e0, x), (e1, r)]
    return terms


# ============================================================================= GLV path
# Restatement of the reference's optional endomorphism path: SplitScalar (FastPrime p) (src/Commitment.hs:293-306),
# FastInnerProduct (Point .. (FastPrime p)) (src/Commitment.hs:374-398), decomposeFastPrimeEis
# (src/Data/Field/Galois/FastPrime.hs:186-205), Eisenstein integers (src/Data/Field/Eis.hs:20-41) and cmConj
# (src/Data/Curve/CM.hs:25-33).  Not the CLI default (app/Main.hs:17-21 keeps the import commented out).
CHAR_EIS_FR = (303414439467246543595250775667605759171, -64502973549206556628585045361533709077)   # FastSECP256K1.hs:56


def eis_conj(e):
    """conjEis (Eis.hs:20-21)."""
    return (e[0] - e[1], -e[1])


def eis_mul(x, y):
    """(*) of Eis (Eis.hs:30-34)."""
    a, b, c = x[0] * y[0], x[1] * y[1], (x[0] - x[1]) * (y[0] - y[1])
    return (a - b, a - c)


def decompose_eis(x: int) -> Tuple[int, int]:
    """decomposeFastPrimeEis (FastPrime.hs:186-205): x = a + b·λ (mod n) with |a|, |b| ≲ 2^128."""
    p_fac = eis_conj(CHAR_EIS_FR)
    x_int = (x % N, 0)
    u, v = eis_mul(x_int, eis_conj(p_fac))

    def rnd(nn, q):
        r = nn - N * q
        if abs(r) > abs(r + N):
            return q - 1
        if abs(r) > abs(r - N):
            return q + 1
        return q
    q = (rnd(u, u >> 256), rnd(v, v >> 256))
    m = eis_mul(q, p_fac)
    return (x_int[0] - m[0], x_int[1] - m[1])


def cm_mul(p: Point) -> Point:
    """cmConj for affine points (CM.hs:25-27): (x, y) -> (β·x, y) = λ·(x, y)."""
    return None if p is None else (BETA * p[0] % P, p[1])


def glv_inner_product(sgs: Sequence[Tuple[int, Point]], ec) -> Point:
    """innerProduct through the FastPrime instances: 129 rows (Commitment.hs:304), digit pairs (:306), basis
    (p00, p11) with the sign difference flag (:387-398), addBasis (:382-385).
    NOTE the reference mis-signs the b-only digit when a == 0 exactly (signum 0 /= signum b is always True, so a
    positive b is subtracted); that input has probability ~2^-128 and is not reproduced here on purpose: the
    restatement follows the reference for every a != 0."""
    terms = []
    for s, g in sgs:
        a, b = decompose_eis(s)
        s0 = (a > 0) - (a < 0)
        s1 = (b > 0) - (b < 0)
        g1 = ec.neg(g) if s0 == -1 else g
        if s0 == s1:
            h1 = ec.neg(cm_mul(cm_mul(g1)))                 # -λ²g' = g' + λg'
        else:
            h1 = ec.add(g1, ec.neg(cm_mul(g1)))
        diff = s0 != s1
        if s0 == 0 and s1 == 1:
            diff = False                                    # the a == 0 corner (see docstring)
        terms.append((abs(a), abs(b), diff, g1, h1))
    v = None
    for row in range(129, 0, -1):
        v = ec.add(v, v)
        for a, b, diff, p00, p11 in terms:
            da, db = (a >> (row - 1)) & 1, (b >> (row - 1)) & 1
            if da and db:
                v = ec.add(p11, v)
            elif da:
                v = ec.add(p00, v)
            elif db:
                q = cm_mul(p00)
                v = ec.add(ec.neg(q) if diff else q, v)
    return v


# ============================================================================= Eisenstein rational reduction and pair fold
# rationalReduceScalar for the FastPrime configuration: the class default (src/Commitment.hs:242-255) over SplitScalar (FastPrime p)
# (:293-306): reducedChar = conjEis . charEis, normScalar = normEis, reduceScalar = decomposeEis; `quot` is the Integral (Eis a)
# instance's nearest-integer division (src/Data/Field/Eis.hs:72-82).  projectivePairIP (:343-353) then runs 65 rows
# (rationalReducedScalarLength, :304) over the FastInnerProduct instance of FastPrime points (:374-398).


def eis_norm(e) -> int:
    """normEis (Eis.hs:23-24)"""
    return e[0] * e[0] - e[0] * e[1] + e[1] * e[1]


def eis_sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def eis_quot(x, m):
    """quot of Integral (Eis a) (Eis.hs:72-82): (x * conj m) / norm m, each component rounded to the nearest integer"""
    m_n = eis_norm(m)
    u, v = eis_mul(x, eis_conj(m))

    def rnd(n):
        q, r = divmod(n, m_n)               # Haskell divMod: floor, 0 <= r < m_n
        sg = (r > 0) - (r < 0)
        return q + sg if m_n - abs(r) < abs(r) else q
    return (rnd(u), rnd(v))


def eis_recompose(e) -> int:
    """recomposeEis (Eis.hs:59-60) in Fr"""
    return (e[0] + LAMBDA * e[1]) % N


def rational_reduce_scalar_eis(x: int):
    """(a, b) as Eisenstein integers with x = a / b: egcd (pRed, 0) (reduceScalar x, 1), the list starting at its second argument
    (Commitment.hs:252), first (r, s) with (normEis r)^2 <= 2n (:247)."""
    prev = (eis_conj(CHAR_EIS_FR), (0, 0))
    cur = (decompose_eis(x), (1, 0))
    while eis_norm(cur[0]) ** 2 > 2 * N:
        q = eis_quot(prev[0], cur[0])
        nxt = (eis_sub(prev[0], eis_mul(q, cur[0])), eis_sub(prev[1], eis_mul(q, cur[1])))
        prev, cur = cur, nxt
    return cur


def pair_ip_eis(s0, g0: Point, s1, g1: Point, ec) -> Point:
    """projectivePairIP (s0, g0) (s1, g1) (Commitment.hs:343-353) for Eisenstein reduced scalars: normalizeBasis of the FastPrime
    instance (:387-398: absolute components, sign folded into p00, p11 = p00 +- lambda p00), 65 rows of dbl' + addBasis (:382-385).
    (The a == 0 corner of that normalizeBasis is mis-signed in the reference, see glv_inner_product; the group element is computed.)"""
    terms = []
    for (a, b), g in ((s0, g0), (s1, g1)):
        sa, sb = (a > 0) - (a < 0), (b > 0) - (b < 0)
        p00 = ec.neg(g) if sa == -1 else g
        lam = cm_mul(p00)
        if sa == sb:
            p11, diff = ec.neg(cm_mul(lam)), False
        else:
            p11, diff = ec.add(p00, ec.neg(lam)), True
        if sa == 0 and sb == 1:
            diff = False
        terms.append((abs(a), abs(b), diff, p00, p11))
    assert all(a < 2**65 and b < 2**65 for a, b, _, _, _ in terms), "component exceeds rationalReducedScalarLength"
    v = None
    for row in range(65, 0, -1):
        v = ec.add(v, v)
        for a, b, diff, p00, p11 in terms:
            da, db = (a >> (row - 1)) & 1, (b >> (row - 1)) & 1
            if da and db:
                v = ec.add(p11, v)
            elif da:
                v = ec.add(p00, v)
            elif db:
                lam = cm_mul(p00)
                v = ec.add(ec.neg(lam) if diff else lam, v)
    return v
