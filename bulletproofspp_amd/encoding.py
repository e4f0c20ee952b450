"""The reference's wire format for proofs and commitments (src/Encoding.hs, src/RangeProof.hs:60-85) — SURVEY.md 8(f) rank 2.

Layout (all of it follows from Data.Binary's big-endian `Word` and the instances at Encoding.hs:75-134):
  * a field element (coordinate or scalar) is FOUR 64-bit words, least-significant word FIRST, each word big-endian
    (`Binary (Prime p)`, :75-86);
  * a list of n points is ceil(n / 8) sign bytes — bit k of byte j is the sign of point 8j + k, sign = (y > p - y) — followed by
    the n x coordinates (`encodeCommitments` / `decodeCommitments`, :119-134; `bitPack` :105-110);
  * a proof file is the final witness scalars (norm part, then linear part) followed by the points rpComs ++ bpComs, where
    bpComs are the argument's responses flattened (X, R per round, LAST round first) (`encodeProof'`, RangeProof.hs:60-66);
  * the commitments file is the list of input commitments alone (app/Main.hs:192-194).
Decoding needs a square root per point (`fromXWithSign`, :97-103): a data-parallel job done by the backend's `lift_x`
(`bppp_lift_x_device` on the GPU), the sign fix-up on the host.  The infinity point has no encoding in the reference either.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .rangeproof import N, Point, RangeProof, decode_field

P = 2**256 - 2**32 - 977


def put_field(v: int) -> bytes:
    """Binary (Prime p) put (Encoding.hs:81-86)"""
    return b"".join(((v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "big") for i in range(4))


def get_field(b: bytes, modulus: int) -> int:
    """Binary (Prime p) get (:76-80): toP reduces"""
    return decode_field(b, modulus)


def encode_commitments(pts: Sequence[Point]) -> bytes:
    """encodeCommitments (Encoding.hs:130-134)"""
    if any(p is None for p in pts):
        raise ValueError("the point at infinity has no encoding")
    signs = [p[1] > P - p[1] for p in pts]
    packed = bytearray((len(pts) + 7) // 8)
    for i, s in enumerate(signs):
        if s:
            packed[i >> 3] |= 1 << (i & 7)
    return bytes(packed) + b"".join(put_field(p[0]) for p in pts)


def decode_commitments(n: int, data: bytes, lift_x) -> Optional[Tuple[List[Point], int]]:
    """decodeCommitments (Encoding.hs:119-128): (points, bytes consumed) or None when some x is not on the curve.
    `lift_x(xs)` returns for every x a point with that x (either root) or None."""
    n_sign = (n + 7) // 8
    need = n_sign + 32 * n
    if len(data) < need:
        return None
    xs = [get_field(data[n_sign + 32 * i:n_sign + 32 * i + 32], P) for i in range(n)]
    roots = lift_x(xs)
    out: List[Point] = []
    for i, (x, r) in enumerate(zip(xs, roots)):
        if r is None:
            return None
        want_big = bool((data[i >> 3] >> (i & 7)) & 1)
        y = r[1]
        if (y > P - y) != want_big:                      # fromXWithSign (:97-103)
            y = P - y
        out.append((x, y))
    return out, need


def encode_proof(num_rp_coms: int, proof: RangeProof) -> Tuple[bytes, bytes]:
    """encodeProof' (RangeProof.hs:60-66): (commitments file, proof file)"""
    rp_coms, n_coms = proof.coms[:num_rp_coms], proof.coms[num_rp_coms:]
    bp_coms = [p for xr in proof.responses for p in xr]
    scalars = list(proof.wit_nrm) + list(proof.wit_lin)
    return encode_commitments(n_coms), b"".join(put_field(s % N) for s in scalars) + encode_commitments(list(rp_coms) + bp_coms)


def decode_proof(num_rp_coms: int, rounds: int, final_lens: Tuple[int, int], n_coms: Sequence[Point], data: bytes, lift_x) -> Optional[RangeProof]:
    """decodeProof' (RangeProof.hs:68-85): the number of rounds and the final witness lengths come from the setup
    (optimalWitnessSize); None on malformed input."""
    num_nrm, num_lin = final_lens
    n_sc = num_nrm + num_lin
    if len(data) < 32 * n_sc:
        return None
    scs = [get_field(data[32 * i:32 * i + 32], N) for i in range(n_sc)]
    num_coms = num_rp_coms + 2 * rounds
    dec = decode_commitments(num_coms, data[32 * n_sc:], lift_x)
    if dec is None:
        return None
    coms, _ = dec
    rp_coms, bp_coms = coms[:num_rp_coms], coms[num_rp_coms:]
    resps = [(bp_coms[2 * i], bp_coms[2 * i + 1]) for i in range(rounds)]
    return RangeProof(list(rp_coms) + list(n_coms), resps, scs[:num_nrm], scs[num_nrm:])


def encode_wide(pts: Sequence[Point]) -> bytes:
    """The points file of the CLI: `encodeFile "points.bin" $ take n $ WE <$> ps` (app/Main.hs:260-262) with the `WideEncoding`
    instance put (WE p) = put x <> put y (app/Main.hs:89-98).  The value encoded is a LIST, so Data.Binary's list instance
    applies: an 8-byte big-endian element count, then every point as both coordinates in full (Binary (Prime p), 32 bytes
    each) — no sign bytes."""
    if any(p is None for p in pts):
        raise ValueError("the point at infinity has no encoding")
    return len(pts).to_bytes(8, "big") + b"".join(put_field(p[0]) + put_field(p[1]) for p in pts)


def decode_wide(data: bytes, check: bool = True) -> List[Point]:
    """`map getWE <$> decodeFile` (app/Main.hs:260): count prefix, then 64 bytes per point, x then y.  `fromA (A x y)`
    (app/Main.hs:94-98) does not check the curve equation; `check=True` (default) does and raises ValueError."""
    if len(data) < 8:
        raise ValueError("points file too short")
    n = int.from_bytes(data[:8], "big")
    if len(data) < 8 + 64 * n:
        raise ValueError("points file too short")
    out: List[Point] = []
    for i in range(n):
        o = 8 + 64 * i
        x, y = get_field(data[o:o + 32], P), get_field(data[o + 32:o + 64], P)
        if check and (y * y - x * x * x - 7) % P:
            raise ValueError("point %d is not on the curve" % i)
        out.append((x, y))
    return out


def gpu_lift_x(gpu):
    """lift_x for decode_*: square roots of x^3 + 7 on the GPU (bppp_lift_x_device; even root, the caller fixes the sign)"""
    import numpy as np
    from .capi import array_to_point, scalars_to_array

    def fn(xs: Sequence[int]) -> List[Point]:
        if not xs:
            return []
        d_x = gpu.to_device(scalars_to_array([x % P for x in xs]))
        d_p = gpu.to_device(np.zeros((len(xs), 8), dtype=np.uint64))
        try:
            gpu.lift_x(d_x, len(xs), d_p)
            out = gpu.download(d_p, (len(xs), 8))
        finally:
            gpu.free(d_x); gpu.free(d_p)
        return [array_to_point(out[i]) for i in range(len(xs))]
    return fn
