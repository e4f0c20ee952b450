#!/usr/bin/env python3
"""Bit-exact Python model of csrc/fe26.cuh (10 x 26-bit limbs, lazy magnitudes), with assertions that every
intermediate fits the 32/64-bit register it lives in on the GPU.  Used to validate the algorithm (random,
edge and worst-case-magnitude inputs) before/after touching the HIP code:  python benchmarks/fe26_model.py"""
import random

P = 2**256 - 2**32 - 977
M26 = (1 << 26) - 1
M22 = (1 << 22) - 1
R0, R1 = 0x3D10, 0x400          # 2^260 = R1*2^26 + R0 (mod p)
PL = [0x3FFFC2F, 0x3FFFFBF] + [M26] * 7 + [M22]   # limbs of p
U64 = (1 << 64) - 1


def u64(x):
    assert 0 <= x <= U64, f"64-bit overflow: {x.bit_length()} bits"
    return x


def u32(x):
    assert 0 <= x < (1 << 32), f"32-bit overflow: {x.bit_length()} bits"
    return x


def val(a):
    return sum(v << (26 * i) for i, v in enumerate(a))


def from_int(x):
    return [(x >> (26 * i)) & M26 for i in range(10)]


def mag_ok(a, m):
    return all(a[i] <= 2 * m * M26 for i in range(9)) and a[9] <= 2 * m * M22


def mul(a, b, sqr=False):
    """inputs magnitude <= 8 (limbs < 2^30); output magnitude 1"""
    def col(k, acc):
        for i in range(max(0, k - 9), min(9, k) + 1):
            acc = u64(acc + a[i] * b[k - i])
        return acc
    # high columns 9..18 with running carry d
    d = col(9, 0)
    t9 = d & M26; d >>= 26
    u = []
    for k in range(10, 19):
        d = col(k, d)
        u.append(d & M26); d >>= 26
    u.append(d)                      # u[9]: leftover carry (< 2^38)
    assert u[9] < (1 << 38)
    # low columns 0..8 folding u_k*R0 + u_{k-1}*R1
    r = [0] * 10
    c = 0
    for k in range(9):
        c = col(k, c)
        c = u64(c + u[k] * R0)
        if k:
            c = u64(c + u[k - 1] * R1)
        r[k] = c & M26; c >>= 26
    c = u64(c + t9 + u[9] * R0 + u[8] * R1)
    r[9] = c & M22
    top = u64((c >> 22) + ((u[9] * R1) << 4))      # units of 2^256
    # fold 2^256 = 0x1000003D1 = 2^32 + 0x3D1
    c = u64(r[0] + top * 0x3D1); r[0] = c & M26; c >>= 26
    c = u64(c + r[1] + (top << 6)); r[1] = c & M26; c >>= 26
    c = u64(c + r[2]); r[2] = c & M26; c >>= 26
    r[3] = u32(r[3] + c)
    assert mag_ok(r, 1), r
    return r


def sqr(a):
    return mul(a, a)


def add(a, b):
    return [u32(x + y) for x, y in zip(a, b)]


def negate(a, m):
    """magnitude m in, m+1 out"""
    assert mag_ok(a, m)
    return [u32(2 * (m + 1) * PL[i] - a[i]) for i in range(10)]


def mul_int(a, k):
    return [u32(x * k) for x in a]


def normalize(a):
    """full normalisation to the canonical representative (any magnitude <= 32)"""
    t = list(a)
    x = t[9] >> 22; t[9] &= M22
    t[0] += x * 0x3D1; t[1] += x << 6
    for i in range(9):
        t[i + 1] += t[i] >> 26; t[i] &= M26
    # now < 2^256 + small; one more possible fold of bit 256
    x = t[9] >> 22; t[9] &= M22
    t[0] += x * 0x3D1; t[1] += x << 6
    for i in range(9):
        t[i + 1] += t[i] >> 26; t[i] &= M26
    assert t[9] >> 22 == 0
    v = val(t)
    assert v < 2**256
    if v >= P:                       # final conditional subtraction
        v -= P
    return from_int(v)


def normalizes_to_zero(a):
    return val(normalize(a)) == 0


def rand_mag(rnd, m, extreme=False):
    if extreme:
        return [2 * m * M26] * 9 + [2 * m * M22]
    return [rnd.randrange(2 * m * M26 + 1) for _ in range(9)] + [rnd.randrange(2 * m * M22 + 1)]


def main():
    rnd = random.Random(26)
    for it in range(20000):
        ma, mb = rnd.choice([1, 2, 3, 5, 8]), rnd.choice([1, 2, 4, 7, 8])
        ext = it < 50
        a, b = rand_mag(rnd, ma, ext), rand_mag(rnd, mb, ext and it % 2 == 0)
        if it % 7 == 0:
            a = from_int(rnd.choice([0, 1, P - 1, P, 2**256 - 1 if False else P - 2]) % (1 << 260))
        r = mul(a, b)
        assert val(r) % P == val(a) * val(b) % P
        assert val(normalize(r)) == val(a) * val(b) % P
        s = mul(a, a)
        assert val(s) % P == val(a) ** 2 % P
        n = negate(a, ma)
        assert (val(n) + val(a)) % P == 0 and mag_ok(n, ma + 1)
        if ma + mb <= 16:
            assert val(normalize(add(a, b))) == (val(a) + val(b)) % P
    # normalize edge cases
    for v in [0, 1, P - 1, P, P + 1, 2**256 - 1, 2**256, 2 * P, 2 * P + 5]:
        t = from_int(v) if v < (1 << 260) else None
        assert val(normalize(t)) == v % P
    print("fe26 model OK")


if __name__ == "__main__":
    main()
