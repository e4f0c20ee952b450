// hostmath.hpp — host-side 256-bit arithmetic used by the PRODUCT (not the oracle).
//
// The device does the data-parallel work; the host keeps only the latency-bound scalar glue the
// reference also evaluates once per call/round:
//   * the final window combine of an MSM (<= 33 points; a 256-doubling dependency chain is
//     ~6x faster on one CPU core than on one GPU lane),
//   * rationalReduceScalar (src/Commitment.hs:242-255): one half-GCD per round,
//   * Fr scalar glue of the round driver (makeEs, normalisation updates; src/Bulletproof.hs:346-378).
// Independent of oracle/ by construction: different representation (u256 class, Montgomery-free
// folding with __int128) and no shared source.
#pragma once
#include <stdint.h>
#include <string.h>
#include <utility>

namespace bppp_host {

typedef unsigned __int128 u128;

struct U256 {
  uint64_t w[4];
  static U256 zero() { U256 r; memset(&r, 0, sizeof r); return r; }
  static U256 one() { U256 r = zero(); r.w[0] = 1; return r; }
  static U256 from_u64(uint64_t x) { U256 r = zero(); r.w[0] = x; return r; }
  static U256 load(const uint64_t *p) { U256 r; memcpy(r.w, p, 32); return r; }
  void store(uint64_t *p) const { memcpy(p, w, 32); }
  bool is_zero() const { return (w[0] | w[1] | w[2] | w[3]) == 0; }
  bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
  bool operator==(const U256 &o) const { return memcmp(w, o.w, 32) == 0; }
};
inline int cmp(const U256 &a, const U256 &b) {
  for (int i = 3; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; }
  return 0;
}
inline uint64_t add_raw(U256 &r, const U256 &a, const U256 &b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
inline uint64_t sub_raw(U256 &r, const U256 &a, const U256 &b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  return br;
}

// Modulus of shape 2^256 - R, R up to 3 limbs.
struct Mod { U256 m; uint64_t r[3]; };
inline const Mod &FQ() {
  static const Mod M = {{{0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL}}, {0x1000003D1ULL, 0, 0}};
  return M;
}
inline const Mod &FR() {
  static const Mod M = {{{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, ~0ULL}},
                        {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1}};
  return M;
}

inline U256 madd(const U256 &a, const U256 &b, const Mod &M) {
  U256 s; uint64_t c = add_raw(s, a, b);
  if (c || cmp(s, M.m) >= 0) sub_raw(s, s, M.m);
  return s;
}
inline U256 msub(const U256 &a, const U256 &b, const Mod &M) {
  U256 d; if (sub_raw(d, a, b)) add_raw(d, d, M.m);
  return d;
}
inline U256 mneg(const U256 &a, const Mod &M) { return a.is_zero() ? a : msub(U256::zero(), a, M); }
inline U256 mmul(const U256 &a, const U256 &b, const Mod &M) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.w[i] * b.w[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  while (t[4] | t[5] | t[6] | t[7]) {
    uint64_t n[8] = {t[0], t[1], t[2], t[3], 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 3; j++) { c += (u128)t[4 + i] * M.r[j] + n[i + j]; n[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + 3; k < 8 && c; k++) { c += n[k]; n[k] = (uint64_t)c; c >>= 64; }
    }
    memcpy(t, n, sizeof t);
  }
  U256 r = {{t[0], t[1], t[2], t[3]}};
  while (cmp(r, M.m) >= 0) sub_raw(r, r, M.m);
  return r;
}
// dedicated Fq multiply (p = 2^256 - 0x1000003D1): the final window combine of every MSM runs ~2,500 of these
inline U256 fqmul(const U256 &a, const U256 &b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.w[i] * b.w[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  const uint64_t R = 0x1000003D1ULL;
  // fold 1: lo + hi * R  (hi * R < 2^289)
  uint64_t r[5];
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * R + t[i]; r[i] = (uint64_t)c; c >>= 64; }
  r[4] = (uint64_t)c;
  // fold 2: r[4] * R (< 2^66) into the low limbs
  c = (u128)r[4] * R;
  U256 o;
  for (int i = 0; i < 4; i++) { c += r[i]; o.w[i] = (uint64_t)c; c >>= 64; }
  // a carry out of 2^256 folds once more and cannot carry again
  if ((uint64_t)c) { u128 d = (u128)o.w[0] + R; o.w[0] = (uint64_t)d; d >>= 64; for (int i = 1; i < 4 && d; i++) { d += o.w[i]; o.w[i] = (uint64_t)d; d >>= 64; } }
  if (cmp(o, FQ().m) >= 0) sub_raw(o, o, FQ().m);
  return o;
}

// dedicated Fr multiply (n = 2^256 - R, R = 2^128 + r1 2^64 + r0): the range-proof prover runs ~30 k of these per proof on the host
inline U256 frmul(const U256 &a, const U256 &b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.w[i] * b.w[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  const uint64_t r0 = 0x402DA1732FC9BEBFULL, r1 = 0x4551231950B75FC4ULL;
  // fold 1: lo + hi * R, hi = t[4..7];  hi * R = hi * (r1:r0) + (hi << 128)   ->  7 limbs (< 2^386)
  uint64_t f[7] = {t[0], t[1], t[2], t[3], 0, 0, 0};
  {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * r0 + f[i]; f[i] = (uint64_t)c; c >>= 64; }
    for (int k = 4; k < 7; k++) { c += f[k]; f[k] = (uint64_t)c; c >>= 64; }
    c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * r1 + f[i + 1]; f[i + 1] = (uint64_t)c; c >>= 64; }
    for (int k = 5; k < 7; k++) { c += f[k]; f[k] = (uint64_t)c; c >>= 64; }
    c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)f[i + 2] + t[4 + i]; f[i + 2] = (uint64_t)c; c >>= 64; }
    f[6] += (uint64_t)c;
  }
  // fold 2: hi2 = f[4..6] (< 2^130) times R -> < 2^260
  uint64_t g[5] = {f[0], f[1], f[2], f[3], 0};
  {
    u128 c = 0;
    for (int i = 0; i < 3; i++) { c += (u128)f[4 + i] * r0 + g[i]; g[i] = (uint64_t)c; c >>= 64; }
    for (int k = 3; k < 5; k++) { c += g[k]; g[k] = (uint64_t)c; c >>= 64; }
    c = 0;
    for (int i = 0; i < 3; i++) { c += (u128)f[4 + i] * r1 + g[i + 1]; g[i + 1] = (uint64_t)c; c >>= 64; }
    g[4] += (uint64_t)c;
    c = 0;
    for (int i = 0; i < 3; i++) { c += (u128)g[i + 2] + f[4 + i]; g[i + 2] = (uint64_t)c; c >>= 64; }
  }
  // fold 3: g[4] (a few bits) times R; a carry out of 2^256 folds once more and cannot carry again
  U256 o;
  {
    const uint64_t h = g[4];
    u128 c = (u128)h * r0 + g[0]; o.w[0] = (uint64_t)c; c >>= 64;
    c += (u128)h * r1 + g[1]; o.w[1] = (uint64_t)c; c >>= 64;
    c += (u128)g[2] + h; o.w[2] = (uint64_t)c; c >>= 64;
    c += g[3]; o.w[3] = (uint64_t)c; c >>= 64;
    if ((uint64_t)c) {
      u128 d = (u128)o.w[0] + r0; o.w[0] = (uint64_t)d; d >>= 64;
      d += (u128)o.w[1] + r1; o.w[1] = (uint64_t)d; d >>= 64;
      d += (u128)o.w[2] + 1; o.w[2] = (uint64_t)d; d >>= 64;
      o.w[3] += (uint64_t)d;
    }
  }
  while (cmp(o, FR().m) >= 0) sub_raw(o, o, FR().m);
  return o;
}

inline U256 mpow(const U256 &a, const U256 &e, const Mod &M) {
  U256 acc = U256::one(), base = a;
  for (int i = 0; i < 256; i++) { if (e.bit(i)) acc = mmul(acc, base, M); base = mmul(base, base, M); }
  return acc;
}
inline U256 minv(const U256 &a, const Mod &M) {  // 0 -> 0 (BatchInverse.hs:18,23)
  U256 e; sub_raw(e, M.m, U256::from_u64(2));
  return mpow(a, e, M);
}

// Montgomery's trick on the host (0 -> 0): one Fermat inversion for a whole vector (lockstep prover: 2-3 per proof per round)
inline void batch_minv(U256 *v, size_t n, const Mod &M) {
  if (!n) return;
  U256 *pre = new U256[n];
  U256 acc = U256::one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!v[i].is_zero()) acc = mmul(acc, v[i], M); }
  U256 y = minv(acc, M);
  for (size_t i = n; i-- > 0;) {
    if (v[i].is_zero()) continue;
    U256 inv = mmul(y, pre[i], M);
    y = mmul(y, v[i], M);
    v[i] = inv;
  }
  delete[] pre;
}

// a lazily-reduced device value: 10 limbs of radix 2^26 (csrc/fq26.hip.h), any magnitude (each limb < 2^32) -> canonical mod p.
// The limbs are packed into a 320-bit integer (< 2^267) and the part above 2^256 folded back with 2^256 = R (mod p): no field multiply
// (the window combine of one MSM converts up to ~260 coordinates; ten multiply-adds each used to be a third of its host time).
inline U256 from_limbs26(const uint32_t *n) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 10; i++) {
    const int bit = 26 * i, k = bit >> 6, sh = bit & 63;
    const u128 v = (u128)n[i] << sh;
    u128 c = (u128)t[k] + (uint64_t)v;
    t[k] = (uint64_t)c;
    c = (c >> 64) + (uint64_t)(v >> 64) + t[k + 1];
    t[k + 1] = (uint64_t)c;
    if (k + 2 < 6) t[k + 2] += (uint64_t)(c >> 64);        // t[k + 2] is still zero or a carry of the previous limb: cannot overflow
  }
  const uint64_t R = 0x1000003D1ULL;
  u128 c = (u128)t[4] * R;                                  // t[4] < 2^11, t[5] = 0
  U256 o;
  for (int i = 0; i < 4; i++) { c += t[i]; o.w[i] = (uint64_t)c; c >>= 64; }
  if ((uint64_t)c) { u128 d = (u128)o.w[0] + R; o.w[0] = (uint64_t)d; d >>= 64; for (int i = 1; i < 4 && d; i++) { d += o.w[i]; o.w[i] = (uint64_t)d; d >>= 64; } }
  if (cmp(o, FQ().m) >= 0) sub_raw(o, o, FQ().m);
  return o;
}

// ---- curve, Jacobian on the host (only for the <= 65-point window combine and group glue)
struct HAff { U256 x, y; bool inf() const { return x.is_zero() && y.is_zero(); } };
struct HJac { U256 X, Y, Z; bool inf() const { return Z.is_zero(); } };
inline HJac hj_inf() { return {U256::one(), U256::one(), U256::zero()}; }
inline U256 fqinv(const U256 &a) {   // a^(p-2); 0 -> 0
  U256 e; sub_raw(e, FQ().m, U256::from_u64(2));
  U256 acc = U256::one(), base = a;
  for (int i = 0; i < 256; i++) { if (e.bit(i)) acc = fqmul(acc, base); base = fqmul(base, base); }
  return acc;
}
inline HJac hj_dbl(const HJac &p) {
  const Mod &M = FQ();
  if (p.inf() || p.Y.is_zero()) return hj_inf();
  U256 A = fqmul(p.X, p.X), B = fqmul(p.Y, p.Y), C = fqmul(B, B);
  U256 t = madd(p.X, B, M); t = msub(msub(fqmul(t, t), A, M), C, M);
  U256 D = madd(t, t, M), E = madd(madd(A, A, M), A, M), F = fqmul(E, E);
  HJac r;
  r.X = msub(msub(F, D, M), D, M);
  U256 c8 = madd(C, C, M); c8 = madd(c8, c8, M); c8 = madd(c8, c8, M);
  r.Y = msub(fqmul(E, msub(D, r.X, M)), c8, M);
  r.Z = fqmul(p.Y, p.Z); r.Z = madd(r.Z, r.Z, M);
  return r;
}
inline HJac hj_add(const HJac &p, const HJac &q) {  // complete general add (add-2007-bl shape)
  const Mod &M = FQ();
  if (p.inf()) return q;
  if (q.inf()) return p;
  U256 z1z1 = fqmul(p.Z, p.Z), z2z2 = fqmul(q.Z, q.Z);
  U256 u1 = fqmul(p.X, z2z2), u2 = fqmul(q.X, z1z1);
  U256 s1 = fqmul(fqmul(p.Y, q.Z), z2z2), s2 = fqmul(fqmul(q.Y, p.Z), z1z1);
  U256 h = msub(u2, u1, M), r = msub(s2, s1, M);
  if (h.is_zero()) return r.is_zero() ? hj_dbl(p) : hj_inf();
  U256 hh = fqmul(h, h), hhh = fqmul(h, hh), v = fqmul(u1, hh);
  HJac o;
  o.X = msub(msub(msub(fqmul(r, r), hhh, M), v, M), v, M);
  o.Y = msub(fqmul(r, msub(v, o.X, M)), fqmul(s1, hhh), M);
  o.Z = fqmul(fqmul(p.Z, q.Z), h);
  return o;
}
inline HAff hj_to_aff(const HJac &p) {
  if (p.inf()) return {U256::zero(), U256::zero()};
  U256 zi = fqinv(p.Z), zi2 = fqmul(zi, zi);
  return {fqmul(p.X, zi2), fqmul(p.Y, fqmul(zi2, zi))};
}
inline HJac hj_from_aff(const HAff &a) { return a.inf() ? hj_inf() : HJac{a.x, a.y, U256::one()}; }
// XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) is the Jacobian point (X*ZZ, Y*ZZZ, ZZ):
//   x = X*ZZ/ZZ^2 = X/ZZ,  y = Y*ZZZ/ZZ^3 = Y*ZZZ/ZZZ^2 = Y/ZZZ.
inline HJac hj_from_xyzz(const U256 &X, const U256 &Y, const U256 &ZZ, const U256 &ZZZ) {
  if (ZZ.is_zero()) return hj_inf();
  return {fqmul(X, ZZ), fqmul(Y, ZZZ), ZZ};
}

// ---- input validation for the verifier entry points (proof-supplied data is untrusted)
inline bool scalars_canonical(const uint64_t *s, size_t n) {
  for (size_t i = 0; i < n; i++) if (cmp(U256::load(s + 4 * i), FR().m) >= 0) return false;
  return true;
}
// every point is the infinity encoding (0,0) or has canonical coordinates with y^2 = x^3 + 7
inline bool points_on_curve(const uint64_t *p, size_t n) {
  const Mod &M = FQ();
  for (size_t i = 0; i < n; i++) {
    U256 x = U256::load(p + 8 * i), y = U256::load(p + 8 * i + 4);
    if (x.is_zero() && y.is_zero()) continue;
    if (cmp(x, M.m) >= 0 || cmp(y, M.m) >= 0) return false;
    U256 rhs = madd(fqmul(fqmul(x, x), x), U256::from_u64(7), M);
    if (!(fqmul(y, y) == rhs)) return false;
  }
  return true;
}

// ---- signed multi-precision integers for rationalReduceScalar (5 limbs + sign)
struct SInt {
  static const int L = 5;
  uint64_t m[L]; bool neg;
  static SInt zero() { SInt r; memset(r.m, 0, sizeof r.m); r.neg = false; return r; }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < L; i++) o |= m[i]; return o == 0; }
  int bits() const { for (int i = L - 1; i >= 0; i--) if (m[i]) return 64 * i + 64 - __builtin_clzll(m[i]); return 0; }
};
inline int mcmp(const uint64_t *a, const uint64_t *b) {
  for (int i = SInt::L - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
inline void madd_mag(uint64_t *o, const uint64_t *a, const uint64_t *b) {
  u128 c = 0; for (int i = 0; i < SInt::L; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
}
inline void msub_mag(uint64_t *o, const uint64_t *a, const uint64_t *b) {
  uint64_t br = 0;
  for (int i = 0; i < SInt::L; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline SInt sadd(const SInt &a, const SInt &b) {
  SInt r;
  if (a.neg == b.neg) { madd_mag(r.m, a.m, b.m); r.neg = a.neg; }
  else if (mcmp(a.m, b.m) >= 0) { msub_mag(r.m, a.m, b.m); r.neg = a.neg; }
  else { msub_mag(r.m, b.m, a.m); r.neg = b.neg; }
  if (r.is_zero()) r.neg = false;
  return r;
}
inline SInt sneg(SInt a) { if (!a.is_zero()) a.neg = !a.neg; return a; }
inline SInt smul(const SInt &a, const SInt &b) {
  SInt r = SInt::zero();
  for (int i = 0; i < SInt::L; i++) {
    u128 c = 0;
    for (int j = 0; i + j < SInt::L; j++) { c += (u128)a.m[i] * b.m[j] + r.m[i + j]; r.m[i + j] = (uint64_t)c; c >>= 64; }
  }
  r.neg = r.is_zero() ? false : (a.neg != b.neg);
  return r;
}
// truncating division (Haskell `quot`, Commitment.hs:254).  Euclid's quotients are almost always a few bits long, so the
// shift-subtract loop runs only over bits(a) - bits(b) + 1 positions.
inline SInt squot(const SInt &a, const SInt &b) {
  SInt q = SInt::zero();
  int na = a.bits(), nb = b.bits();
  if (nb == 0 || na < nb) return q;
  int sh = na - nb;
  uint64_t rem[SInt::L], d[SInt::L + 1] = {0};
  memcpy(rem, a.m, sizeof rem);
  // d = |b| << sh
  for (int i = 0; i < SInt::L; i++) {
    int w = i + (sh >> 6), o = sh & 63;
    if (w < SInt::L + 1) d[w] |= b.m[i] << o;
    if (o && w + 1 < SInt::L + 1) d[w + 1] |= b.m[i] >> (64 - o);
  }
  for (int i = sh; i >= 0; i--) {
    if (d[SInt::L] == 0 && mcmp(rem, d) >= 0) { msub_mag(rem, rem, d); q.m[i >> 6] |= 1ULL << (i & 63); }
    for (int k = 0; k < SInt::L; k++) d[k] = (d[k] >> 1) | (d[k + 1] << 63);   // d >>= 1
    d[SInt::L] >>= 1;
  }
  q.neg = q.is_zero() ? false : (a.neg != b.neg);
  return q;
}

// Joint sparse form (Solinas, "Low-weight binary representations for pairs of integers") of two magnitudes < 2^129:
// digits (u0_i, u1_i) in {-1,0,1}^2 with sum_i u_i 2^i = k and at most half of the rows non-zero on average.
// Output: 4 bits per row, [1:0] = code(u0), [3:2] = code(u1), code(0)=0, code(+1)=1, code(-1)=3; 130 rows in 17 words.
// Consumed by foldcore.hip.h (device) as the add schedule of b'*GL + a'*GR.
inline void jsf_recode(const uint64_t k0_[3], const uint64_t k1_[3], uint32_t out[17]) {
  uint64_t k0[3] = {k0_[0], k0_[1], k0_[2]}, k1[3] = {k1_[0], k1_[1], k1_[2]};
  memset(out, 0, 17 * sizeof(uint32_t));
  unsigned d0 = 0, d1 = 0;
  auto nz = [](const uint64_t *k) { return (k[0] | k[1] | k[2]) != 0; };
  auto shr1 = [](uint64_t *k) { k[0] = (k[0] >> 1) | (k[1] << 63); k[1] = (k[1] >> 1) | (k[2] << 63); k[2] >>= 1; };
  for (int i = 0; i < 130 && (nz(k0) || nz(k1) || d0 || d1); i++) {
    unsigned l0 = (unsigned)(k0[0] & 7) + d0, l1 = (unsigned)(k1[0] & 7) + d1;     // low bits of k + d (mod 8 is all we need)
    int u0 = 0, u1 = 0;
    if (l0 & 1) { u0 = 2 - (int)(l0 & 3); if (((l0 & 7) == 3 || (l0 & 7) == 5) && (l1 & 3) == 2) u0 = -u0; }
    if (l1 & 1) { u1 = 2 - (int)(l1 & 3); if (((l1 & 7) == 3 || (l1 & 7) == 5) && (l0 & 3) == 2) u1 = -u1; }
    if (2 * (int)d0 == 1 + u0) d0 = 1 - d0;
    if (2 * (int)d1 == 1 + u1) d1 = 1 - d1;
    shr1(k0); shr1(k1);
    unsigned code = (u0 == 0 ? 0u : u0 > 0 ? 1u : 3u) | ((u1 == 0 ? 0u : u1 > 0 ? 1u : 3u) << 2);
    out[i >> 3] |= code << (4 * (i & 7));
  }
}

// reduceScalar (Commitment.hs:276-279): signed representative in (-n/2, n/2]
inline SInt reduce_scalar(const U256 &x) {
  SInt r = SInt::zero();
  U256 neg; sub_raw(neg, FR().m, x);
  if (cmp(x, neg) > 0) { memcpy(r.m, neg.w, 32); r.neg = true; }
  else { memcpy(r.m, x.w, 32); }
  if (r.is_zero()) r.neg = false;
  return r;
}
// extractScalar = fromInteger (Commitment.hs:274)
inline U256 extract_scalar(const SInt &s) {
  U256 v; memcpy(v.w, s.m, 32);   // |s| < 2^130 in every use
  return s.neg ? mneg(v, FR()) : v;
}
// rationalReduceScalar (Commitment.hs:242-255): the egcd list starts at its second argument
// (:252); the first (r, s) with r^2 <= 2n is returned (:247).
// The plain multi-limb restatement; rational_reduce_scalar below takes the same steps with word-sized quotient estimates.
inline std::pair<SInt, SInt> rational_reduce_scalar_plain(const U256 &x) {
  SInt pr = SInt::zero(), ps = SInt::zero(), cr = reduce_scalar(x), cs = SInt::zero();
  memcpy(pr.m, FR().m.w, 32);
  cs.m[0] = 1;
  SInt two_n = SInt::zero(); memcpy(two_n.m, FR().m.w, 32); madd_mag(two_n.m, two_n.m, two_n.m);
  for (;;) {
    bool big = cr.bits() > 130;
    if (!big) { SInt a = cr; a.neg = false; SInt sq = smul(a, a); big = mcmp(sq.m, two_n.m) > 0; }
    if (!big) break;
    SInt q = squot(pr, cr);
    SInt nr = sadd(pr, sneg(smul(q, cr)));
    SInt ns = sadd(ps, sneg(smul(q, cs)));
    pr = cr; ps = cs; cr = nr; cs = ns;
  }
  return {cr, cs};
}

// The same (r, s), step for step, on magnitudes.  With r_0 = n, r_1 = x' = reduceScalar x, s_0 = 0, s_1 = 1 and truncating
// quotients q_i = quot r_(i-1) r_i:  |r_(i+1)| = |r_(i-1)| mod |r_i|,  |s_(i+1)| = |s_(i-1)| + |q_i| |s_i|;  r_i is positive for
// even i and has the sign of x' for odd i;  every q_i has the sign of x', so s_i alternates (+ for odd i, - for even i) when
// x' > 0 and stays positive when x' < 0.  A quotient is read off the leading 64 bits of both remainders when the two bounds
// h0 / (h1 + 1) <= q <= (h0 + 1) / h1 agree (Euclid's quotients are a few bits long, so they nearly always do); otherwise that
// one step goes through the shift-subtract division.  A batch prover calls this twice per proof and round on the host.
inline std::pair<SInt, SInt> rational_reduce_scalar(const U256 &x) {
  const SInt x1 = reduce_scalar(x);
  uint64_t R0[SInt::L], R1[SInt::L], S0[SInt::L] = {0}, S1[SInt::L] = {1, 0, 0, 0, 0};
  memset(R0, 0, sizeof R0); memcpy(R0, FR().m.w, 32);
  memcpy(R1, x1.m, sizeof R1);
  uint64_t two_n[SInt::L] = {0}; memcpy(two_n, FR().m.w, 32); madd_mag(two_n, two_n, two_n);
  auto bits = [](const uint64_t *m) { for (int i = SInt::L - 1; i >= 0; i--) if (m[i]) return 64 * i + 64 - __builtin_clzll(m[i]); return 0; };
  auto top64 = [](const uint64_t *m, int k) {          // bits k .. k+63
    const int w = k >> 6, o = k & 63;
    uint64_t v = m[w] >> o;
    if (o && w + 1 < SInt::L) v |= m[w + 1] << (64 - o);
    return v;
  };
  int idx = 1;                                        // (R1, S1) = (|r_idx|, |s_idx|)
  for (;;) {
    const int b1 = bits(R1);
    bool big = b1 > 130;
    if (!big) { SInt a = SInt::zero(); memcpy(a.m, R1, sizeof R1); SInt sq = smul(a, a); big = mcmp(sq.m, two_n) > 0; }
    if (!big) break;
    const int b0 = bits(R0), k = b0 > 64 ? b0 - 64 : 0;
    const uint64_t h0 = top64(R0, k), h1 = top64(R1, k);
    uint64_t R2[SInt::L], S2[SInt::L];
    bool done = false;
    if (h1) {
      const u128 ql = (u128)h0 / ((u128)h1 + 1), qh = ((u128)h0 + 1) / h1;
      if (ql == qh && (ql >> 62) == 0) {
        const uint64_t q = (uint64_t)ql;
        u128 c = 0; uint64_t br = 0;
        for (int i = 0; i < SInt::L; i++) {             // R2 = R0 - q R1
          c += (u128)q * R1[i];
          const u128 d = (u128)R0[i] - (uint64_t)c - br;
          R2[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; c >>= 64;
        }
        c = 0;
        for (int i = 0; i < SInt::L; i++) { c += (u128)q * S1[i] + S0[i]; S2[i] = (uint64_t)c; c >>= 64; }   // S2 = S0 + q S1
        done = true;
      }
    }
    if (!done) {
      SInt a = SInt::zero(), b = SInt::zero(), sa = SInt::zero(), sb = SInt::zero();
      memcpy(a.m, R0, sizeof R0); memcpy(b.m, R1, sizeof R1); memcpy(sa.m, S0, sizeof S0); memcpy(sb.m, S1, sizeof S1);
      const SInt q = squot(a, b);
      const SInt nr = sadd(a, sneg(smul(q, b))), ns = sadd(sa, smul(q, sb));
      memcpy(R2, nr.m, sizeof R2); memcpy(S2, ns.m, sizeof S2);
    }
    memcpy(R0, R1, sizeof R0); memcpy(R1, R2, sizeof R1); memcpy(S0, S1, sizeof S0); memcpy(S1, S2, sizeof S1);
    idx++;
  }
  SInt r = SInt::zero(), sc = SInt::zero();
  memcpy(r.m, R1, sizeof R1); memcpy(sc.m, S1, sizeof S1);
  r.neg = (idx & 1) ? x1.neg : false;
  sc.neg = x1.neg ? false : !(idx & 1);
  if (r.is_zero()) r.neg = false;
  if (sc.is_zero()) sc.neg = false;
  return {r, sc};
}

}  // namespace bppp_host
