// hosteis.hpp — Eisenstein-integer side of the reference's "fast" configuration, on the host (once per round per sub-argument).
//
// SplitScalar (FastPrime p) (src/Commitment.hs:293-306): ReducedScalar = Eis Integer, reducedChar = conjEis . charEis,
// normScalar = normEis, reduceScalar = decomposeEis, rationalReducedScalarLength = 65; rationalReduceScalar itself is the class
// default (src/Commitment.hs:242-255) running over the Integral (Eis a) instance, whose quotRem rounds to the NEAREST Eisenstein
// integer (src/Data/Field/Eis.hs:72-82).  decomposeEis for secp256k1's scalar field is decomposeFastPrimeEis
// (src/Data/Field/Galois/FastPrime.hs:186-205) with charEis from src/Data/Curve/Weierstrass/FastSECP256K1.hs:56.
// Integers here are sign-magnitude with 512 bits of magnitude (the largest intermediate, x * conj m, stays under 2^390).
#pragma once
#include <stdint.h>
#include <string.h>
#include <utility>
#include "hostmath.hpp"

namespace bppp_eis {
using bppp_host::U256;
using bppp_host::u128;

struct Big {
  static const int L = 8;
  uint64_t m[L]; bool neg;
  static Big zero() { Big r; memset(r.m, 0, sizeof r.m); r.neg = false; return r; }
  static Big from_u64(uint64_t v) { Big r = zero(); r.m[0] = v; return r; }
  static Big from_u256(const U256 &v) { Big r = zero(); memcpy(r.m, v.w, 32); return r; }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < L; i++) o |= m[i]; return o == 0; }
  int bits() const { for (int i = L - 1; i >= 0; i--) if (m[i]) return 64 * i + 64 - __builtin_clzll(m[i]); return 0; }
};
inline int magcmp(const Big &a, const Big &b) { for (int i = Big::L - 1; i >= 0; i--) if (a.m[i] != b.m[i]) return a.m[i] < b.m[i] ? -1 : 1; return 0; }
inline Big magadd(const Big &a, const Big &b) { Big r; u128 c = 0; for (int i = 0; i < Big::L; i++) { c += (u128)a.m[i] + b.m[i]; r.m[i] = (uint64_t)c; c >>= 64; } r.neg = false; return r; }
inline Big magsub(const Big &a, const Big &b) {   // |a| >= |b|
  Big r; uint64_t br = 0;
  for (int i = 0; i < Big::L; i++) { u128 d = (u128)a.m[i] - b.m[i] - br; r.m[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  r.neg = false; return r;
}
inline Big neg(Big a) { if (!a.is_zero()) a.neg = !a.neg; return a; }
inline Big add(const Big &a, const Big &b) {
  Big r;
  if (a.neg == b.neg) { r = magadd(a, b); r.neg = a.neg; }
  else if (magcmp(a, b) >= 0) { r = magsub(a, b); r.neg = a.neg; }
  else { r = magsub(b, a); r.neg = b.neg; }
  if (r.is_zero()) r.neg = false;
  return r;
}
inline Big sub(const Big &a, const Big &b) { return add(a, neg(b)); }
inline Big mul(const Big &a, const Big &b) {
  Big r = Big::zero();
  for (int i = 0; i < Big::L; i++) {
    u128 c = 0;
    for (int j = 0; i + j < Big::L; j++) { c += (u128)a.m[i] * b.m[j] + r.m[i + j]; r.m[i + j] = (uint64_t)c; c >>= 64; }
  }
  r.neg = r.is_zero() ? false : (a.neg != b.neg);
  return r;
}
inline int cmp(const Big &a, const Big &b) {      // signed comparison
  if (a.neg != b.neg) return a.neg ? -1 : 1;
  int c = magcmp(a, b);
  return a.neg ? -c : c;
}
inline Big absb(Big a) { a.neg = false; return a; }
// floor division by a POSITIVE divisor (Haskell divMod): n = q d + r, 0 <= r < d
inline void divmod_floor(const Big &n, const Big &d, Big &q, Big &r) {
  q = Big::zero(); r = Big::zero();
  const int nb = n.bits();
  for (int i = nb - 1; i >= 0; i--) {             // schoolbook shift-subtract on the magnitude
    for (int k = Big::L - 1; k > 0; k--) r.m[k] = (r.m[k] << 1) | (r.m[k - 1] >> 63);
    r.m[0] = (r.m[0] << 1) | ((n.m[i >> 6] >> (i & 63)) & 1);
    if (magcmp(r, d) >= 0) { r = magsub(r, d); q.m[i >> 6] |= 1ULL << (i & 63); }
  }
  if (n.neg && !r.is_zero()) { q = magadd(q, Big::from_u64(1)); r = magsub(d, r); }   // towards minus infinity
  q.neg = n.neg && !q.is_zero();
  r.neg = false;
}

struct Eis { Big a, b; };                                                         // a + b * unity3
inline Eis econj(const Eis &x) { return {sub(x.a, x.b), neg(x.b)}; }              // conjEis (Eis.hs:20-21)
inline Big enorm(const Eis &x) { return add(sub(mul(x.a, x.a), mul(x.a, x.b)), mul(x.b, x.b)); }   // normEis (:23-24)
inline Eis eadd(const Eis &x, const Eis &y) { return {add(x.a, y.a), add(x.b, y.b)}; }
inline Eis esub(const Eis &x, const Eis &y) { return {sub(x.a, y.a), sub(x.b, y.b)}; }
inline Eis emul(const Eis &x, const Eis &y) {                                    // (*) (Eis.hs:30-34)
  Big a1 = mul(x.a, y.a), b1 = mul(x.b, y.b), c1 = mul(sub(x.a, x.b), sub(y.a, y.b));
  return {sub(a1, b1), sub(a1, c1)};
}
// quot of the Integral (Eis a) instance (Eis.hs:72-82): component-wise division of x * conj m by norm m, rounded to nearest
inline Eis equot(const Eis &x, const Eis &m) {
  const Big mN = enorm(m);
  const Eis uv = emul(x, econj(m));
  auto rnd = [&](const Big &n) {
    Big q, r;
    divmod_floor(n, mN, q, r);
    // if m - |r| < |r| then q + signum r else q      (r >= 0 here, as divMod's remainder with a positive modulus)
    if (magcmp(magsub(mN, r), r) < 0) q = add(q, Big::from_u64(1));
    return q;
  };
  return {rnd(uv.a), rnd(uv.b)};
}

inline Big big_from_dec_pair(uint64_t hi, uint64_t lo) { Big r = Big::zero(); r.m[0] = lo; r.m[1] = hi; return r; }
// charEis of the scalar field (FastSECP256K1.hs:56): (303414439467246543595250775667605759171, -64502973549206556628585045361533709077)
inline Eis char_eis_fr() {
  Eis c;
  c.a = big_from_dec_pair(0xE4437ED6010E8828ULL, 0x6F547FA90ABFE4C3ULL);
  c.b = big_from_dec_pair(0x3086D221A7D46BCDULL, 0xE86C90E49284EB15ULL); c.b.neg = true;
  return c;
}
inline Big order_big() { return Big::from_u256(bppp_host::FR().m); }

// decomposeFastPrimeEis (FastPrime.hs:186-205): x = a + b unity3 (mod n) with the reference's one-step rounding
inline Eis decompose_eis(const U256 &x) {
  const Big N = order_big();
  const Eis pFac = econj(char_eis_fr());
  const Eis xInt = {Big::from_u256(x), Big::zero()};
  const Eis uv = emul(xInt, econj(pFac));
  auto shr256_floor = [](const Big &n) {                   // unsafeShiftR on a (possibly negative) Integer: floor (n / 2^256)
    Big q = Big::zero();
    for (int i = 0; i + 4 < Big::L; i++) q.m[i] = n.m[i + 4];
    if (n.neg) {
      bool low = (n.m[0] | n.m[1] | n.m[2] | n.m[3]) != 0;
      if (low) q = magadd(q, Big::from_u64(1));
      q.neg = !q.is_zero();
    }
    return q;
  };
  auto rnd = [&](const Big &n, Big q) {
    const Big r = sub(n, mul(N, q));
    if (magcmp(absb(r), absb(add(r, N))) > 0) return sub(q, Big::from_u64(1));
    if (magcmp(absb(r), absb(sub(r, N))) > 0) return add(q, Big::from_u64(1));
    return q;
  };
  const Eis q = {rnd(uv.a, shr256_floor(uv.a)), rnd(uv.b, shr256_floor(uv.b))};
  return esub(xInt, emul(q, pFac));
}

// rationalReduceScalar (Commitment.hs:242-255) over the FastPrime instance (:293-306): the egcd list starts at its SECOND argument
// (reduceScalar x, 1); the first (r, s) with (normEis r)^2 <= 2 n is returned.  r = s * x in Z[unity3] / (conj charEis).
inline std::pair<Eis, Eis> rational_reduce_eis(const U256 &x) {
  // floor (sqrt (2 n)) = 0x1_6A09E667F3BCC908_B2FB1366EA957D3D ... computed once: (normEis r)^2 > 2n  <=>  normEis r > isqrt (2n)
  static const Big ROOT = [] {
    Big two_n = magadd(order_big(), order_big());
    Big lo = Big::zero(), hi = Big::zero(); hi.m[2] = 4;     // 2^130 > sqrt(2n)
    while (magcmp(magadd(lo, Big::from_u64(1)), hi) < 0) {
      Big mid = magadd(lo, hi);
      for (int k = 0; k < Big::L - 1; k++) mid.m[k] = (mid.m[k] >> 1) | (mid.m[k + 1] << 63);
      mid.m[Big::L - 1] >>= 1;
      if (magcmp(mul(mid, mid), two_n) <= 0) lo = mid; else hi = mid;
    }
    return lo;
  }();
  Eis pr = econj(char_eis_fr()), ps = {Big::zero(), Big::zero()};
  Eis cr = decompose_eis(x), cs = {Big::from_u64(1), Big::zero()};
  for (int guard = 0; guard < 600; guard++) {
    if (magcmp(enorm(cr), ROOT) <= 0) break;       // normEis >= 0
    const Eis q = equot(pr, cr);
    const Eis nr = esub(pr, emul(q, cr)), ns = esub(ps, emul(q, cs));
    pr = cr; ps = cs; cr = nr; cs = ns;
  }
  return {cr, cs};
}

}  // namespace bppp_eis
