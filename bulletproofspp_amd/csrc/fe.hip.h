// fe.hip.h — 256-bit modular arithmetic for gfx950 (8 x 32-bit limbs in VGPRs).
//
// Device counterpart of the reference's field layer: `Prime p` from galois-field (default path,
// app/Main.hs:17) / FastPrime primops addField# negField# mulField# sqrField# invField#
// (src/Data/Field/Galois/FastPrime/Internal.hs:909-988).  Results are always the canonical
// integer in [0, m), which is the only thing the reference's semantics fix.
//
// Two moduli: Fq (coordinates, p = 2^256 - 2^32 - 977) and Fr (scalars, n = group order).
// Both have the shape 2^256 - r, so a 512-bit product is reduced by folding hi*r into lo
// (the same idea as Internal.hs:943-956, which folds three times with 64-bit limbs).
//
// No MFMA: this is carry-chained integer arithmetic on the VALU (v_mad_u64_u32 / v_add_co).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BPPP_DI __device__ __forceinline__

namespace bppp {

struct fe { uint32_t v[8]; };

// ---- constants (little-endian 32-bit limbs)
// p = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE FFFFFC2F
__device__ __constant__ static const uint32_t FP_M[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                                         0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
// n = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE BAAEDCE6 AF48A03B BFD25E8C D0364141
__device__ __constant__ static const uint32_t FR_M[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                                         0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
// 2^256 - n = 1 4551231950b75fc4 402da1732fc9bebf (Internal.hs:48-51), 5 limbs
__device__ __constant__ static const uint32_t FR_R[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u};

BPPP_DI fe fe_zero() { fe r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
BPPP_DI fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }
BPPP_DI bool fe_is_zero(const fe &a) {
  uint32_t o = 0;
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
BPPP_DI bool fe_eq(const fe &a, const fe &b) {
  uint32_t o = 0;
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a + b, returns carry
BPPP_DI uint32_t raw_add(fe &r, const fe &a, const fe &b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
// r = a - b, returns borrow (1 if a < b)
BPPP_DI uint32_t raw_sub(fe &r, const fe &a, const fe &b) {
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (int64_t)a.v[i] - (int64_t)b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)(c & 1);
}

template <int MOD> BPPP_DI fe mod_m() {
  fe m;
#pragma unroll
  for (int i = 0; i < 8; i++) m.v[i] = MOD ? FR_M[i] : FP_M[i];
  return m;
}

// literal forms (avoid constant-memory loads on the hot Fq path)
BPPP_DI fe fp_modulus() {
  fe m; m.v[0] = 0xFFFFFC2Fu; m.v[1] = 0xFFFFFFFEu;
  for (int i = 2; i < 8; i++) m.v[i] = 0xFFFFFFFFu;
  return m;
}
BPPP_DI fe fr_modulus() {
  fe m; m.v[0] = 0xD0364141u; m.v[1] = 0xBFD25E8Cu; m.v[2] = 0xAF48A03Bu; m.v[3] = 0xBAAEDCE6u;
  m.v[4] = 0xFFFFFFFEu; m.v[5] = 0xFFFFFFFFu; m.v[6] = 0xFFFFFFFFu; m.v[7] = 0xFFFFFFFFu;
  return m;
}
template <int MOD> BPPP_DI fe modulus() { return MOD ? fr_modulus() : fp_modulus(); }

// canonical (a + b) mod m for canonical inputs — addField# (Internal.hs:909-924)
template <int MOD> BPPP_DI fe fe_add(const fe &a, const fe &b) {
  fe s, t;
  uint32_t c = raw_add(s, a, b);
  uint32_t br = raw_sub(t, s, modulus<MOD>());
  bool use_t = c | (br ^ 1u);   // overflowed 2^256, or s >= m
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = use_t ? t.v[i] : s.v[i];
  return r;
}
template <int MOD> BPPP_DI fe fe_sub(const fe &a, const fe &b) {
  fe d, t;
  uint32_t br = raw_sub(d, a, b);
  raw_add(t, d, modulus<MOD>());
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? t.v[i] : d.v[i];
  return r;
}
// negField# (Internal.hs:927-932)
template <int MOD> BPPP_DI fe fe_neg(const fe &a) {
  fe t;
  raw_sub(t, modulus<MOD>(), a);
  bool z = fe_is_zero(a);
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = z ? 0u : t.v[i];
  return r;
}
template <int MOD> BPPP_DI fe fe_dbl(const fe &a) { return fe_add<MOD>(a, a); }

// 8x8 schoolbook product, operand scanning; each step is one v_mad_u64_u32 plus carry adds.
BPPP_DI void mul_wide(uint32_t t[16], const fe &a, const fe &b) {
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[0] * b.v[j]; t[j] = (uint32_t)c; c >>= 32; }
  t[8] = (uint32_t)c;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c += (uint64_t)a.v[i] * b.v[j] + t[i + j];
      t[i + j] = (uint32_t)c; c >>= 32;
    }
    t[i + 8] = (uint32_t)c;
  }
}

// 36-product squaring (sqr256With256#, Internal.hs:580-681): off-diagonal terms doubled.
BPPP_DI void sqr_wide(uint32_t t[16], const fe &a) {
  // off-diagonal products a[i]*a[j], i<j, accumulated at t[i+j]
#pragma unroll
  for (int k = 0; k < 16; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      c += (uint64_t)a.v[i] * a.v[j] + t[i + j];
      t[i + j] = (uint32_t)c; c >>= 32;
    }
    t[i + 8] = (uint32_t)c;
  }
  // double
  uint32_t top = 0;
#pragma unroll
  for (int k = 1; k < 16; k++) { uint32_t nt = t[k] >> 31; t[k] = (t[k] << 1) | top; top = nt; }
  // add diagonals
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)a.v[i] * a.v[i];
    c += (uint64_t)t[2 * i] + (uint32_t)d; t[2 * i] = (uint32_t)c; c >>= 32;
    c += (uint64_t)t[2 * i + 1] + (uint32_t)(d >> 32); t[2 * i + 1] = (uint32_t)c; c >>= 32;
  }
}

// Reduce a 512-bit value mod p = 2^256 - (2^32 + 977): t = lo + hi*977 + (hi << 32), twice.
BPPP_DI fe fp_reduce(const uint32_t t[16]) {
  uint32_t acc[10];
  uint64_t c = 0, m = 0;
  // acc = lo + hi*977 + (hi << 32); m carries hi*977, c carries the column sum
#pragma unroll
  for (int k = 0; k < 8; k++) {
    m += (uint64_t)t[8 + k] * 977u;
    c += (uint64_t)t[k] + (uint32_t)m + (k ? t[8 + k - 1] : 0u);
    m >>= 32;
    acc[k] = (uint32_t)c; c >>= 32;
  }
  c += m + t[15];
  acc[8] = (uint32_t)c; acc[9] = (uint32_t)(c >> 32);
  // second fold: e = acc[8..9] (< 2^34) times (2^32 + 977)
  uint64_t e = ((uint64_t)acc[9] << 32) | acc[8];
  uint64_t e977 = e * 977u;                 // < 2^44
  fe r;
  c = (uint64_t)acc[0] + (uint32_t)e977;
  r.v[0] = (uint32_t)c; c >>= 32;
  c += (uint64_t)acc[1] + (uint32_t)(e977 >> 32) + (uint32_t)e;
  r.v[1] = (uint32_t)c; c >>= 32;
  c += (uint64_t)acc[2] + (uint32_t)(e >> 32);
  r.v[2] = (uint32_t)c; c >>= 32;
#pragma unroll
  for (int k = 3; k < 8; k++) { c += acc[k]; r.v[k] = (uint32_t)c; c >>= 32; }
  // a final carry out of 2^256 (c is 0 or 1) folds once more; cannot carry again
  uint64_t f = c * 0x1000003D1ull;
  c = (uint64_t)r.v[0] + (uint32_t)f; r.v[0] = (uint32_t)c; c >>= 32;
  c += (uint64_t)r.v[1] + (uint32_t)(f >> 32); r.v[1] = (uint32_t)c; c >>= 32;
#pragma unroll
  for (int k = 2; k < 8; k++) { c += r.v[k]; r.v[k] = (uint32_t)c; c >>= 32; }
  // canonicalise: r < 2^256 < 2p, so one conditional subtraction
  fe s;
  uint32_t br = raw_sub(s, r, fp_modulus());
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : s.v[i];
  return r;
}

// Reduce a 512-bit value mod n = 2^256 - R (R is 129 bits): fold hi*R into lo until hi vanishes.
BPPP_DI fe fr_reduce(const uint32_t t[16]) {
  // fold 1: 8-limb hi x 5-limb R -> 13 limbs, plus lo
  uint32_t a[14];
#pragma unroll
  for (int k = 0; k < 14; k++) a[k] = k < 8 ? t[k] : 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { c += (uint64_t)t[8 + i] * FR_R[j] + a[i + j]; a[i + j] = (uint32_t)c; c >>= 32; }
#pragma unroll
    for (int k = i + 5; k < 14; k++) { c += a[k]; a[k] = (uint32_t)c; c >>= 32; }
  }
  // fold 2: hi = a[8..13] (< 2^130+) x R -> < 2^260, plus lo
  uint32_t b[12];
#pragma unroll
  for (int k = 0; k < 12; k++) b[k] = k < 8 ? a[k] : 0u;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { c += (uint64_t)a[8 + i] * FR_R[j] + b[i + j]; b[i + j] = (uint32_t)c; c >>= 32; }
#pragma unroll
    for (int k = i + 5; k < 12; k++) { c += b[k]; b[k] = (uint32_t)c; c >>= 32; }
  }
  // fold 3: hi = b[8] (a few bits; b[9..11] are zero) x R
  fe r;
  {
    uint64_t c = 0;
    uint32_t h = b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      c += (uint64_t)b[k] + (k < 5 ? (uint64_t)h * FR_R[k] : 0ull);
      r.v[k] = (uint32_t)c; c >>= 32;
    }
    // possible carry out: fold R once more (value is then small, cannot carry again)
    uint32_t h2 = (uint32_t)c;
    c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      c += (uint64_t)r.v[k] + (k < 5 ? (uint64_t)h2 * FR_R[k] : 0ull);
      r.v[k] = (uint32_t)c; c >>= 32;
    }
  }
  fe s;
  uint32_t br = raw_sub(s, r, fr_modulus());
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : s.v[i];
  return r;
}

// mulField# (Internal.hs:943-956) / sqrField# (:960-973)
template <int MOD> BPPP_DI fe fe_mul(const fe &a, const fe &b) {
  uint32_t t[16];
  mul_wide(t, a, b);
  return MOD ? fr_reduce(t) : fp_reduce(t);
}
template <int MOD> BPPP_DI fe fe_sqr(const fe &a) {
  uint32_t t[16];
  sqr_wide(t, a);
  return MOD ? fr_reduce(t) : fp_reduce(t);
}

// invField# (Internal.hs:981-983; GMP there): Fermat a^(m-2), 0 -> 0 like batchInverse.
template <int MOD> __device__ __noinline__ fe fe_inv(const fe &a) {
  fe e = modulus<MOD>();
  e.v[0] -= 2;  // low limb of both moduli is > 2
  fe acc = fe_one(), base = a;
  for (int i = 0; i < 256; i++) {
    if ((e.v[i >> 5] >> (i & 31)) & 1) acc = fe_mul<MOD>(acc, base);
    base = fe_sqr<MOD>(base);
  }
  return acc;
}

// The same inverse by the binary extended Euclid (right-shift) algorithm: ~40 instructions per step over at most ~510 steps,
// against 256 squarings + ~128 multiplications of ~450 instructions each — about 8x fewer instructions.  Variable time and
// divergent across lanes, so it is for the places where ONE lane of a wavefront inverts (the per-proof batched inversion of
// trrp.hip); public data only.  0 -> 0.
template <int MOD> BPPP_DI fe fe_inv_vartime(const fe &a) {
  if (fe_is_zero(a)) return fe_zero();
  const fe m = modulus<MOD>();
  fe u = a, v = m, x1 = fe_one(), x2 = fe_zero();
  auto is_one = [](const fe &f) { uint32_t o = f.v[0] ^ 1u; for (int i = 1; i < 8; i++) o |= f.v[i]; return o == 0; };
  auto shr1 = [](fe &f, uint32_t top) { for (int i = 0; i < 7; i++) f.v[i] = (f.v[i] >> 1) | (f.v[i + 1] << 31); f.v[7] = (f.v[7] >> 1) | (top << 31); };
  auto halve = [&](fe &f) {                       // f/2 mod m for f in [0, m)
    uint32_t top = 0;
    if (f.v[0] & 1u) { fe s; top = raw_add(s, f, m); f = s; }
    shr1(f, top);
  };
  auto geq = [](const fe &p, const fe &q) { for (int i = 7; i >= 0; i--) if (p.v[i] != q.v[i]) return p.v[i] > q.v[i]; return true; };
  while (!is_one(u) && !is_one(v)) {
    while (!(u.v[0] & 1u)) { shr1(u, 0); halve(x1); }
    while (!(v.v[0] & 1u)) { shr1(v, 0); halve(x2); }
    if (geq(u, v)) { fe d; raw_sub(d, u, v); u = d; x1 = fe_sub<MOD>(x1, x2); }
    else { fe d; raw_sub(d, v, u); v = d; x2 = fe_sub<MOD>(x2, x1); }
  }
  return is_one(u) ? x1 : x2;
}

// ---- 16-byte vector loads/stores of field elements (2 x dwordx4 per element)
BPPP_DI fe fe_load(const uint32_t *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 lo = q[0], hi = q[1];
  fe r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
BPPP_DI void fe_store(uint32_t *p, const fe &a) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

}  // namespace bppp
