// modinv.hip.h — modular inversion by Bernstein-Yang "safegcd" division steps, 30 bits at a time.
//
// The reference inverts through GMP (`invField#`, src/Data/Field/Galois/FastPrime/Internal.hs:981-983; `recip` of Prime p);
// only the value matters: x^-1 mod m in [0, m), and 0 -> 0 as batchInverse defines it (src/Data/Field/BatchInverse.hs:18).
// Fermat's a^(m-2) costs 255 squarings + ~15 (Fq chain) or ~128 (Fr) multiplications, i.e. 270-380 field multiplications of a
// dependent chain.  Division steps need 20 rounds of [30 branch-free steps on 32-bit words + two 9-limb updates with
// 32x32->64 multiply-adds] ~ 14 k instructions ~ 60 field multiplications, identical for every input (no divergence), and the same
// code serves both moduli.  The construction (signed 30-bit limbs, 2x2 transition matrices scaled by 2^30, the zeta = -delta - 1/2
// bookkeeping, 600 steps for 256-bit inputs) is the published one used by libsecp256k1's modinv32.
#pragma once
#include "fe.hip.h"

namespace bppp {

struct s30 { int32_t v[9]; };
static constexpr int32_t MI_M30 = 0x3FFFFFFF;

// MOD = 0: p = 2^256 - 2^32 - 977;  MOD = 1: n (group order).  Plain 30-bit limbs and m^-1 mod 2^30.
template <int MOD> BPPP_DI int32_t mi_mod_limb(int i) {
  if (MOD == 0) return i == 0 ? 0x3FFFFC2F : i == 1 ? 0x3FFFFFFB : i == 8 ? 0xFFFF : 0x3FFFFFFF;
  return i == 0 ? 0x10364141 : i == 1 ? 0x3F497A33 : i == 2 ? 0x348A03BB : i == 3 ? 0x2BB739AB : i == 4 ? 0x3FFFFEBA : i == 8 ? 0xFFFF : 0x3FFFFFFF;
}
template <int MOD> BPPP_DI uint32_t mi_mod_inv30() { return MOD == 0 ? 0x2DDACACFu : 0x2A774EC1u; }

// 30 division steps on the low words; returns the new zeta and the transition matrix (scaled by 2^30)
BPPP_DI int32_t mi_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    uint32_t m1 = (uint32_t)(zeta >> 31);          // zeta < 0
    uint32_t m2 = 0u - (g & 1u);                   // g odd
    uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
    g += x & m2; q += y & m2; r += z & m2;
    m1 &= m2;
    zeta = (int32_t)(((uint32_t)zeta ^ m1) - 1u);
    f += g & m1; u += q & m1; v += r & m1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
  return zeta;
}

// (d, e) <- t * (d, e) / 2^30  (mod m)
template <int MOD> BPPP_DI void mi_update_de(s30 &d, s30 &e, int32_t u, int32_t v, int32_t q, int32_t r) {
  int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int32_t di = d.v[0], ei = e.v[0];
  int64_t cd = (int64_t)u * di + (int64_t)v * ei;
  int64_t ce = (int64_t)q * di + (int64_t)r * ei;
  md -= (int32_t)((mi_mod_inv30<MOD>() * (uint32_t)cd + (uint32_t)md) & (uint32_t)MI_M30);
  me -= (int32_t)((mi_mod_inv30<MOD>() * (uint32_t)ce + (uint32_t)me) & (uint32_t)MI_M30);
  cd += (int64_t)mi_mod_limb<MOD>(0) * md;
  ce += (int64_t)mi_mod_limb<MOD>(0) * me;
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    di = d.v[i]; ei = e.v[i];
    cd += (int64_t)u * di + (int64_t)v * ei;
    ce += (int64_t)q * di + (int64_t)r * ei;
    cd += (int64_t)mi_mod_limb<MOD>(i) * md;
    ce += (int64_t)mi_mod_limb<MOD>(i) * me;
    d.v[i - 1] = (int32_t)cd & MI_M30; cd >>= 30;
    e.v[i - 1] = (int32_t)ce & MI_M30; ce >>= 30;
  }
  d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// (f, g) <- t * (f, g) / 2^30  (exact)
BPPP_DI void mi_update_fg(s30 &f, s30 &g, int32_t u, int32_t v, int32_t q, int32_t r) {
  int32_t fi = f.v[0], gi = g.v[0];
  int64_t cf = (int64_t)u * fi + (int64_t)v * gi;
  int64_t cg = (int64_t)q * fi + (int64_t)r * gi;
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    fi = f.v[i]; gi = g.v[i];
    cf += (int64_t)u * fi + (int64_t)v * gi;
    cg += (int64_t)q * fi + (int64_t)r * gi;
    f.v[i - 1] = (int32_t)cf & MI_M30; cf >>= 30;
    g.v[i - 1] = (int32_t)cg & MI_M30; cg >>= 30;
  }
  f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
// bring d (in (-2m, m)) into [0, m), negated first when sign < 0
template <int MOD> BPPP_DI void mi_normalize(s30 &r, int32_t sign) {
  int32_t add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += mi_mod_limb<MOD>(i) & add;
  int32_t neg = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ neg) - neg;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= MI_M30; }
  add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] += mi_mod_limb<MOD>(i) & add;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= MI_M30; }
}

BPPP_DI s30 s30_from_fe(const fe &a) {
  s30 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 30 * i, w = bit >> 5, o = bit & 31;
    uint32_t lo = a.v[w] >> o;
    if (o > 2 && w + 1 < 8) lo |= a.v[w + 1] << (32 - o);
    r.v[i] = (int32_t)(lo & (uint32_t)MI_M30);
  }
  return r;
}
BPPP_DI fe fe_from_s30(const s30 &a) {           // limbs in [0, 2^30), value < 2^256
  fe r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int bit = 32 * w, i = bit / 30, o = bit % 30;
    uint32_t lo = (uint32_t)a.v[i] >> o;
    if (i + 1 < 9) lo |= (uint32_t)a.v[i + 1] << (30 - o);
    if (o > 28 && i + 2 < 9) lo |= (uint32_t)a.v[i + 2] << (60 - o);
    r.v[w] = lo;
  }
  return r;
}

// x^-1 mod m for canonical x; 0 -> 0.  Same instruction stream for every input.
template <int MOD> BPPP_DI fe fe_modinv(const fe &x) {
  s30 d, e, f, g = s30_from_fe(x);
#pragma unroll
  for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = i == 0; f.v[i] = mi_mod_limb<MOD>(i); }
  int32_t zeta = -1;
  for (int it = 0; it < 20; it++) {
    int32_t u, v, q, r;
    zeta = mi_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], u, v, q, r);
    mi_update_de<MOD>(d, e, u, v, q, r);
    mi_update_fg(f, g, u, v, q, r);
  }
  mi_normalize<MOD>(d, f.v[8]);
  return fe_from_s30(d);
}

}  // namespace bppp
