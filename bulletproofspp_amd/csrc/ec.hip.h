// ec.hip.h — secp256k1 group law on the device: affine inputs, XYZZ accumulators over fq26.
//
// Replaces the reference's NormalAdd / FastDouble layer (src/Commitment.hs:58-176): `nrmlAdd`
// (affine + projective/Jacobian mixed add, :128-144, :156-169), `dbl'` (:111-113) and
// `normalize(s)` / jacToAff (:121-126, :172-176).  Only the resulting GROUP ELEMENT (canonical
// affine x, y, or infinity) is part of the reference's semantics, so the device uses the
// extended-Jacobian "XYZZ" form (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed add 8M+2S,
// general add 12M+2S, no inversion until the final conversion.
//
// Unlike the reference's formulas (incomplete for P = Q, acknowledged at Commitment.hs:98,110)
// every routine here follows the group law for all inputs: infinity, P = Q, P = -Q.
//
// Coordinates are fq26 values with these magnitude invariants (see fq26.hip.h):
//   affine x, y : 1        XYZZ  X <= 5, Y <= 3, ZZ = ZZZ = 1
// Infinity: affine (0,0) (never on y^2 = x^3+7); XYZZ with every limb of ZZ exactly zero (ZZ of a
// finite point is a product of non-zero factors, so it is never 0 mod p and never all-zero).
// In memory: affine = 16 u32 (two canonical 8x32 values, the ABI format); XYZZ = 40 u32 (raw limbs).
#pragma once
#include "fq26.hip.h"

namespace bppp {

struct aff { fq x, y; };
struct xyzz { fq X, Y, ZZ, ZZZ; };
static constexpr int XYZZ_WORDS = 40;   // u32 per stored XYZZ point (160 B)

BPPP_DI bool aff_is_inf(const aff &p) { return fq_all_zero(p.x) && fq_all_zero(p.y); }   // canonical inputs
BPPP_DI bool xyzz_is_inf(const xyzz &p) { return fq_all_zero(p.ZZ); }
BPPP_DI xyzz xyzz_inf() { xyzz r; r.X = fq_zero(); r.Y = fq_zero(); r.ZZ = fq_zero(); r.ZZZ = fq_zero(); return r; }
BPPP_DI aff aff_inf() { aff r; r.x = fq_zero(); r.y = fq_zero(); return r; }
BPPP_DI xyzz xyzz_from_aff(const aff &p) {
  xyzz r; r.X = p.x; r.Y = p.y;
  r.ZZ = fq_zero(); r.ZZ.n[0] = aff_is_inf(p) ? 0u : 1u; r.ZZZ = r.ZZ;
  return r;
}
// negateV (Commitment.hs:102) with the sign folded into the point as normalizeBasis does (:366).
// Input y has magnitude 1; the result keeps magnitude <= 2 (accepted by every consumer below).
BPPP_DI aff aff_cneg(const aff &p, bool neg) {
  aff r; r.x = p.x;
  fq ny = fq_neg<1>(p.y);
  bool inf = aff_is_inf(p);
#pragma unroll
  for (int i = 0; i < 10; i++) r.y.n[i] = (neg && !inf) ? ny.n[i] : p.y.n[i];
  return r;
}

// doubling of an affine point into XYZZ (mdbl-2008-s-1), a = 0.  y magnitude <= 2.
BPPP_DI xyzz xyzz_dbl_aff(const aff &p) {
  if (aff_is_inf(p) || fq_normalizes_to_zero(p.y)) return xyzz_inf();
  fq U = fq_mul_int(p.y, 2);                        // <= 4
  fq V = fq_sqr(U), W = fq_mul(U, V), S = fq_mul(p.x, V);
  fq M = fq_mul_int(fq_sqr(p.x), 3);                // 3
  xyzz r;
  r.X = fq_sub<2>(fq_sqr(M), fq_mul_int(S, 2));     // 1 + 3 = 4
  r.Y = fq_sub<1>(fq_mul(M, fq_sub<4>(S, r.X)), fq_mul(W, p.y));   // 1 + 2 = 3
  r.ZZ = V; r.ZZZ = W;
  return r;
}
// doubling in XYZZ (dbl-2008-s-1), a = 0
BPPP_DI xyzz xyzz_dbl(const xyzz &p) {
  if (xyzz_is_inf(p) || fq_normalizes_to_zero(p.Y)) return xyzz_inf();
  fq U = fq_mul_int(p.Y, 2);                        // <= 6
  fq V = fq_sqr(U), W = fq_mul(U, V), S = fq_mul(p.X, V);
  fq M = fq_mul_int(fq_sqr(p.X), 3);
  xyzz r;
  r.X = fq_sub<2>(fq_sqr(M), fq_mul_int(S, 2));     // 4
  r.Y = fq_sub<1>(fq_mul(M, fq_sub<4>(S, r.X)), fq_mul(W, p.Y));   // 3
  r.ZZ = fq_mul(V, p.ZZ); r.ZZZ = fq_mul(W, p.ZZZ);
  return r;
}

// acc += q (q affine, y magnitude <= 2): the device's nrmlAdd (madd-2008-s), complete.
BPPP_DI void xyzz_madd(xyzz &acc, const aff &q) {
  if (aff_is_inf(q)) return;                                   // nrmlAdd O p = p (Commitment.hs:128)
  if (xyzz_is_inf(acc)) { acc = xyzz_from_aff(q); return; }    // (:129)
  fq U2 = fq_mul(q.x, acc.ZZ), S2 = fq_mul(q.y, acc.ZZZ);
  fq Pd = fq_sub<5>(U2, acc.X);                                // 1 + 6 = 7
  fq R = fq_sub<3>(S2, acc.Y);                                 // 1 + 4 = 5
  if (fq_normalizes_to_zero(Pd)) {                             // same x: P = Q or P = -Q
    if (fq_normalizes_to_zero(R)) acc = xyzz_dbl_aff(q); else acc = xyzz_inf();
    return;
  }
  fq PP = fq_sqr(Pd), PPP = fq_mul(Pd, PP), Q = fq_mul(acc.X, PP);
  fq X3 = fq_sub<3>(fq_sqr(R), fq_add(PPP, fq_mul_int(Q, 2)));                 // 1 + 4 = 5
  fq Y3 = fq_sub<1>(fq_mul(R, fq_sub<5>(Q, X3)), fq_mul(acc.Y, PPP));          // 1 + 2 = 3
  acc.ZZ = fq_mul(acc.ZZ, PP); acc.ZZZ = fq_mul(acc.ZZZ, PPP);
  acc.X = X3; acc.Y = Y3;
}

// acc += q (both XYZZ): add-2008-s, complete.
BPPP_DI void xyzz_add(xyzz &acc, const xyzz &q) {
  if (xyzz_is_inf(q)) return;
  if (xyzz_is_inf(acc)) { acc = q; return; }
  fq U1 = fq_mul(acc.X, q.ZZ), U2 = fq_mul(q.X, acc.ZZ);
  fq S1 = fq_mul(acc.Y, q.ZZZ), S2 = fq_mul(q.Y, acc.ZZZ);
  fq Pd = fq_sub<1>(U2, U1), R = fq_sub<1>(S2, S1);            // 3, 3
  if (fq_normalizes_to_zero(Pd)) {
    if (fq_normalizes_to_zero(R)) acc = xyzz_dbl(acc); else acc = xyzz_inf();
    return;
  }
  fq PP = fq_sqr(Pd), PPP = fq_mul(Pd, PP), Q = fq_mul(U1, PP);
  fq X3 = fq_sub<3>(fq_sqr(R), fq_add(PPP, fq_mul_int(Q, 2)));                 // 5
  fq Y3 = fq_sub<1>(fq_mul(R, fq_sub<5>(Q, X3)), fq_mul(S1, PPP));             // 3
  acc.ZZ = fq_mul(fq_mul(acc.ZZ, q.ZZ), PP);
  acc.ZZZ = fq_mul(fq_mul(acc.ZZZ, q.ZZZ), PPP);
  acc.X = X3; acc.Y = Y3;
}

// normalize / jacToAff (Commitment.hs:121, :172-173): one inversion per point; canonical output.
BPPP_DI aff xyzz_to_aff(const xyzz &p) {
  if (xyzz_is_inf(p)) return aff_inf();
  fq inv = fq_inv(fq_mul(p.ZZ, p.ZZZ));
  aff r;
  r.x = fq_normalize(fq_mul(p.X, fq_mul(inv, p.ZZZ)));   // X / ZZ
  r.y = fq_normalize(fq_mul(p.Y, fq_mul(inv, p.ZZ)));    // Y / ZZZ
  return r;
}

// ---- memory
BPPP_DI aff aff_load(const uint32_t *p) {             // ABI format: canonical x ++ y, 8 x u32 each
  aff r; r.x = fq_from_fe(fe_load(p)); r.y = fq_from_fe(fe_load(p + 8)); return r;
}
BPPP_DI void aff_store(uint32_t *p, const aff &a) {   // a canonical (xyzz_to_aff output) or lazily reduced
  fe_store(p, fq_to_fe(a.x)); fe_store(p + 8, fq_to_fe(a.y));
}
BPPP_DI xyzz xyzz_load(const uint32_t *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint32_t w[40];
#pragma unroll
  for (int i = 0; i < 10; i++) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
  xyzz r;
#pragma unroll
  for (int i = 0; i < 10; i++) { r.X.n[i] = w[i]; r.Y.n[i] = w[10 + i]; r.ZZ.n[i] = w[20 + i]; r.ZZZ.n[i] = w[30 + i]; }
  return r;
}
BPPP_DI void xyzz_store(uint32_t *p, const xyzz &a) {
  uint32_t w[40];
#pragma unroll
  for (int i = 0; i < 10; i++) { w[i] = a.X.n[i]; w[10 + i] = a.Y.n[i]; w[20 + i] = a.ZZ.n[i]; w[30 + i] = a.ZZZ.n[i]; }
  uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 10; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// wave shuffles of whole points (wavefront = 64)
BPPP_DI fq fq_shfl_down(const fq &a, int d) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = (uint32_t)__shfl_down((int)a.n[i], d, 64);
  return r;
}
BPPP_DI xyzz xyzz_shfl_down(const xyzz &p, int d) {
  xyzz r; r.X = fq_shfl_down(p.X, d); r.Y = fq_shfl_down(p.Y, d);
  r.ZZ = fq_shfl_down(p.ZZ, d); r.ZZZ = fq_shfl_down(p.ZZZ, d);
  return r;
}

}  // namespace bppp
