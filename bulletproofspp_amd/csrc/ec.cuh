// ec.cuh — secp256k1 group law on the device: affine inputs, XYZZ accumulators.
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
// Infinity: affine (0,0) (never on y^2 = x^3+7); XYZZ with ZZ = 0.
#pragma once
#include "fe.cuh"

namespace bppp {

struct aff { fe x, y; };
struct xyzz { fe X, Y, ZZ, ZZZ; };

BPPP_DI bool aff_is_inf(const aff &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
BPPP_DI bool xyzz_is_inf(const xyzz &p) { return fe_is_zero(p.ZZ); }
BPPP_DI xyzz xyzz_inf() { xyzz r; r.X = fe_zero(); r.Y = fe_zero(); r.ZZ = fe_zero(); r.ZZZ = fe_zero(); return r; }
BPPP_DI aff aff_inf() { aff r; r.x = fe_zero(); r.y = fe_zero(); return r; }
BPPP_DI xyzz xyzz_from_aff(const aff &p) {
  xyzz r; r.X = p.x; r.Y = p.y;
  bool inf = aff_is_inf(p);
  r.ZZ = inf ? fe_zero() : fe_one(); r.ZZZ = r.ZZ;
  return r;
}
// negateV (Commitment.hs:102) with the sign folded into the point as normalizeBasis does (:366)
BPPP_DI aff aff_cneg(const aff &p, bool neg) {
  aff r; r.x = p.x;
  fe ny = fe_neg<0>(p.y);
#pragma unroll
  for (int i = 0; i < 8; i++) r.y.v[i] = neg ? ny.v[i] : p.y.v[i];
  return r;
}

typedef fe (*fpfn)(const fe &, const fe &);
#define FPM(a, b) fe_mul<0>(a, b)
#define FPS(a) fe_sqr<0>(a)
#define FPA(a, b) fe_add<0>(a, b)
#define FPB(a, b) fe_sub<0>(a, b)

// doubling of an affine point into XYZZ (mdbl-2008-s-1), a = 0
BPPP_DI xyzz xyzz_dbl_aff(const aff &p) {
  if (aff_is_inf(p) || fe_is_zero(p.y)) return xyzz_inf();
  fe U = FPA(p.y, p.y), V = FPS(U), W = FPM(U, V), S = FPM(p.x, V);
  fe xx = FPS(p.x), M = FPA(FPA(xx, xx), xx);
  xyzz r;
  r.X = FPB(FPB(FPS(M), S), S);
  r.Y = FPB(FPM(M, FPB(S, r.X)), FPM(W, p.y));
  r.ZZ = V; r.ZZZ = W;
  return r;
}
// doubling in XYZZ (dbl-2008-s-1), a = 0
BPPP_DI xyzz xyzz_dbl(const xyzz &p) {
  if (xyzz_is_inf(p) || fe_is_zero(p.Y)) return xyzz_inf();
  fe U = FPA(p.Y, p.Y), V = FPS(U), W = FPM(U, V), S = FPM(p.X, V);
  fe xx = FPS(p.X), M = FPA(FPA(xx, xx), xx);
  xyzz r;
  r.X = FPB(FPB(FPS(M), S), S);
  r.Y = FPB(FPM(M, FPB(S, r.X)), FPM(W, p.Y));
  r.ZZ = FPM(V, p.ZZ); r.ZZZ = FPM(W, p.ZZZ);
  return r;
}

// acc += q (q affine): the device's nrmlAdd (madd-2008-s), complete.
BPPP_DI void xyzz_madd(xyzz &acc, const aff &q) {
  if (aff_is_inf(q)) return;                                   // nrmlAdd O p = p (Commitment.hs:128)
  if (xyzz_is_inf(acc)) { acc = xyzz_from_aff(q); return; }    // (:129)
  fe U2 = FPM(q.x, acc.ZZ), S2 = FPM(q.y, acc.ZZZ);
  fe Pd = FPB(U2, acc.X), R = FPB(S2, acc.Y);
  if (fe_is_zero(Pd)) {                                        // same x: P = Q or P = -Q
    if (fe_is_zero(R)) acc = xyzz_dbl_aff(q); else acc = xyzz_inf();
    return;
  }
  fe PP = FPS(Pd), PPP = FPM(Pd, PP), Q = FPM(acc.X, PP);
  fe X3 = FPB(FPB(FPB(FPS(R), PPP), Q), Q);
  fe Y3 = FPB(FPM(R, FPB(Q, X3)), FPM(acc.Y, PPP));
  acc.ZZ = FPM(acc.ZZ, PP); acc.ZZZ = FPM(acc.ZZZ, PPP);
  acc.X = X3; acc.Y = Y3;
}

// acc += q (both XYZZ): add-2008-s, complete.
BPPP_DI void xyzz_add(xyzz &acc, const xyzz &q) {
  if (xyzz_is_inf(q)) return;
  if (xyzz_is_inf(acc)) { acc = q; return; }
  fe U1 = FPM(acc.X, q.ZZ), U2 = FPM(q.X, acc.ZZ);
  fe S1 = FPM(acc.Y, q.ZZZ), S2 = FPM(q.Y, acc.ZZZ);
  fe Pd = FPB(U2, U1), R = FPB(S2, S1);
  if (fe_is_zero(Pd)) {
    if (fe_is_zero(R)) acc = xyzz_dbl(acc); else acc = xyzz_inf();
    return;
  }
  fe PP = FPS(Pd), PPP = FPM(Pd, PP), Q = FPM(U1, PP);
  fe X3 = FPB(FPB(FPB(FPS(R), PPP), Q), Q);
  fe Y3 = FPB(FPM(R, FPB(Q, X3)), FPM(S1, PPP));
  acc.ZZ = FPM(FPM(acc.ZZ, q.ZZ), PP);
  acc.ZZZ = FPM(FPM(acc.ZZZ, q.ZZZ), PPP);
  acc.X = X3; acc.Y = Y3;
}

// normalize / jacToAff (Commitment.hs:121, :172-173): one inversion per point.
__device__ __noinline__ aff xyzz_to_aff(const xyzz &p) {
  if (xyzz_is_inf(p)) return aff_inf();
  fe inv = fe_inv<0>(FPM(p.ZZ, p.ZZZ));
  aff r;
  r.x = FPM(p.X, FPM(inv, p.ZZZ));   // X / ZZ
  r.y = FPM(p.Y, FPM(inv, p.ZZ));    // Y / ZZZ
  return r;
}

// ---- memory helpers: affine = 16 u32 (x ++ y); xyzz = 32 u32
BPPP_DI aff aff_load(const uint32_t *p) { aff r; r.x = fe_load(p); r.y = fe_load(p + 8); return r; }
BPPP_DI void aff_store(uint32_t *p, const aff &a) { fe_store(p, a.x); fe_store(p + 8, a.y); }
BPPP_DI xyzz xyzz_load(const uint32_t *p) {
  xyzz r; r.X = fe_load(p); r.Y = fe_load(p + 8); r.ZZ = fe_load(p + 16); r.ZZZ = fe_load(p + 24); return r;
}
BPPP_DI void xyzz_store(uint32_t *p, const xyzz &a) {
  fe_store(p, a.X); fe_store(p + 8, a.Y); fe_store(p + 16, a.ZZ); fe_store(p + 24, a.ZZZ);
}

// wave shuffles of whole points (wavefront = 64)
BPPP_DI fe fe_shfl_down(const fe &a, int d) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_down((int)a.v[i], d, 64);
  return r;
}
BPPP_DI xyzz xyzz_shfl_down(const xyzz &p, int d) {
  xyzz r; r.X = fe_shfl_down(p.X, d); r.Y = fe_shfl_down(p.Y, d);
  r.ZZ = fe_shfl_down(p.ZZ, d); r.ZZZ = fe_shfl_down(p.ZZZ, d);
  return r;
}

}  // namespace bppp
