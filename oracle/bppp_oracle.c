/*
 * bppp_oracle.c — CPU restatement of the reference's hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X build.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (bulletproofspp_amd/lib/libbppp_hip.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests — the reference ships no
 * golden vectors, KATs or working test-suite for this path, its arithmetic
 * lives in un-vendored Hackage packages (elliptic-curve-0.3.0, galois-field-1.0.1,
 * stack.yaml:44-45) and no GHC exists in the build container.  What this oracle IS pinned by:
 *   (i)   the reference's constants (generator, beta, lambda, 3^160 limb dump, 2^256-n limbs;
 *         src/Data/Curve/Weierstrass/FastSECP256K1.hs:37-60,134-141,
 *         src/Data/Field/Galois/FastPrime/Internal.hs:48-51,108-126) — tests/test_oracle_constants.py
 *   (ii)  OpenSSL libcrypto's secp256k1 (EC_POINT_mul / EC_POINT_add) on random inputs —
 *         oracle/openssl_check.c, tests/test_oracle_openssl.py
 *   (iii) an independent pure-Python big-int restatement (oracle/pyoracle.py).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Scalars ("Fr") are integers mod the group order n; coordinates ("Fq") integers mod p.
 * Limb layout everywhere: 4 x uint64 little-endian (limb 0 = least significant), the layout of
 * FastPrime's (# Word#,Word#,Word#,Word# #) (Internal.hs:152-176) and of Encoding.hs:75-86.
 * Affine points: 8 x uint64 = x[4] ++ y[4]; the point at infinity is encoded (0,0)
 * (0^3+7 != 0, so (0,0) is never on the curve).
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;
typedef struct { fe m; uint64_t r[3]; } modulus; /* m = 2^256 - r */

/* p = 2^256 - 2^32 - 977 (FastSECP256K1.hs:33); n = group order (FastSECP256K1.hs:47) */
static const modulus MP = {{{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}},
                           {0x1000003D1ULL, 0, 0}};
/* 2^256 - n = 0x1_4551231950b75fc4_402da1732fc9bebf (Internal.hs:48-51) */
static const modulus MN = {{{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}},
                           {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1}};

static const modulus *pick(int which) { return which ? &MN : &MP; }

static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static int fe_cmp(const fe *a, const fe *b) {
  for (int i = 3; i >= 0; i--) { if (a->v[i] < b->v[i]) return -1; if (a->v[i] > b->v[i]) return 1; }
  return 0;
}
static uint64_t raw_add(fe *o, const fe *a, const fe *b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; o->v[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t raw_sub(fe *o, const fe *a, const fe *b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->v[i] - b->v[i] - br; o->v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
/* addField# (Internal.hs:909-924): (a+b) mod m, fully reduced */
static void fe_add(fe *o, const fe *a, const fe *b, const modulus *M) {
  fe t; uint64_t c = raw_add(&t, a, b);
  if (c || fe_cmp(&t, &M->m) >= 0) raw_sub(&t, &t, &M->m);
  *o = t;
}
static void fe_sub(fe *o, const fe *a, const fe *b, const modulus *M) {
  fe t; if (raw_sub(&t, a, b)) raw_add(&t, &t, &M->m);
  *o = t;
}
/* negField# (Internal.hs:927-932) */
static void fe_neg(fe *o, const fe *a, const modulus *M) {
  if (fe_is_zero(a)) { *o = *a; return; }
  raw_sub(o, &M->m, a);
}
/* mulField# (Internal.hs:943-956): schoolbook 4x4 (mul256With256# :483-575), then fold the
 * high half by multiplying with r = 2^256 - m until it vanishes (the reference unrolls three
 * "carry x r" folds; a loop gives the same fully reduced value in [0, m)). */
static void fe_mul(fe *o, const fe *a, const fe *b, const modulus *M) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[i] * b->v[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  for (;;) {
    if ((t[4] | t[5] | t[6] | t[7]) == 0) break;
    uint64_t hi[4] = {t[4], t[5], t[6], t[7]}, n[8] = {t[0], t[1], t[2], t[3], 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 3; j++) { c += (u128)hi[i] * M->r[j] + n[i + j]; n[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + 3; k < 8 && c; k++) { c += n[k]; n[k] = (uint64_t)c; c >>= 64; }
    }
    memcpy(t, n, sizeof t);
  }
  fe r = {{t[0], t[1], t[2], t[3]}};
  while (fe_cmp(&r, &M->m) >= 0) raw_sub(&r, &r, &M->m);
  *o = r;
}
static void fe_sqr(fe *o, const fe *a, const modulus *M) { fe_mul(o, a, a, M); } /* sqrField# :960-973 */
/* invField# (Internal.hs:981-983) uses GMP recipModBigNat; same value via Fermat, 0 -> 0 */
static void fe_inv(fe *o, const fe *a, const modulus *M) {
  fe e, two = {{2, 0, 0, 0}}, acc = {{1, 0, 0, 0}}, base = *a;
  raw_sub(&e, &M->m, &two);
  for (int i = 0; i < 256; i++) {
    if ((e.v[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base, M);
    fe_sqr(&base, &base, M);
  }
  *o = acc;
}

/* batchInverse (src/Data/Field/BatchInverse.hs:14-24): Montgomery trick, 0 -> 0 */
static void fe_batch_inverse(fe *xs, size_t n, const modulus *M) {
  if (!n) return;
  fe *pre = (fe *)malloc(n * sizeof(fe));
  fe acc = {{1, 0, 0, 0}};
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fe_is_zero(&xs[i])) fe_mul(&acc, &acc, &xs[i], M); }
  fe y; fe_inv(&y, &acc, M);
  for (size_t i = n; i-- > 0;) {
    if (fe_is_zero(&xs[i])) continue;
    fe inv; fe_mul(&inv, &y, &pre[i], M);
    fe_mul(&y, &y, &xs[i], M);
    xs[i] = inv;
  }
  free(pre);
}

/* ------------------------------------------------------------------ curve: y^2 = x^3 + 7 */
typedef struct { fe x, y; } aff;           /* (0,0) = infinity */
typedef struct { fe X, Y, Z; } jac;        /* Z = 0 = infinity (Commitment.hs:129,173) */
static const fe FE_ONE = {{1, 0, 0, 0}};
static int aff_is_inf(const aff *a) { return fe_is_zero(&a->x) && fe_is_zero(&a->y); }
static void jac_set_inf(jac *j) { memset(j, 0, sizeof *j); j->X = FE_ONE; j->Y = FE_ONE; }

/* dbl' (Commitment.hs:111-113) -> elliptic-curve's Jacobian `dbl`; a = 0 so dbl-2009-l.
 * The group element is formula-independent; only the canonical affine image is compared. */
static void jac_dbl(jac *o, const jac *p) {
  const modulus *M = &MP;
  if (fe_is_zero(&p->Z) || fe_is_zero(&p->Y)) { jac_set_inf(o); return; }
  fe A, B, C, D, E, F, t, X3, Y3, Z3;
  fe_sqr(&A, &p->X, M); fe_sqr(&B, &p->Y, M); fe_sqr(&C, &B, M);
  fe_add(&t, &p->X, &B, M); fe_sqr(&t, &t, M); fe_sub(&t, &t, &A, M); fe_sub(&t, &t, &C, M); fe_add(&D, &t, &t, M);
  fe_add(&E, &A, &A, M); fe_add(&E, &E, &A, M);
  fe_sqr(&F, &E, M);
  fe_sub(&X3, &F, &D, M); fe_sub(&X3, &X3, &D, M);
  fe_sub(&t, &D, &X3, M); fe_mul(&Y3, &E, &t, M);
  fe_add(&t, &C, &C, M); fe_add(&t, &t, &t, M); fe_add(&t, &t, &t, M); fe_sub(&Y3, &Y3, &t, M);
  fe_mul(&Z3, &p->Y, &p->Z, M); fe_add(&Z3, &Z3, &Z3, M);
  o->X = X3; o->Y = Y3; o->Z = Z3;
}

/* nrmlAdd for Jacobian points (Commitment.hs:128-144), formula names as in the reference.
 * The reference's formula is incomplete for P = Q (h = 0 gives Z3 = 0; acknowledged at
 * Commitment.hs:98,110); the oracle follows the group law there: h = 0, r = 0 -> doubling,
 * h = 0, r != 0 -> infinity.  */
static void jac_nrml_add(jac *o, const aff *a, const jac *p) {
  const modulus *M = &MP;
  if (aff_is_inf(a)) { *o = *p; return; }                                   /* :128 */
  if (fe_is_zero(&p->Z)) { o->X = a->x; o->Y = a->y; o->Z = FE_ONE; return; } /* :129 */
  fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3, tmp;
  fe_sqr(&z1z1, &p->Z, M);
  fe_mul(&u2, &a->x, &z1z1, M);
  fe_mul(&s2, &a->y, &p->Z, M); fe_mul(&s2, &s2, &z1z1, M);
  fe_sub(&h, &u2, &p->X, M);
  fe_sub(&r, &s2, &p->Y, M);
  if (fe_is_zero(&h)) {
    if (fe_is_zero(&r)) { jac_dbl(o, p); return; }
    jac_set_inf(o); return;
  }
  fe_add(&r, &r, &r, M);
  fe_sqr(&hh, &h, M);
  fe_add(&i, &hh, &hh, M); fe_add(&i, &i, &i, M);
  fe_mul(&j, &h, &i, M);
  fe_mul(&v, &p->X, &i, M);
  fe_mul(&t, &p->Y, &j, M);
  fe_sqr(&x3, &r, M); fe_sub(&x3, &x3, &j, M); fe_sub(&x3, &x3, &v, M); fe_sub(&x3, &x3, &v, M);
  fe_sub(&tmp, &v, &x3, M); fe_mul(&y3, &r, &tmp, M); fe_sub(&y3, &y3, &t, M); fe_sub(&y3, &y3, &t, M);
  fe_add(&tmp, &p->Z, &h, M); fe_sqr(&z3, &tmp, M); fe_sub(&z3, &z3, &z1z1, M); fe_sub(&z3, &z3, &hh, M);
  o->X = x3; o->Y = y3; o->Z = z3;
}

/* normalize / jacToAff (Commitment.hs:121,172-173) */
static void jac_to_aff(aff *o, const jac *p) {
  const modulus *M = &MP;
  if (fe_is_zero(&p->Z)) { memset(o, 0, sizeof *o); return; }
  fe zi, zi2, zi3; fe_inv(&zi, &p->Z, M); fe_sqr(&zi2, &zi, M); fe_mul(&zi3, &zi2, &zi, M);
  fe_mul(&o->x, &p->X, &zi2, M); fe_mul(&o->y, &p->Y, &zi3, M);
}
static void aff_neg(aff *o, const aff *a) { o->x = a->x; fe_neg(&o->y, &a->y, &MP); }

/* reduceScalar for Prime p (Commitment.hs:276-279): signed representative in (-n/2, n/2].
 * Returns magnitude in *mag and sign (1 = negative). */
static int reduce_scalar(fe *mag, const fe *s) {
  fe neg; raw_sub(&neg, &MN.m, s);          /* n - s */
  if (fe_cmp(s, &neg) > 0) { *mag = neg; return 1; }
  *mag = *s; return 0;
}

/* innerProduct, default 256-row bit-serial Straus (Commitment.hs:325-335) with the
 * Prime-p instance's normalizeBasis / addBasis (:355-367) and getDigit = testBit |s| (:288).
 * Points arrive affine, so `normalizes` (:123-126) is the identity on them. */
void orc_inner_product(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t *out_xy) {
  aff *bs = (aff *)malloc((n ? n : 1) * sizeof(aff));
  fe *ss = (fe *)malloc((n ? n : 1) * sizeof(fe));
  for (size_t i = 0; i < n; i++) {
    fe s; memcpy(&s, scalars + 4 * i, 32);
    aff g; memcpy(&g, points + 8 * i, 64);
    int neg = reduce_scalar(&ss[i], &s);                 /* first reduceScalar <$> sgs (:329) */
    if (neg && !aff_is_inf(&g)) aff_neg(&bs[i], &g); else bs[i] = g;  /* sign (:366) */
  }
  jac v; jac_set_inf(&v);
  for (int row = 256; row >= 1; row--) {                 /* go len zeroV (:334-335) */
    jac_dbl(&v, &v);
    for (size_t i = 0; i < n; i++)
      if ((ss[i].v[(row - 1) >> 6] >> ((row - 1) & 63)) & 1) jac_nrml_add(&v, &bs[i], &v);
  }
  aff r; jac_to_aff(&r, &v);
  memcpy(out_xy, &r, 64);
  free(bs); free(ss);
}

/* projectivePairIP (Commitment.hs:343-353): b*g0 + a*g1 with 129 rows
 * (rationalReducedScalarLength = 129, :286).  Scalars arrive as sign + 192-bit magnitude
 * (3 limbs; values are < 2^130). */
void orc_pair_ip(const uint64_t *s0_mag, int s0_neg, const uint64_t *g0,
                 const uint64_t *s1_mag, int s1_neg, const uint64_t *g1, uint64_t *out_xy) {
  aff b[2]; memcpy(&b[0], g0, 64); memcpy(&b[1], g1, 64);
  if (s0_neg && !aff_is_inf(&b[0])) aff_neg(&b[0], &b[0]);
  if (s1_neg && !aff_is_inf(&b[1])) aff_neg(&b[1], &b[1]);
  const uint64_t *m[2] = {s0_mag, s1_mag};
  jac v; jac_set_inf(&v);
  for (int row = 129; row >= 1; row--) {
    jac_dbl(&v, &v);
    for (int k = 0; k < 2; k++)
      if ((m[k][(row - 1) >> 6] >> ((row - 1) & 63)) & 1) jac_nrml_add(&v, &b[k], &v);
  }
  aff r; jac_to_aff(&r, &v);
  memcpy(out_xy, &r, 64);
}

/* collapsePoints b a over a whole vector (Bulletproof.hs:213-214 mapped by mapHalves,
 * Bulletproof.hs:88-90): adjacent pairs (g0,g1),(g2,g3),..; odd length pads with zeroV. */
void orc_fold_points(const uint64_t *b_mag, int b_neg, const uint64_t *a_mag, int a_neg,
                     const uint64_t *pts, size_t n, uint64_t *out) {
  uint64_t inf[8] = {0};
  for (size_t j = 0; j < (n + 1) / 2; j++) {
    const uint64_t *gl = pts + 16 * j;
    const uint64_t *gr = (2 * j + 1 < n) ? pts + 16 * j + 8 : inf;
    orc_pair_ip(b_mag, b_neg, gl, a_mag, a_neg, gr, out + 8 * j);
  }
}

/* ------------------------------------------------------------------ signed big integers for
 * rationalReduceScalar (Commitment.hs:242-255).  5 x 64-bit magnitude + sign. */
#define SB 5
typedef struct { uint64_t m[SB]; int neg; } sbig;
static int mag_cmp(const uint64_t *a, const uint64_t *b) {
  for (int i = SB - 1; i >= 0; i--) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
  return 0;
}
static int mag_zero(const uint64_t *a) { uint64_t o = 0; for (int i = 0; i < SB; i++) o |= a[i]; return o == 0; }
static void mag_add(uint64_t *o, const uint64_t *a, const uint64_t *b) {
  u128 c = 0; for (int i = 0; i < SB; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
}
static void mag_sub(uint64_t *o, const uint64_t *a, const uint64_t *b) { /* a >= b */
  uint64_t br = 0;
  for (int i = 0; i < SB; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
static void sb_add(sbig *o, const sbig *a, const sbig *b) {
  sbig r;
  if (a->neg == b->neg) { mag_add(r.m, a->m, b->m); r.neg = a->neg; }
  else if (mag_cmp(a->m, b->m) >= 0) { mag_sub(r.m, a->m, b->m); r.neg = a->neg; }
  else { mag_sub(r.m, b->m, a->m); r.neg = b->neg; }
  if (mag_zero(r.m)) r.neg = 0;
  *o = r;
}
static void sb_mul(sbig *o, const sbig *a, const sbig *b) { /* truncated to SB limbs; inputs small enough */
  sbig r; memset(&r, 0, sizeof r);
  for (int i = 0; i < SB; i++) {
    u128 c = 0;
    for (int j = 0; i + j < SB; j++) { c += (u128)a->m[i] * b->m[j] + r.m[i + j]; r.m[i + j] = (uint64_t)c; c >>= 64; }
  }
  r.neg = mag_zero(r.m) ? 0 : (a->neg ^ b->neg);
  *o = r;
}
static int mag_bits(const uint64_t *a) {
  for (int i = SB - 1; i >= 0; i--) if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
  return 0;
}
/* `quot`: truncation toward zero (Commitment.hs:254) */
static void sb_quot(sbig *q, const sbig *a, const sbig *b) {
  uint64_t rem[SB] = {0}, quo[SB] = {0};
  int nb = mag_bits(a->m);
  for (int i = nb - 1; i >= 0; i--) {
    /* rem = rem*2 + bit */
    uint64_t c = (a->m[i >> 6] >> (i & 63)) & 1;
    for (int k = 0; k < SB; k++) { uint64_t nc = rem[k] >> 63; rem[k] = (rem[k] << 1) | c; c = nc; }
    if (mag_cmp(rem, b->m) >= 0) { mag_sub(rem, rem, b->m); quo[i >> 6] |= 1ULL << (i & 63); }
  }
  memcpy(q->m, quo, sizeof quo);
  q->neg = mag_zero(quo) ? 0 : (a->neg ^ b->neg);
}

/* rationalReduceScalar for Prime p (Commitment.hs:242-255 with the instance at :269-288):
 * the list `egcd (n,0) (x^,1)` starts at its SECOND argument (:252); return the first (r,s)
 * with r^2 <= 2n (:247).  Outputs: a = r, b = s as sign + 3-limb magnitude (< 2^130). */
void orc_rational_reduce(const uint64_t *x, uint64_t *a_mag, int *a_neg, uint64_t *b_mag, int *b_neg) {
  fe xs; memcpy(&xs, x, 32);
  fe mag; int neg = reduce_scalar(&mag, &xs);
  sbig pr, ps, cr, cs, two_n;
  memset(&pr, 0, sizeof pr); memset(&ps, 0, sizeof ps); memset(&cr, 0, sizeof cr); memset(&cs, 0, sizeof cs);
  memcpy(pr.m, MN.m.v, 32);                 /* (pRed, 0) */
  memcpy(cr.m, mag.v, 32); cr.neg = fe_is_zero(&mag) ? 0 : neg;  /* (reduceScalar x, 1) */
  cs.m[0] = 1;
  memset(&two_n, 0, sizeof two_n); memcpy(two_n.m, MN.m.v, 32); mag_add(two_n.m, two_n.m, two_n.m);
  for (;;) {
    /* cond: |r|^2 > 2n.  |r| < 2^257; if it has more than 129 bits its square exceeds 2n < 2^258 */
    int big = mag_bits(cr.m) > 130;
    if (!big) { sbig sq; sbig t = cr; t.neg = 0; /* need full square: up to 260 bits fits in 5 limbs */
      sb_mul(&sq, &t, &t); big = mag_cmp(sq.m, two_n.m) > 0; }
    if (!big) break;
    sbig q, t, nr, ns;
    sb_quot(&q, &pr, &cr);
    sb_mul(&t, &q, &cr); t.neg ^= 1; if (mag_zero(t.m)) t.neg = 0; sb_add(&nr, &pr, &t);
    sb_mul(&t, &q, &cs); t.neg ^= 1; if (mag_zero(t.m)) t.neg = 0; sb_add(&ns, &ps, &t);
    pr = cr; ps = cs; cr = nr; cs = ns;
  }
  memcpy(a_mag, cr.m, 24); *a_neg = cr.neg;
  memcpy(b_mag, cs.m, 24); *b_neg = cs.neg;
}

/* ------------------------------------------------------------------ thin exported helpers */
void orc_fe_add(const uint64_t *a, const uint64_t *b, uint64_t *o, int which) { fe_add((fe *)o, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_sub(const uint64_t *a, const uint64_t *b, uint64_t *o, int which) { fe_sub((fe *)o, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, int which) { fe_mul((fe *)o, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_inv(const uint64_t *a, uint64_t *o, int which) { fe_inv((fe *)o, (const fe *)a, pick(which)); }
void orc_fe_batch_inverse(uint64_t *xs, size_t n, int which) { fe_batch_inverse((fe *)xs, n, pick(which)); }
void orc_fe_mul_many(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n, int which) {
  for (size_t i = 0; i < n; i++) fe_mul((fe *)(o + 4 * i), (const fe *)(a + 4 * i), (const fe *)(b + 4 * i), pick(which));
}
/* complete affine + affine (group law), used to combine partial sums and by tests */
void orc_point_add(const uint64_t *p, const uint64_t *q, uint64_t *out_xy) {
  aff a, b, r; memcpy(&a, p, 64); memcpy(&b, q, 64);
  jac j;
  if (aff_is_inf(&a)) jac_set_inf(&j); else { j.X = a.x; j.Y = a.y; j.Z = FE_ONE; }
  jac_nrml_add(&j, &b, &j);
  jac_to_aff(&r, &j); memcpy(out_xy, &r, 64);
}
/* s *^ p (Commitment.hs:107) by plain double-and-add over the unsigned scalar */
void orc_point_mul(const uint64_t *s, const uint64_t *p, uint64_t *out_xy) {
  aff a; memcpy(&a, p, 64);
  jac v; jac_set_inf(&v);
  for (int i = 255; i >= 0; i--) {
    jac_dbl(&v, &v);
    if ((s[i >> 6] >> (i & 63)) & 1) jac_nrml_add(&v, &a, &v);
  }
  aff r; jac_to_aff(&r, &v); memcpy(out_xy, &r, 64);
}
int orc_on_curve(const uint64_t *p) {
  aff a; memcpy(&a, p, 64);
  if (aff_is_inf(&a)) return 1;
  if (fe_cmp(&a.x, &MP.m) >= 0 || fe_cmp(&a.y, &MP.m) >= 0) return 0;
  fe l, r, seven = {{7, 0, 0, 0}};
  fe_sqr(&l, &a.y, &MP); fe_sqr(&r, &a.x, &MP); fe_mul(&r, &r, &a.x, &MP); fe_add(&r, &r, &seven, &MP);
  return fe_eq(&l, &r);
}
/* y from x: p = 3 mod 4 so sqrt = a^((p+1)/4); returns 0 if x^3+7 is a non-residue.
 * Mirrors pointX in getPoints (app/Main.hs:68-72); which root galois-field's `sr` returns is
 * unverifiable offline, the build's documented choice is the EVEN root. */
int orc_lift_x(const uint64_t *x, uint64_t *out_xy) {
  fe xx; memcpy(&xx, x, 32);
  if (fe_cmp(&xx, &MP.m) >= 0) return 0;
  fe rhs, seven = {{7, 0, 0, 0}}, e, acc = FE_ONE, base, one = FE_ONE, chk;
  fe_sqr(&rhs, &xx, &MP); fe_mul(&rhs, &rhs, &xx, &MP); fe_add(&rhs, &rhs, &seven, &MP);
  raw_add(&e, &MP.m, &one);                 /* p + 1 (no overflow: p+1 < 2^256) */
  for (int k = 0; k < 2; k++) {             /* >> 2 */
    for (int i = 0; i < 4; i++) e.v[i] = (e.v[i] >> 1) | (i < 3 ? e.v[i + 1] << 63 : 0);
  }
  base = rhs;
  for (int i = 0; i < 256; i++) {
    if ((e.v[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base, &MP);
    fe_sqr(&base, &base, &MP);
  }
  fe_sqr(&chk, &acc, &MP);
  if (!fe_eq(&chk, &rhs)) return 0;
  if (acc.v[0] & 1) fe_neg(&acc, &acc, &MP);
  memcpy(out_xy, &xx, 32); memcpy(out_xy + 4, &acc, 32);
  return 1;
}
