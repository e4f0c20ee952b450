// fq26.hip.h — the coordinate field Fq (p = 2^256 - 2^32 - 977) in 10 x 26-bit limbs with lazy reduction.
//
// Why this representation on gfx950 (measured, benchmarks/valu_microbench.hip): v_mad_u64_u32 costs about
// the same as any other multiply or 64-bit add (~4.8 cycles per wave-instruction per SIMD) while 32-bit
// add / and / mov cost ~2.4.  With 8 x 32-bit limbs every partial product drags a carry: the compiler's
// multiply is 453 instructions around 73 v_mad_u64_u32.  With 26-bit limbs the 100 partial products of a
// column-wise (Comba) product sum into 64-bit accumulators WITHOUT carries — one v_mad_u64_u32 each and
// nothing else — and additions / negations are 10 independent 32-bit adds with no carry chain at all.
//
// Same role as the reference's FastPrime layer (src/Data/Field/Galois/FastPrime/Internal.hs:909-988:
// addField#, negField#, mulField#, sqrField#, invField#); only canonical values ever leave the device,
// which is all the reference's semantics fix.  The algorithm is modelled bit-exactly (with 64-bit overflow
// assertions at worst-case magnitudes) in benchmarks/fe26_model.py.
//
// Magnitude rule: a value has magnitude m when limb[i] <= 2*m*(2^26-1) for i < 9 and limb[9] <= 2*m*(2^22-1).
// mul / sqr take magnitudes <= 8 and return 1; add adds magnitudes; neg<M> takes <= M and returns M + 1.
#pragma once
#include "fe.hip.h"
#include "modinv.hip.h"

namespace bppp {

struct fq { uint32_t n[10]; };

static constexpr uint32_t FQ_M26 = 0x3FFFFFFu, FQ_M22 = 0x3FFFFFu;
static constexpr uint32_t FQ_R0 = 0x3D10u, FQ_R1 = 0x400u;   // 2^260 = R1 * 2^26 + R0 (mod p)

BPPP_DI uint32_t fq_plimb(int i) { return i == 0 ? 0x3FFFC2Fu : i == 1 ? 0x3FFFFBFu : i == 9 ? FQ_M22 : FQ_M26; }

BPPP_DI fq fq_zero() { fq r; for (int i = 0; i < 10; i++) r.n[i] = 0; return r; }
BPPP_DI fq fq_one() { fq r = fq_zero(); r.n[0] = 1; return r; }
BPPP_DI bool fq_all_zero(const fq &a) {        // exact all-limbs-zero test (the infinity marker)
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) o |= a.n[i];
  return o == 0;
}

// ---- multiplication: 100 carry-free v_mad_u64_u32 + 20 for the 2^260 fold
// Cost model (benchmarks/valu_microbench.hip, >= 2 wavefronts per SIMD): v_mad_u64_u32, every 64-bit add / shift and every VOP3 32-bit
// op issue in ~4.8 cycles per wave-instruction, VOP1/VOP2 32-bit ops (and, add, mov, shift) in ~2.8.  What a multiplication costs beyond
// its 120 products is therefore counted in 64-bit adds and shifts.
// Round 1-3 form (kept as fq_mul_cols for the record, benchmarks/fqmul_variants.hip): plain C columns — the compiler starts every column
// from zero and adds the previous column's carry with one v_lshl_add_u64, and turns u * R1 (R1 = 2^10) into a 64-bit shift plus a 64-bit
// add: 111 v_mad_u64_u32 + 32 v_lshl_add_u64 + 22 v_lshrrev_b64 + 11 v_lshlrev_b64 + 33 v_mov + 23 v_and = 233 instructions, 176 G/s.
// Round 4 form: TWO carry chains — H = columns 9 .. 18, L = columns 0 .. 8 with the fold of H's limbs — in which the carry of a column is
// the ADDEND of the next column's first product (no 64-bit add), pinned product by product (an empty asm after each), with R0 / R1 held in
// scalar registers so that u * R1 stays one v_mad_u64_u32: 119 v_mad_u64_u32 + 7 v_lshl_add_u64 + 22 v_lshrrev_b64 + 3 v_lshlrev_b64 +
// 6 v_mov + 23 v_and (+ 42 s_nop 0 the compiler puts between back-to-back dependent products: hidden by the other wavefronts of the
// SIMD) — 203 G/s in the same microbenchmark, bit-identical results (also at the magnitude-8 bounds); k_acc_points 1.106 -> 1.04 ms.
// A third form that fills those slots with the 45 low products as free column sums (benchmarks/fqmul_variants.h v3: no s_nop, 15
// v_lshl_add_u64) measures 196 G/s there and the same in the kernels: not kept.  ISA histograms: profiles/r04_fq_mul_isa_histogram.txt.
BPPP_DI uint64_t fq_madc(uint32_t x, uint32_t y, uint64_t acc) { uint64_t r = (uint64_t)x * y + acc; asm("" : "+v"(r)); return r; }
BPPP_DI uint32_t fq_sreg(uint32_t v) { asm("" : "+s"(v)); return v; }
// the tail shared by mul and sqr: limb 9 and everything above 2^256 folded back (2^256 = 2^32 + 0x3D1)
BPPP_DI void fq_mul_tail(fq &r, uint64_t c, uint32_t t9, uint64_t u9, uint32_t u8) {
  c += (uint64_t)t9 + u9 * FQ_R0 + (uint64_t)u8 * FQ_R1;
  r.n[9] = (uint32_t)c & FQ_M22;
  const uint64_t top = (c >> 22) + ((u9 * FQ_R1) << 4);   // units of 2^256 = 2^32 + 0x3D1
  c = (uint64_t)r.n[0] + top * 0x3D1u; r.n[0] = (uint32_t)c & FQ_M26; c >>= 26;
  c += (uint64_t)r.n[1] + (top << 6); r.n[1] = (uint32_t)c & FQ_M26; c >>= 26;
  c += r.n[2]; r.n[2] = (uint32_t)c & FQ_M26; c >>= 26;
  r.n[3] += (uint32_t)c;
}
BPPP_DI fq fq_mul(const fq &a, const fq &b) {
  const uint32_t R0 = fq_sreg(FQ_R0), R1 = fq_sreg(FQ_R1);
  uint64_t d = 0;
#pragma unroll
  for (int i = 0; i <= 9; i++) d = fq_madc(a.n[i], b.n[9 - i], d);
  const uint32_t t9 = (uint32_t)d & FQ_M26; d >>= 26;
  uint32_t u[9];
#pragma unroll
  for (int k = 10; k <= 18; k++) {
#pragma unroll
    for (int i = k - 9; i <= 9; i++) d = fq_madc(a.n[i], b.n[k - i], d);
    u[k - 10] = (uint32_t)d & FQ_M26; d >>= 26;
  }
  const uint64_t u9 = d;                           // leftover carry, < 2^38
  fq r;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) c = fq_madc(a.n[i], b.n[k - i], c);
    c = fq_madc(u[k], R0, c);
    if (k) c = fq_madc(u[k - 1], R1, c);
    r.n[k] = (uint32_t)c & FQ_M26; c >>= 26;
  }
  fq_mul_tail(r, c, t9, u9, u[8]);
  return r;
}

// squaring: 55 products (off-diagonal terms use the doubled limb), the same two chains
#define FQ_SQCHAIN(acc, k)                                                                              \
  _Pragma("unroll") for (int i = ((k) > 9 ? (k)-9 : 0); 2 * i < (k); i++) acc = fq_madc(a2[i], a.n[(k)-i], acc); \
  if (((k)&1) == 0) acc = fq_madc(a.n[(k) / 2], a.n[(k) / 2], acc);

BPPP_DI fq fq_sqr(const fq &a) {
  const uint32_t R0 = fq_sreg(FQ_R0), R1 = fq_sreg(FQ_R1);
  uint32_t a2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) a2[i] = a.n[i] << 1;     // < 2^31 for magnitude <= 8
  uint64_t d = 0;
  FQ_SQCHAIN(d, 9)
  const uint32_t t9 = (uint32_t)d & FQ_M26; d >>= 26;
  uint32_t u[9];
#pragma unroll
  for (int k = 10; k <= 18; k++) {
    FQ_SQCHAIN(d, k)
    u[k - 10] = (uint32_t)d & FQ_M26; d >>= 26;
  }
  const uint64_t u9 = d;
  fq r;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    FQ_SQCHAIN(c, k)
    c = fq_madc(u[k], R0, c);
    if (k) c = fq_madc(u[k - 1], R1, c);
    r.n[k] = (uint32_t)c & FQ_M26; c >>= 26;
  }
  fq_mul_tail(r, c, t9, u9, u[8]);
  return r;
}

// the round 1-3 form of the multiplication (plain columns, the compiler's own order): a cross-check for tests and benchmarks
#define FQ_COL(acc, k)                                                                 \
  _Pragma("unroll") for (int i = ((k) > 9 ? (k)-9 : 0); i <= ((k) < 9 ? (k) : 9); i++) \
      acc += (uint64_t)a.n[i] * b.n[(k)-i];
BPPP_DI fq fq_mul_cols(const fq &a, const fq &b) {
  uint64_t d = 0;
  FQ_COL(d, 9)
  const uint32_t t9 = (uint32_t)d & FQ_M26; d >>= 26;
  uint32_t u[9];
#pragma unroll
  for (int k = 10; k <= 18; k++) {
    FQ_COL(d, k)
    u[k - 10] = (uint32_t)d & FQ_M26; d >>= 26;
  }
  const uint64_t u9 = d;
  fq r;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    FQ_COL(c, k)
    c += (uint64_t)u[k] * FQ_R0;
    if (k) c += (uint64_t)u[k - 1] * FQ_R1;
    r.n[k] = (uint32_t)c & FQ_M26; c >>= 26;
  }
  fq_mul_tail(r, c, t9, u9, u[8]);
  return r;
}

// ---- carry-free linear operations
BPPP_DI fq fq_add(const fq &a, const fq &b) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + b.n[i];
  return r;
}
// -a for a of magnitude <= M; result magnitude M + 1 (negField#, Internal.hs:927-932)
template <int M> BPPP_DI fq fq_neg(const fq &a) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = 2u * (M + 1) * fq_plimb(i) - a.n[i];
  return r;
}
// a - b for b of magnitude <= MB; result magnitude mag(a) + MB + 1
template <int MB> BPPP_DI fq fq_sub(const fq &a, const fq &b) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + (2u * (MB + 1) * fq_plimb(i) - b.n[i]);
  return r;
}
BPPP_DI fq fq_mul_int(const fq &a, uint32_t k) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] * k;
  return r;
}

// ---- normalisation
// one folding pass: result < 2^256 + small, limbs < 2^26 (limb 9 may carry bit 22)
BPPP_DI void fq_weak_pass(uint32_t t[10]) {
  uint32_t x = t[9] >> 22; t[9] &= FQ_M22;
  t[0] += x * 0x3D1u; t[1] += x << 6;
#pragma unroll
  for (int i = 0; i < 9; i++) { t[i + 1] += t[i] >> 26; t[i] &= FQ_M26; }
}
// true iff a = 0 (mod p); any magnitude <= 16
BPPP_DI bool fq_normalizes_to_zero(const fq &a) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = a.n[i];
  fq_weak_pass(t);
  uint32_t z0 = 0, z1 = FQ_M26;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    z0 |= t[i];
    uint32_t pat = i == 0 ? 0x3D0u : i == 1 ? 0x40u : i == 9 ? 0x3C00000u : 0u;
    z1 &= t[i] ^ pat;
  }
  return (z0 == 0) | (z1 == FQ_M26);
}
// canonical representative in [0, p), limbs < 2^26
BPPP_DI fq fq_normalize(const fq &a) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = a.n[i];
  fq_weak_pass(t);
  fq_weak_pass(t);           // second pass absorbs a possible bit 256; now t < 2^256
  // s = t + (2^256 - p); if it reaches 2^256 then t >= p and the answer is s - 2^256
  uint32_t s[10];
  uint32_t cy = 0x3D1u;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    uint32_t v = t[i] + cy + (i == 1 ? 0x40u : 0u);
    if (i < 9) { s[i] = v & FQ_M26; cy = v >> 26; } else { s[i] = v; }
  }
  bool ge = (s[9] >> 22) != 0;
  s[9] &= FQ_M22;
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = ge ? s[i] : t[i];
  return r;
}

// ---- conversion to / from the canonical 8 x 32-bit form used in memory at the ABI boundary
BPPP_DI fq fq_from_fe(const fe &a) {
  fq r;
  r.n[0] = a.v[0] & FQ_M26;
  r.n[1] = ((a.v[0] >> 26) | (a.v[1] << 6)) & FQ_M26;
  r.n[2] = ((a.v[1] >> 20) | (a.v[2] << 12)) & FQ_M26;
  r.n[3] = ((a.v[2] >> 14) | (a.v[3] << 18)) & FQ_M26;
  r.n[4] = ((a.v[3] >> 8) | (a.v[4] << 24)) & FQ_M26;
  r.n[5] = (a.v[4] >> 2) & FQ_M26;
  r.n[6] = ((a.v[4] >> 28) | (a.v[5] << 4)) & FQ_M26;
  r.n[7] = ((a.v[5] >> 22) | (a.v[6] << 10)) & FQ_M26;
  r.n[8] = ((a.v[6] >> 16) | (a.v[7] << 16)) & FQ_M26;
  r.n[9] = a.v[7] >> 10;
  return r;
}
BPPP_DI fe fq_to_fe(const fq &a_) {
  fq a = fq_normalize(a_);
  fe r;
  r.v[0] = a.n[0] | (a.n[1] << 26);
  r.v[1] = (a.n[1] >> 6) | (a.n[2] << 20);
  r.v[2] = (a.n[2] >> 12) | (a.n[3] << 14);
  r.v[3] = (a.n[3] >> 18) | (a.n[4] << 8);
  r.v[4] = (a.n[4] >> 24) | (a.n[5] << 2) | (a.n[6] << 28);
  r.v[5] = (a.n[6] >> 4) | (a.n[7] << 22);
  r.v[6] = (a.n[7] >> 10) | (a.n[8] << 16);
  r.v[7] = (a.n[8] >> 16) | (a.n[9] << 10);
  return r;
}

// ---- inversion a^(p-2) by the standard secp256k1 addition chain (255 squarings + 15 multiplications);
// 0 -> 0 like batchInverse (src/Data/Field/BatchInverse.hs:18).  invField# (Internal.hs:981-983) uses GMP.
BPPP_DI fq fq_sqr_n(fq x, int n) {
  for (int i = 0; i < n; i++) x = fq_sqr(x);
  return x;
}
BPPP_DI fq fq_inv_fermat(const fq &a) {
  fq x2 = fq_mul(fq_sqr(a), a);
  fq x3 = fq_mul(fq_sqr(x2), a);
  fq x6 = fq_mul(fq_sqr_n(x3, 3), x3);
  fq x9 = fq_mul(fq_sqr_n(x6, 3), x3);
  fq x11 = fq_mul(fq_sqr_n(x9, 2), x2);
  fq x22 = fq_mul(fq_sqr_n(x11, 11), x11);
  fq x44 = fq_mul(fq_sqr_n(x22, 22), x22);
  fq x88 = fq_mul(fq_sqr_n(x44, 44), x44);
  fq x176 = fq_mul(fq_sqr_n(x88, 88), x88);
  fq x220 = fq_mul(fq_sqr_n(x176, 44), x44);
  fq x223 = fq_mul(fq_sqr_n(x220, 3), x3);
  fq t = fq_mul(fq_sqr_n(x223, 23), x22);
  t = fq_mul(fq_sqr_n(t, 5), a);
  t = fq_mul(fq_sqr_n(t, 3), x2);
  t = fq_mul(fq_sqr_n(t, 2), a);
  return t;
}
// square-root candidate a^((p+1)/4) (p = 3 mod 4) by the same kind of chain: (p+1)/4 has three runs of ones, 223, 22 and 2 long:
// 253 squarings + 13 multiplications.  The caller checks r^2 = a (a non-residue gives a root of -a).  pointX of the reference
// (app/Main.hs:68-72, src/Encoding.hs:97-103) bottoms out here.
BPPP_DI fq fq_sqrt_candidate(const fq &a) {
  fq x2 = fq_mul(fq_sqr(a), a);
  fq x3 = fq_mul(fq_sqr(x2), a);
  fq x6 = fq_mul(fq_sqr_n(x3, 3), x3);
  fq x9 = fq_mul(fq_sqr_n(x6, 3), x3);
  fq x11 = fq_mul(fq_sqr_n(x9, 2), x2);
  fq x22 = fq_mul(fq_sqr_n(x11, 11), x11);
  fq x44 = fq_mul(fq_sqr_n(x22, 22), x22);
  fq x88 = fq_mul(fq_sqr_n(x44, 44), x44);
  fq x176 = fq_mul(fq_sqr_n(x88, 88), x88);
  fq x220 = fq_mul(fq_sqr_n(x176, 44), x44);
  fq x223 = fq_mul(fq_sqr_n(x220, 3), x3);
  fq t = fq_mul(fq_sqr_n(x223, 23), x22);
  t = fq_mul(fq_sqr_n(t, 6), x2);
  return fq_sqr_n(t, 2);
}
// production inverse: safegcd division steps on the canonical value (modinv.hip.h): ~60 multiplications' worth of instructions instead
// of the chain's 270, the same for every lane; 0 -> 0
BPPP_DI fq fq_inv(const fq &a) { return fq_from_fe(fe_modinv<0>(fq_to_fe(a))); }

// ---- memory: a lazily-reduced element is stored as its 10 raw limbs (40 B)
BPPP_DI void fq_store10(uint32_t *p, const fq &a) {
#pragma unroll
  for (int i = 0; i < 10; i++) p[i] = a.n[i];
}
BPPP_DI fq fq_load10(const uint32_t *p) {
  fq r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = p[i];
  return r;
}

}  // namespace bppp
