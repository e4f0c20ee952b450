// fr26.hip.h — the scalar field Fr (n = the group order of secp256k1 = 2^256 - R, R 129 bits) in 10 x 26-bit limbs with lazy reduction.
//
// The verifier's and the prover's per-proof field algebra (verifyTRRPM's public scalars, expandChallenges, the phase kernels) is tens of
// thousands of Fr multiplications per proof.  fe.hip.h's 8 x 32-bit multiply is 785 instructions: 120 v_mad_u64_u32 drowned in carry
// chains (v_addc_co_u32 costs as much as a multiply on this chip: 4.3 cycles per wave-instruction, benchmarks/valu_microbench.hip).  Here,
// as in fq26.hip.h, the partial products of a column sum into a 64-bit accumulator WITHOUT carries and additions / negations are ten
// independent 32-bit operations.  What differs from Fq is the modulus: 2^260 = K (mod n) with K = 16 R of 133 bits (six limbs), so the
// high half folds in three passes — 11 x 6, 6 x 6 and 1 x 5 products — instead of Fq's 20: 207 v_mad_u64_u32 per multiplication.
//
// Same role as the reference's Fr arithmetic (`Prime n` of galois-field; FastPrime's addField# / mulField# / negField#,
// src/Data/Field/Galois/FastPrime/Internal.hs:909-988): only canonical values ever leave a kernel (fr_to_fe).
//
// Magnitude rule (as fq26): a value has magnitude m when limb[i] <= 2 m (2^26 - 1) for i < 9 and limb[9] <= 2 m (2^22 - 1).
// mul / sqr take magnitudes <= 8 and return 1; add adds magnitudes; neg<M> / sub<M> take a subtrahend of magnitude <= M and add M + 1.
#pragma once
#include "fe.hip.h"

namespace bppp {

struct fr { uint32_t n[10]; };

static constexpr uint32_t FR_M26 = 0x3FFFFFFu, FR_M22 = 0x3FFFFFu;
// K = 2^260 mod n = 16 R and R = 2^256 mod n, 26-bit limbs
BPPP_DI uint32_t fr_klimb(int i) { return i == 0 ? 0x9BEBF0u : i == 1 ? 0x285CCBFu : i == 2 ? 0x3C4402Du : i == 3 ? 0x2542DD7u : i == 4 ? 0x551231u : 0x5u; }
BPPP_DI uint32_t fr_rlimb(int i) { return i == 0 ? 0x3C9BEBFu : i == 1 ? 0x3685CCBu : i == 2 ? 0x1FC4402u : i == 3 ? 0x6542DDu : 0x1455123u; }
BPPP_DI uint32_t fr_nlimb(int i) {
  return i == 0 ? 0x364141u : i == 1 ? 0x97A334u : i == 2 ? 0x203BBFDu : i == 3 ? 0x39ABD22u : i == 4 ? 0x2BAAEDCu : i == 9 ? FR_M22 : FR_M26;
}
// multiples of n with every limb in [2 M (2^26 - 1), 2 (M + 1) (2^26 - 1)] (limb 9: 2^22): what a subtrahend of magnitude <= M is taken from
BPPP_DI uint32_t fr_subc(int M, int i) {
  const uint32_t T[8][10] = {
      {0xCD90504u, 0xE5E8CCDu, 0xC0EEFF1u, 0xE6AF487u, 0xEEABB70u, 0xFFFFFFBu, 0xFFFFFFCu, 0xFFFFFFCu, 0xFFFFFFCu, 0xFFFFFCu},
      {0x15458786u, 0x178DD333u, 0x141667E9u, 0x15A06ECAu, 0x14601928u, 0x17FFFFF9u, 0x17FFFFFAu, 0x17FFFFFAu, 0x17FFFFFAu, 0x17FFFFAu},
      {0x1DB20A08u, 0x1CBD1999u, 0x1C1DDFE2u, 0x1CD5E90Du, 0x1DD576E0u, 0x1FFFFFF6u, 0x1FFFFFF8u, 0x1FFFFFF8u, 0x1FFFFFF8u, 0x1FFFFF8u},
      {0x261E8C8Au, 0x25EC5FFFu, 0x242557DAu, 0x240B6350u, 0x274AD498u, 0x27FFFFF3u, 0x27FFFFF6u, 0x27FFFFF6u, 0x27FFFFF6u, 0x27FFFF6u},
      {0x2E8B0F0Cu, 0x2F1BA665u, 0x2C2CCFD2u, 0x2F40DD93u, 0x2CC0324Fu, 0x2FFFFFF1u, 0x2FFFFFF4u, 0x2FFFFFF4u, 0x2FFFFFF4u, 0x2FFFFF4u},
      {0x36F7918Eu, 0x344AECCBu, 0x343447CBu, 0x367657D6u, 0x36359007u, 0x37FFFFEEu, 0x37FFFFF2u, 0x37FFFFF2u, 0x37FFFFF2u, 0x37FFFF2u},
      {0x3F641410u, 0x3D7A3331u, 0x3C3BBFC3u, 0x3DABD219u, 0x3FAAEDBFu, 0x3FFFFFEBu, 0x3FFFFFF0u, 0x3FFFFFF0u, 0x3FFFFFF0u, 0x3FFFFF0u},
      {0x47D09692u, 0x46A97997u, 0x444337BBu, 0x44E14C5Cu, 0x45204B77u, 0x47FFFFE9u, 0x47FFFFEEu, 0x47FFFFEEu, 0x47FFFFEEu, 0x47FFFEEu}};
  return T[M - 1][i];
}

BPPP_DI fr fr_zero() { fr r; for (int i = 0; i < 10; i++) r.n[i] = 0; return r; }
BPPP_DI fr fr_one() { fr r = fr_zero(); r.n[0] = 1; return r; }

// ---- the fold shared by mul and sqr: lo[0..9] are the ten low column sums (each < 2^63.4, NOT carry-propagated; lo[9] is the limb
// already extracted from column 9), u[0..10] the carry-normalised high part (26-bit limbs, u[10] the leftover of at most 13 bits):
// value = sum lo[k] 2^(26k) + 2^260 sum u[j] 2^(26j).  Result: magnitude 1.
BPPP_DI fr fr_fold(uint64_t lo[10], const uint32_t u[11]) {
  // pass 1: U x K into columns 0 .. 15 (column k gets u[j] K[k - j]); the low ten become limbs, 10 .. 15 the next high part V
  uint32_t r[10], v[7];
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < 10) c += lo[k];
#pragma unroll
    for (int j = (k > 5 ? k - 5 : 0); j <= (k < 10 ? k : 10); j++) c += (uint64_t)u[j] * fr_klimb(k - j);
    if (k < 10) r[k] = (uint32_t)c & FR_M26; else v[k - 10] = (uint32_t)c & FR_M26;
    c >>= 26;
  }
  v[6] = (uint32_t)c;                                    // U K < 2^(273 + 133): at most a few bits here
  // pass 2: V x K into columns 0 .. 12
  uint32_t w[3];
  c = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    if (k < 10) c += r[k];
#pragma unroll
    for (int j = (k > 5 ? k - 5 : 0); j <= (k < 6 ? k : 6); j++) c += (uint64_t)v[j] * fr_klimb(k - j);
    if (k < 10) r[k] = (uint32_t)c & FR_M26; else w[k - 10] = (uint32_t)c & FR_M26;
    c >>= 26;
  }
  w[2] = (uint32_t)c;                                    // zero in fact (V K < 2^(7 * 26 + 133 - ...)); kept for exactness
  // pass 3: everything above 2^256 — the top four bits of limb 9 and W — times R = 2^256 mod n (five limbs)
  const uint64_t e = (uint64_t)(r[9] >> 22) + ((uint64_t)w[0] << 4) + ((uint64_t)w[1] << 30) + ((uint64_t)w[2] << 56);
  r[9] &= FR_M22;
  fr out;
  c = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    c += r[k];
    if (k < 5) c += e * fr_rlimb(k);                     // e < 2^36 here (W < 2^32 in practice): < 2^62
    if (k < 9) { out.n[k] = (uint32_t)c & FR_M26; c >>= 26; } else out.n[9] = (uint32_t)c;   // limb 9 <= 2^22 + carry: magnitude 1
  }
  return out;
}

#define FR_COL(acc, k)                                                                 \
  _Pragma("unroll") for (int i = ((k) > 9 ? (k)-9 : 0); i <= ((k) < 9 ? (k) : 9); i++) \
      acc += (uint64_t)a.n[i] * b.n[(k)-i];

BPPP_DI fr fr_mul(const fr &a, const fr &b) {
  uint64_t lo[10];
  uint32_t u[11];
  uint64_t d = 0;
  FR_COL(d, 9)
  lo[9] = (uint32_t)d & FR_M26; d >>= 26;
#pragma unroll
  for (int k = 10; k <= 18; k++) {
    FR_COL(d, k)
    u[k - 10] = (uint32_t)d & FR_M26; d >>= 26;
  }
  u[9] = (uint32_t)d & FR_M26; u[10] = (uint32_t)(d >> 26);       // leftover carry < 2^38
#pragma unroll
  for (int k = 0; k < 9; k++) { uint64_t c = 0; FR_COL(c, k) lo[k] = c; }
  return fr_fold(lo, u);
}

#define FR_SQCOL(acc, k)                                                                    \
  _Pragma("unroll") for (int i = ((k) > 9 ? (k)-9 : 0); 2 * i < (k); i++)                   \
      acc += (uint64_t)a2[i] * a.n[(k)-i];                                                   \
  if (((k)&1) == 0) acc += (uint64_t)a.n[(k) / 2] * a.n[(k) / 2];

BPPP_DI fr fr_sqr(const fr &a) {
  uint32_t a2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) a2[i] = a.n[i] << 1;     // < 2^31 for magnitude <= 8
  uint64_t lo[10];
  uint32_t u[11];
  uint64_t d = 0;
  FR_SQCOL(d, 9)
  lo[9] = (uint32_t)d & FR_M26; d >>= 26;
#pragma unroll
  for (int k = 10; k <= 18; k++) {
    FR_SQCOL(d, k)
    u[k - 10] = (uint32_t)d & FR_M26; d >>= 26;
  }
  u[9] = (uint32_t)d & FR_M26; u[10] = (uint32_t)(d >> 26);
#pragma unroll
  for (int k = 0; k < 9; k++) { uint64_t c = 0; FR_SQCOL(c, k) lo[k] = c; }
  return fr_fold(lo, u);
}

// ---- carry-free linear operations
BPPP_DI fr fr_add(const fr &a, const fr &b) {
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + b.n[i];
  return r;
}
template <int M> BPPP_DI fr fr_neg(const fr &a) {        // -a for a of magnitude <= M; result magnitude M + 1
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = fr_subc(M, i) - a.n[i];
  return r;
}
template <int MB> BPPP_DI fr fr_sub(const fr &a, const fr &b) {   // a - b for b of magnitude <= MB; result magnitude mag(a) + MB + 1
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + (fr_subc(MB, i) - b.n[i]);
  return r;
}
BPPP_DI fr fr_mul_int(const fr &a, uint32_t k) {
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = a.n[i] * k;
  return r;
}

// ---- normalisation: one folding pass leaves a value < 2^256 + 2^160 in limbs < 2^26 (limb 9 may carry bit 22)
BPPP_DI void fr_weak_pass(uint32_t t[10]) {
  const uint32_t x = t[9] >> 22; t[9] &= FR_M22;          // x < 2^10 for magnitudes <= 16
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    c += t[i];
    if (i < 5) c += (uint64_t)x * fr_rlimb(i);
    if (i < 9) { t[i] = (uint32_t)c & FR_M26; c >>= 26; } else t[9] = (uint32_t)c;
  }
}
// canonical representative in [0, n), limbs < 2^26
BPPP_DI fr fr_normalize(const fr &a) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = a.n[i];
  fr_weak_pass(t);
  fr_weak_pass(t);           // absorbs a possible bit 256 of the first pass: now t < 2^256
  // s = t + R; if it reaches 2^256 then t >= n and the answer is s - 2^256
  uint32_t s[10], cy = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const uint32_t v = t[i] + cy + (i < 5 ? fr_rlimb(i) : 0u);
    if (i < 9) { s[i] = v & FR_M26; cy = v >> 26; } else s[i] = v;
  }
  const bool ge = (s[9] >> 22) != 0;
  s[9] &= FR_M22;
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = ge ? s[i] : t[i];
  return r;
}
BPPP_DI bool fr_is_zero(const fr &a) {                    // a = 0 (mod n); any magnitude <= 16
  const fr t = fr_normalize(a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) o |= t.n[i];
  return o == 0;
}

// ---- "reduced" linear operations: the result is carry-propagated and folded back to magnitude 1, whatever chain of them produced the
// inputs (magnitude <= 2 each).  ~40 instructions; for loops that accumulate and for code that does not track magnitudes.
BPPP_DI fr fr_weak(const fr &a) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = a.n[i];
  fr_weak_pass(t);
  fr r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.n[i] = t[i];
  return r;
}
BPPP_DI fr fr_addr(const fr &a, const fr &b) { return fr_weak(fr_add(a, b)); }
BPPP_DI fr fr_subr(const fr &a, const fr &b) { return fr_weak(fr_sub<2>(a, b)); }
BPPP_DI fr fr_negr(const fr &a) { return fr_weak(fr_neg<2>(a)); }
BPPP_DI fr fr_dblr(const fr &a) { return fr_weak(fr_add(a, a)); }

// ---- conversion to / from the canonical 8 x 32-bit form used in memory (same bit layout as fq26)
BPPP_DI fr fr_from_fe(const fe &a) {
  fr r;
  r.n[0] = a.v[0] & FR_M26;
  r.n[1] = ((a.v[0] >> 26) | (a.v[1] << 6)) & FR_M26;
  r.n[2] = ((a.v[1] >> 20) | (a.v[2] << 12)) & FR_M26;
  r.n[3] = ((a.v[2] >> 14) | (a.v[3] << 18)) & FR_M26;
  r.n[4] = ((a.v[3] >> 8) | (a.v[4] << 24)) & FR_M26;
  r.n[5] = (a.v[4] >> 2) & FR_M26;
  r.n[6] = ((a.v[4] >> 28) | (a.v[5] << 4)) & FR_M26;
  r.n[7] = ((a.v[5] >> 22) | (a.v[6] << 10)) & FR_M26;
  r.n[8] = ((a.v[6] >> 16) | (a.v[7] << 16)) & FR_M26;
  r.n[9] = a.v[7] >> 10;
  return r;
}
BPPP_DI fe fr_to_fe(const fr &a_) {
  const fr a = fr_normalize(a_);
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
BPPP_DI fr fr_load(const uint32_t *p) { return fr_from_fe(fe_load(p)); }          // canonical 8 x 32 in memory
BPPP_DI void fr_store(uint32_t *p, const fr &a) { fe_store(p, fr_to_fe(a)); }

}  // namespace bppp
