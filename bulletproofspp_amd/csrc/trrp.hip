// trrp.hip — the verifier's public scalars of B typed-reciprocal range proofs, derived on the device from their challenges.
//
// Replaces, for a batch, the scalar work of verifyTRRPM (src/RangeProof/TypedReciprocal.hs:449-467): makePhase2s with the unit
// witness (:185-205), makeSharedCoeffs (:213-216), makePublicConsts (:246-274), makeBpCoeffs (:391-396) and the opening scalars
// of TranscriptTRRP (:293-297, inputCoeffs :325-328).  Inputs per proof: the seven Fiat-Shamir challenges (e, x, r0, q, x', r1, t)
// — hashing stays with the injected oracle on the host (src/ZKP.hs:96-101).  Outputs are exactly the per-proof arrays
// bppp_nl_verify_batch_device consumes (q, sp, pub_norm, pub_lin_c, initCom scalars), written where it will read them, so a
// batch of fresh proofs goes from challenges to the combined MSM without its O(nrmLen + linLen) scalars crossing PCIe.
//
// A group of lanes per proof (16: four proofs per wavefront).  All field inversions of a proof — e, q0 and every (e + symbol) of the reciprocal argument — are ONE
// inversion: block-wide Montgomery trick (prefix and suffix product scans in LDS).  Fr arithmetic in 10 x 26-bit lazy limbs (fr26.hip.h:
// 413 instructions per multiplication against 785 for the 8 x 32 form), canonical 8 x 32 values in memory.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "ctx.hpp"
#include "fe.hip.h"
#include "fr26.hip.h"
#include "modinv.hip.h"
#include "trrp.hpp"
#include "../../include/bppp.h"

namespace bppp {

// Fr multiply / square as real functions: ~80 call sites would otherwise inline to 37 k instructions (220 KB of code, several
// times the instruction cache) for a kernel whose wavefronts all sit in different phases
__device__ __noinline__ fr frm(fr a, fr b) { return fr_mul(a, b); }
__device__ __noinline__ fr frs(fr a) { return fr_sqr(a); }

BPPP_DI fr fr_pow_u32(fr base, uint32_t e) {
  fr acc = fr_one();
  while (e) { if (e & 1u) acc = frm(acc, base); base = frs(base); e >>= 1; }
  return acc;
}
BPPP_DI fr lds_get(const uint32_t *p, uint32_t i) { fr r; for (int k = 0; k < 10; k++) r.n[k] = p[i * 10 + k]; return r; }
BPPP_DI void lds_put(uint32_t *p, uint32_t i, const fr &a) { for (int k = 0; k < 10; k++) p[i * 10 + k] = a.n[k]; }


// G lanes per proof (G | 64), 64 / G proofs per wavefront: lane t of a group owns the contiguous chunks [t*C, (t+1)*C) of every list, so
// the running powers of q0 advance by one multiplication per position.  The per-LANE start-up (first powers of q0 and q0^-1, the
// prefix / suffix product scans of the inversion, the slot tables: ~100 multiplications) does not shrink with the chunk, and the one
// safegcd inversion (~14 k instructions on a single lane) is paid per wavefront whatever it serves: with a whole wavefront per proof
// (round 1) start-up was two thirds of the ~11 k multiplications per 64by64 proof.  G = 16: the chunks are four times longer, the
// scans two steps shorter and one inversion pass serves four proofs.  (A 256-lane workgroup per proof spent three times the
// single-wavefront figure on start-up alone.)
template <int G>
__global__ void __launch_bounds__(64) k_trrp_public(TrrpDims D, uint32_t batch, uint32_t lds_words_per_proof, const uint32_t *__restrict__ pos_kind,
                                                    const uint32_t *__restrict__ pos_range,
                                                    const uint32_t *__restrict__ pos_slot, const uint32_t *__restrict__ pos_sym,
                                                    const uint32_t *__restrict__ pos_coeff, const uint32_t *__restrict__ range_min,
                                                    const uint32_t *__restrict__ range_assumed, const uint32_t *__restrict__ syms,
                                                    const uint32_t *__restrict__ cs_slot, const uint32_t *__restrict__ cs_sym,
                                                    const uint32_t *__restrict__ pub_is_out, const uint32_t *__restrict__ pub_amount,
                                                    const uint32_t *__restrict__ pub_sym, const uint32_t *__restrict__ ch,
                                                    uint32_t *__restrict__ out_q, uint32_t *__restrict__ out_sp, uint32_t *__restrict__ out_norm,
                                                    uint32_t *__restrict__ out_cs, uint32_t *__restrict__ out_init) {
  extern __shared__ uint32_t lds_all[];
  const uint32_t t = threadIdx.x % G, grp = threadIdx.x / G, m = 2 + D.nsyms;
  const uint32_t b_raw = blockIdx.x * (64 / G) + grp;
  const bool live = b_raw < batch;                 // a group past the end recomputes the last proof and stores nothing (it must keep up with the barriers)
  const uint32_t b = live ? b_raw : batch - 1;
  uint32_t *lds = lds_all + (size_t)grp * lds_words_per_proof;
  uint32_t *inv = lds;                         // [m] the inverted list: e, q0, e + sym_k (holds the lanes' running products first)
  uint32_t *sa = inv + (size_t)m * 10;          // [G] scan scratch A
  uint32_t *sb = sa + G * 10;                   // [G] scan scratch B
  uint32_t *x2 = sb + G * 10;                   // [nr]  x^(2(j+1))
  uint32_t *sl = x2 + (size_t)D.nr * 10;        // [3][TRRP_MAX_SLOTS] per base slot: t^2 v, 2 t^5 v / e, 2 t^3 v   (v = x^(3+2 slot))
  const uint32_t *c = ch + (size_t)b * 56;
  const fr e = fr_load(c), x = fr_load(c + 8), r0 = fr_load(c + 16), q = fr_load(c + 24), xp = fr_load(c + 32), r1 = fr_load(c + 40), tt = fr_load(c + 48);
  fr q0 = frs(q);                         // qPowers' : q^2 (NL, NormArgument.hs:148) or -q^2 (IP, InnerProductArgument.hs:231)
  if (D.flavour) q0 = fr_negr(q0);

  // ---- one inversion for the whole proof (batchInverse semantics: 0 -> 0)
  const uint32_t K = (m + G - 1) / G, lo = min(m, t * K), hi = min(m, lo + K);
  fr local = fr_one();
  for (uint32_t i = lo; i < hi; i++) {
    fr a = i == 0 ? e : i == 1 ? q0 : fr_addr(e, fr_load(syms + (size_t)(i - 2) * 8));
    lds_put(inv, i, local);                      // product of this lane's elements before element i
    if (!fr_is_zero(a)) local = frm(local, a);
  }
  lds_put(sa, t, local); lds_put(sb, t, local);
  __syncthreads();
  for (int d = 1; d < G; d <<= 1) {            // inclusive prefix (sa) and suffix (sb) products over the lanes of the group
    fr pa = lds_get(sa, t), pb = lds_get(sb, t);
    fr oa = (int)t - d >= 0 ? lds_get(sa, t - d) : fr_one();
    fr ob = t + d < (uint32_t)G ? lds_get(sb, t + d) : fr_one();
    __syncthreads();
    lds_put(sa, t, frm(pa, oa)); lds_put(sb, t, frm(pb, ob));
    __syncthreads();
  }
  fr others = frm(t ? lds_get(sa, t - 1) : fr_one(), t + 1 < (uint32_t)G ? lds_get(sb, t + 1) : fr_one());
  fr total = lds_get(sa, G - 1);
  __syncthreads();
  if (t == 0) lds_put(sa, 0, fr_from_fe(fe_modinv<1>(fr_to_fe(total))));        // one active lane per group: division steps (~14 k instructions), all groups in one pass
  __syncthreads();
  {
    fr suf = frm(lds_get(sa, 0), others);  // 1 / (product of this lane's own elements), then times the ones already passed
    for (uint32_t i = hi; i-- > lo;) {
      fr a = i == 0 ? e : i == 1 ? q0 : fr_addr(e, fr_load(syms + (size_t)(i - 2) * 8));   // recomputed: one LDS array instead of two
      if (fr_is_zero(a)) { lds_put(inv, i, fr_zero()); continue; }
      lds_put(inv, i, frm(suf, lds_get(inv, i)));
      suf = frm(suf, a);
    }
  }
  // ---- per-range and per-base-slot tables
  const fr xx = frs(x), x3 = frm(xx, x);
  const fr t2 = frs(tt), t3 = frm(t2, tt), t4 = frs(t2), t5 = frm(t4, tt), t6 = frs(t3);
  const fr two_t5 = fr_dblr(t5), two_t3 = fr_dblr(t3);
  {                                                           // x^(2(j+1)) for this lane's ranges j = t, t + G, ...: one power, then steps of x^(2G)
    fr xj = fr_pow_u32(xx, t + 1);
    const fr xg = fr_pow_u32(xx, G);
    for (uint32_t j = t; j < D.nr; j += G) { lds_put(x2, j, xj); xj = frm(xj, xg); }
  }
  __syncthreads();
  const fr e_inv = lds_get(inv, 0), q0_inv = lds_get(inv, 1);
  for (uint32_t s_ = t; s_ < (uint32_t)TRRP_MAX_SLOTS; s_ += G) {
    fr v = frm(x3, fr_pow_u32(xx, s_));                      // makeBaseMap: x^3, x^5, ... (:349)
    lds_put(sl, s_, frm(t2, v));
    lds_put(sl, TRRP_MAX_SLOTS + s_, frm(frm(two_t5, e_inv), v));
    lds_put(sl, 2 * TRRP_MAX_SLOTS + s_, frm(two_t3, v));
  }
  __syncthreads();

  // ---- norm positions: publicTerms (TypedReciprocal.hs:262-274) with u, v, c of makePhase2s (:193-205), regrouped:
  //   digit:  p = t^2 e + q^-2i (t^2 v + t^3 u + t^4 c),        ts0 = q^2i (p^2 + 2 t^5) + 2 t^5 v / e
  //   type:   p = t^2 e + t^3 x' + q^-2i (t^2 v + t^3 x' u),     ts0 = q^2i p^2
  fr acc = fr_zero();
  {
    const uint32_t C = (D.nlen + G - 1) / G, plo = min(D.nlen, t * C), phi = min(D.nlen, plo + C);
    fr q2 = fr_pow_u32(q0, plo + 1), qi2 = fr_pow_u32(q0_inv, plo + 1);
    const fr t2e = frm(t2, e), t3xp = frm(t3, xp), t2e_t = fr_addr(t2e, t3xp);
    for (uint32_t i = plo; i < phi; i++) {
      const uint32_t kf = pos_kind[i], kind = kf & 0xFFu;
      const fr xr = lds_get(x2, pos_range[i]);
      fr p, ts0;
      if (kind == K_TYPING) {
        fr A = frm(t2, (kf & F_IO) ? fr_negr(x) : x);
        if (!(kf & F_IA)) A = fr_addr(A, frm(t3xp, xr));
        p = fr_addr(t2e_t, frm(qi2, A));
        ts0 = frm(q2, frs(p));
      } else {
        const uint32_t slot = pos_slot[i], sy = pos_sym[i];
        fr A = fr_addr(lds_get(sl, slot), frm(t3, frm(xr, fr_load(pos_coeff + (size_t)i * 8))));
        if (kind == K_INLINE && sy != NO_SYM) {
          const fr si = lds_get(inv, 2 + sy);                      // "if s == 0 then 0" is tested on the INVERTED value (:205): e + s = 0 gives c = 0
          if (!fr_is_zero(si)) A = fr_addr(A, frm(frm(t2, lds_get(sl, slot)), fr_subr(e_inv, si)));      // t^4 v = t^2 (t^2 v): the slot table has t^2 v
        }
        p = fr_addr(t2e, frm(qi2, A));
        ts0 = fr_addr(frm(q2, fr_addr(frs(p), two_t5)), lds_get(sl, TRRP_MAX_SLOTS + slot));
      }
      if (live) fr_store(out_norm + ((size_t)b * D.nlen + i) * 8, p);
      acc = fr_addr(acc, ts0);
      q2 = frm(q2, q0); qi2 = frm(qi2, q0_inv);
    }
  }
  // z (:254): -2 t^5 sum_j min_j x^(2(j+1))  -  [typed] 2 t^5 x pubSum
  for (uint32_t j = t; j < D.nr; j += G)
    if (!range_assumed[j]) acc = fr_subr(acc, frm(two_t5, frm(fr_load(range_min + (size_t)j * 8), lds_get(x2, j))));
  if (D.has_types)
    for (uint32_t j = t; j < D.npub; j += G) {
      fr term = frm(frm(two_t5, x), frm(fr_load(pub_amount + (size_t)j * 8), lds_get(inv, 2 + pub_sym[j])));
      acc = pub_is_out[j] ? fr_addr(acc, term) : fr_subr(acc, term);
    }
  __syncthreads();
  lds_put(sa, t, acc);
  __syncthreads();
  for (int d = G / 2; d >= 1; d >>= 1) {
    if ((int)t < d) lds_put(sa, t, fr_addr(lds_get(sa, t), lds_get(sa, t + d)));
    __syncthreads();
  }
  if (t == 0 && live) { fr_store(out_sp + (size_t)b * 8, lds_get(sa, 0)); fr_store(out_q + (size_t)b * 8, q); }

  // ---- linear weights: makeBpCoeffs (:391-396) over makeSharedCoeffs (:213-216)
  const fr rs = frm(r0, r1);
  if (live)
    for (uint32_t j = t; j < D.llen; j += G) {
      fr v;
      if (j == 0) v = D.has_types ? fr_negr(xp) : fr_zero();
      else if (j == 1) v = frm(rs, tt);
      else if (j == 2) v = frm(rs, t2);
      else if (j == 3) v = frm(rs, t3);
      else if (j == 4) v = frm(r0, t4);
      else if (j == 5) v = frm(rs, t6);
      else v = frm(lds_get(sl, 2 * TRRP_MAX_SLOTS + cs_slot[j - 6]), fr_subr(e_inv, lds_get(inv, 2 + cs_sym[j - 6])));
      fr_store(out_cs + ((size_t)b * D.llen + j) * 8, v);
    }
  // ---- initCom scalars in commitment order blCom : rCom : dmCom : mCom : nComs  (openWith of TranscriptTRRP, :293-297)
  const uint32_t ninit = 4 + D.nr;
  fr qr = D.has_types ? fr_pow_u32(q0, t + 1) : fr_zero();
  const fr qG = D.has_types ? fr_pow_u32(q0, G) : fr_zero();
  if (live) {
    for (uint32_t r = t; r < D.nr; r += G) {
      fr ic = range_assumed[r] ? fr_zero() : lds_get(x2, r);                     // inputCoeffs (:325-328)
      if (D.has_types) { ic = fr_addr(ic, qr); qr = frm(qr, qG); }
      fr_store(out_init + ((size_t)b * ninit + 4 + r) * 8, frm(two_t5, ic));
    }
    if (t < 4) fr_store(out_init + ((size_t)b * ninit + t) * 8, t == 0 ? fr_one() : t == 1 ? t3 : t == 2 ? t2 : tt);
  }
}


// ------------------------------------------------------------------------------------------------ the same scalars in three kernels
// k_trrp_public above keeps ~30 field elements live per lane (320 VGPRs: ONE wavefront per SIMD, where an instruction issues every ~5.5
// cycles) and repeats the per-proof start-up in every lane.  For batches that fill the chip the work is split so that every kernel has
// a small live set (two to four wavefronts per SIMD) and the per-proof part runs once:
//   k_trrp_pre   16 lanes per proof: the batched inversion (e, q0, e + symbols), x^(2(j+1)), the slot tables, q0^(C t + 1) and
//                q0^-(C t + 1) for the 64 chunk starts, every challenge-derived constant — into a per-proof scratch record in HBM
//                (raw 10-limb values of magnitude 1, csrc/fr26.hip.h)
//   k_trrp_pos   one wavefront per proof: lane t walks positions [C t, C (t + 1)); the record is read at wave-uniform addresses
//   k_trrp_lin   one lane per linear weight / initCom scalar
// Same outputs as k_trrp_public bit for bit (canonical values; tests/test_gpu_rangeproof.py, tests/test_gpu_native_verify.py).
enum : uint32_t { TC_Q = 0, TC_Q0, TC_Q0I, TC_EI, TC_T, TC_T2, TC_T3, TC_2T5, TC_T2E, TC_T3XP, TC_T2ET, TC_T2X, TC_T2NX, TC_2T5X, TC_NXP, TC_L1, TC_L2, TC_L3, TC_L4, TC_L5,
                  TC_COUNT };
struct TrrpRec { uint32_t inv, x2, sl, qs, qis, words; };       // word offsets of the tables inside a proof's record (constants first)
__host__ __device__ inline TrrpRec trrp_rec(const TrrpDims &D) {
  TrrpRec r;
  r.inv = TC_COUNT * 10; r.x2 = r.inv + (2 + D.nsyms) * 10; r.sl = r.x2 + D.nr * 10; r.qs = r.sl + 3 * TRRP_MAX_SLOTS * 10; r.qis = r.qs + 64 * 10;
  r.words = r.qis + 64 * 10;
  return r;
}
BPPP_DI fr rec_get(const uint32_t *p, uint32_t i) { fr r; for (int k = 0; k < 10; k++) r.n[k] = p[i * 10 + k]; return r; }
BPPP_DI void rec_put(uint32_t *p, uint32_t i, const fr &a) { for (int k = 0; k < 10; k++) p[i * 10 + k] = a.n[k]; }
BPPP_DI fr fr_shfl_up(const fr &a, int d, int width) { fr r; for (int k = 0; k < 10; k++) r.n[k] = (uint32_t)__shfl_up((int)a.n[k], d, width); return r; }

template <int G>
__global__ void __launch_bounds__(64) k_trrp_pre(TrrpDims D, uint32_t batch, const uint32_t *__restrict__ syms, const uint32_t *__restrict__ ch, uint32_t *__restrict__ rec_all) {
  __shared__ uint32_t lds_all[(64 / G) * 2 * G * 10];
  const uint32_t t = threadIdx.x % G, grp = threadIdx.x / G, m = 2 + D.nsyms;
  const uint32_t b_raw = blockIdx.x * (64 / G) + grp;
  const bool live = b_raw < batch;
  const uint32_t b = live ? b_raw : batch - 1;                 // a group past the end recomputes the last proof (it must keep up with the barriers) and stores the same values
  const TrrpRec R = trrp_rec(D);
  uint32_t *rec = rec_all + (size_t)b * R.words;
  uint32_t *sa = lds_all + (size_t)grp * 2 * G * 10, *sb = sa + G * 10;
  const uint32_t *c = ch + (size_t)b * 56;
  const fr e = fr_load(c), x = fr_load(c + 8), r0 = fr_load(c + 16), q = fr_load(c + 24), xp = fr_load(c + 32), r1 = fr_load(c + 40), tt = fr_load(c + 48);
  fr q0 = frs(q);
  if (D.flavour) q0 = fr_negr(q0);
  // ---- one inversion per proof (batchInverse semantics: 0 -> 0): local prefix products go straight to the record
  uint32_t *inv = rec + R.inv;
  const uint32_t K = (m + G - 1) / G, lo = min(m, t * K), hi = min(m, lo + K);
  fr local = fr_one();
  for (uint32_t i = lo; i < hi; i++) {
    const fr a = i == 0 ? e : i == 1 ? q0 : fr_addr(e, fr_load(syms + (size_t)(i - 2) * 8));
    rec_put(inv, i, local);
    if (!fr_is_zero(a)) local = frm(local, a);
  }
  lds_put(sa, t, local); lds_put(sb, t, local);
  __syncthreads();
  for (int d = 1; d < G; d <<= 1) {
    const fr pa = lds_get(sa, t), pb = lds_get(sb, t);
    const fr oa = (int)t - d >= 0 ? lds_get(sa, t - d) : fr_one();
    const fr ob = t + d < (uint32_t)G ? lds_get(sb, t + d) : fr_one();
    __syncthreads();
    lds_put(sa, t, frm(pa, oa)); lds_put(sb, t, frm(pb, ob));
    __syncthreads();
  }
  const fr others = frm(t ? lds_get(sa, t - 1) : fr_one(), t + 1 < (uint32_t)G ? lds_get(sb, t + 1) : fr_one());
  const fr total = lds_get(sa, G - 1);
  __syncthreads();
  if (t == 0) lds_put(sa, 0, fr_from_fe(fe_modinv<1>(fr_to_fe(total))));
  __syncthreads();
  {
    fr suf = frm(lds_get(sa, 0), others);
    for (uint32_t i = hi; i-- > lo;) {
      const fr a = i == 0 ? e : i == 1 ? q0 : fr_addr(e, fr_load(syms + (size_t)(i - 2) * 8));
      if (fr_is_zero(a)) { rec_put(inv, i, fr_zero()); continue; }
      rec_put(inv, i, frm(suf, rec_get(inv, i)));
      suf = frm(suf, a);
    }
  }
  __syncthreads();                                               // inv[0], inv[1] were written by lane 0 of the group
  __threadfence_block();
  const fr e_inv = rec_get(inv, 0), q0_inv = rec_get(inv, 1);
  // ---- tables
  const fr xx = frs(x), x3 = frm(xx, x);
  const fr t2 = frs(tt), t3 = frm(t2, tt);
  {
    fr xj = fr_pow_u32(xx, t + 1);
    const fr xg = fr_pow_u32(xx, G);
    for (uint32_t j = t; j < D.nr; j += G) { rec_put(rec + R.x2, j, xj); xj = frm(xj, xg); }
  }
  const fr t4 = frs(t2), t5 = frm(t4, tt), two_t5 = fr_dblr(t5), two_t3 = fr_dblr(t3);
  for (uint32_t s_ = t; s_ < (uint32_t)TRRP_MAX_SLOTS; s_ += G) {
    const fr v = frm(x3, fr_pow_u32(xx, s_));
    rec_put(rec + R.sl, s_, frm(t2, v));
    rec_put(rec + R.sl, TRRP_MAX_SLOTS + s_, frm(frm(two_t5, e_inv), v));
    rec_put(rec + R.sl, 2 * TRRP_MAX_SLOTS + s_, frm(two_t3, v));
  }
  {                                                              // chunk starts of k_trrp_pos: q0^(C j + 1), q0^-(C j + 1), j < 64: lane t takes j = t, t + G, ...
    const uint32_t C = (D.nlen + 63) / 64;
    const fr qc = fr_pow_u32(q0, C), qic = fr_pow_u32(q0_inv, C);
    fr a = frm(q0, fr_pow_u32(qc, t)), ai = frm(q0_inv, fr_pow_u32(qic, t));
    const fr sa_ = fr_pow_u32(qc, G), sai = fr_pow_u32(qic, G);
    for (uint32_t j = t; j < 64; j += G) { rec_put(rec + R.qs, j, a); rec_put(rec + R.qis, j, ai); a = frm(a, sa_); ai = frm(ai, sai); }
  }
  if (t == 0) {
    const fr t2e = frm(t2, e), t3xp = frm(t3, xp), t2x = frm(t2, x), rs = frm(r0, r1), t6 = frs(t3);
    rec_put(rec, TC_Q, q); rec_put(rec, TC_Q0, q0); rec_put(rec, TC_Q0I, q0_inv); rec_put(rec, TC_EI, e_inv); rec_put(rec, TC_T, tt); rec_put(rec, TC_T2, t2);
    rec_put(rec, TC_T3, t3); rec_put(rec, TC_2T5, two_t5); rec_put(rec, TC_T2E, t2e); rec_put(rec, TC_T3XP, t3xp); rec_put(rec, TC_T2ET, fr_addr(t2e, t3xp));
    rec_put(rec, TC_T2X, t2x); rec_put(rec, TC_T2NX, fr_negr(t2x)); rec_put(rec, TC_2T5X, frm(two_t5, x)); rec_put(rec, TC_NXP, fr_negr(xp));
    rec_put(rec, TC_L1, frm(rs, tt)); rec_put(rec, TC_L2, frm(rs, t2)); rec_put(rec, TC_L3, frm(rs, t3)); rec_put(rec, TC_L4, frm(r0, t4)); rec_put(rec, TC_L5, frm(rs, t6));
  }
}

// one wavefront per proof; the record (written by k_trrp_pre) is only read here
__global__ void __launch_bounds__(64, 2) k_trrp_pos(TrrpDims D, uint32_t batch, const uint32_t *__restrict__ pos_kind, const uint32_t *__restrict__ pos_range,
                                                    const uint32_t *__restrict__ pos_slot, const uint32_t *__restrict__ pos_sym, const uint32_t *__restrict__ pos_coeff,
                                                    const uint32_t *__restrict__ range_min, const uint32_t *__restrict__ range_assumed,
                                                    const uint32_t *__restrict__ pub_is_out, const uint32_t *__restrict__ pub_amount, const uint32_t *__restrict__ pub_sym,
                                                    const uint32_t *__restrict__ rec_all, uint32_t *__restrict__ out_q, uint32_t *__restrict__ out_sp,
                                                    uint32_t *__restrict__ out_norm) {
  __shared__ uint32_t part[64 * 10];
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const TrrpRec R = trrp_rec(D);
  const uint32_t *rec = rec_all + (size_t)b * R.words;
  const uint32_t C = (D.nlen + 63) / 64, plo = min(D.nlen, t * C), phi = min(D.nlen, plo + C);
  fr acc = fr_zero();
  if (plo < phi) {
    fr q2 = rec_get(rec + R.qs, t), qi2 = rec_get(rec + R.qis, t);
    for (uint32_t i = plo; i < phi; i++) {
      const uint32_t kf = pos_kind[i], kind = kf & 0xFFu;
      const fr xr = rec_get(rec + R.x2, pos_range[i]);
      fr p, ts0;
      if (kind == K_TYPING) {
        fr A = rec_get(rec, (kf & F_IO) ? TC_T2NX : TC_T2X);
        if (!(kf & F_IA)) A = fr_addr(A, frm(rec_get(rec, TC_T3XP), xr));
        p = fr_addr(rec_get(rec, TC_T2ET), frm(qi2, A));
        ts0 = frm(q2, frs(p));
      } else {
        const uint32_t slot = pos_slot[i], sy = pos_sym[i];
        const fr sl0 = rec_get(rec + R.sl, slot);
        fr A = fr_addr(sl0, frm(rec_get(rec, TC_T3), frm(xr, fr_load(pos_coeff + (size_t)i * 8))));
        if (kind == K_INLINE && sy != NO_SYM) {
          const fr si = rec_get(rec + R.inv, 2 + sy);             // "if s == 0 then 0" is tested on the INVERTED value (:205)
          if (!fr_is_zero(si)) A = fr_addr(A, frm(frm(rec_get(rec, TC_T2), sl0), fr_subr(rec_get(rec, TC_EI), si)));
        }
        p = fr_addr(rec_get(rec, TC_T2E), frm(qi2, A));
        ts0 = fr_addr(frm(q2, fr_addr(frs(p), rec_get(rec, TC_2T5))), rec_get(rec + R.sl, TRRP_MAX_SLOTS + slot));
      }
      fr_store(out_norm + ((size_t)b * D.nlen + i) * 8, p);
      acc = fr_addr(acc, ts0);
      q2 = frm(q2, rec_get(rec, TC_Q0)); qi2 = frm(qi2, rec_get(rec, TC_Q0I));
    }
  }
  // z (:254): -2 t^5 sum_j min_j x^(2(j+1))  -  [typed] 2 t^5 x pubSum
  for (uint32_t j = t; j < D.nr; j += 64)
    if (!range_assumed[j]) acc = fr_subr(acc, frm(rec_get(rec, TC_2T5), frm(fr_load(range_min + (size_t)j * 8), rec_get(rec + R.x2, j))));
  if (D.has_types)
    for (uint32_t j = t; j < D.npub; j += 64) {
      const fr term = frm(rec_get(rec, TC_2T5X), frm(fr_load(pub_amount + (size_t)j * 8), rec_get(rec + R.inv, 2 + pub_sym[j])));
      acc = pub_is_out[j] ? fr_addr(acc, term) : fr_subr(acc, term);
    }
  lds_put(part, t, acc);
  __syncthreads();
  for (int d = 32; d >= 1; d >>= 1) {
    if ((int)t < d) lds_put(part, t, fr_addr(lds_get(part, t), lds_get(part, t + d)));
    __syncthreads();
  }
  if (t == 0) { fr_store(out_sp + (size_t)b * 8, lds_get(part, 0)); fr_store(out_q + (size_t)b * 8, rec_get(rec, TC_Q)); }
}

// one lane per output: the llen linear weights (makeBpCoeffs :391-396 over makeSharedCoeffs :213-216), then the 4 + nr initCom scalars
__global__ void __launch_bounds__(256) k_trrp_lin(TrrpDims D, uint32_t batch, const uint32_t *__restrict__ range_assumed, const uint32_t *__restrict__ cs_slot,
                                                  const uint32_t *__restrict__ cs_sym, const uint32_t *__restrict__ rec_all, uint32_t *__restrict__ out_cs,
                                                  uint32_t *__restrict__ out_init) {
  const uint32_t per = D.llen + 4 + D.nr;
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * per) return;
  const uint32_t b = (uint32_t)(g / per), j = (uint32_t)(g % per);
  const TrrpRec R = trrp_rec(D);
  const uint32_t *rec = rec_all + (size_t)b * R.words;
  if (j < D.llen) {
    fr v;
    if (j == 0) v = D.has_types ? rec_get(rec, TC_NXP) : fr_zero();
    else if (j < 6) v = rec_get(rec, TC_L1 + (j - 1));
    else v = frm(rec_get(rec + R.sl, 2 * TRRP_MAX_SLOTS + cs_slot[j - 6]), fr_subr(rec_get(rec, TC_EI), rec_get(rec + R.inv, 2 + cs_sym[j - 6])));
    fr_store(out_cs + ((size_t)b * D.llen + j) * 8, v);
    return;
  }
  const uint32_t r = j - D.llen, ninit = 4 + D.nr;             // commitment order blCom : rCom : dmCom : mCom : nComs (openWith of TranscriptTRRP, :293-297)
  fr v;
  if (r == 0) v = fr_one();
  else if (r == 1) v = rec_get(rec, TC_T3);
  else if (r == 2) v = rec_get(rec, TC_T2);
  else if (r == 3) v = rec_get(rec, TC_T);
  else {
    const uint32_t rr = r - 4;
    fr ic = range_assumed[rr] ? fr_zero() : rec_get(rec + R.x2, rr);                          // inputCoeffs (:325-328)
    if (D.has_types) ic = fr_addr(ic, fr_pow_u32(rec_get(rec, TC_Q0), rr + 1));
    v = frm(rec_get(rec, TC_2T5), ic);
  }
  fr_store(out_init + ((size_t)b * ninit + r) * 8, v);
}

}  // namespace bppp

using namespace bppp;


extern "C" {

void bppp_trrp_destroy(bppp_trrp *o) {
  if (!o) return;
  bppp_ctx *ctx = o->ctx;                  // kept alive by this handle's reference even after bppp_ctx_destroy
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (uint32_t *p : {o->pos_kind, o->pos_range, o->pos_slot, o->pos_sym, o->pos_coeff, o->range_min, o->range_assumed, o->syms, o->cs_slot, o->cs_sym, o->pub_is_out,
                      o->pub_amount, o->pub_sym})
    if (p) hipFree(p);
  delete o;
  ctx_release(ctx);
}

int bppp_trrp_create(bppp_ctx *ctx, int flavour, int has_types, size_t nlen, size_t llen, size_t nranges, const uint32_t *pos_kind, const uint32_t *pos_range,
                     const uint32_t *pos_slot, const uint32_t *pos_sym, const uint64_t *pos_coeff, const uint64_t *range_min, const uint32_t *range_assumed,
                     size_t nsyms, const uint64_t *syms, const uint32_t *cs_slot, const uint32_t *cs_sym, size_t npub, const uint32_t *pub_is_out,
                     const uint64_t *pub_amount, const uint32_t *pub_sym, bppp_trrp **out) {
  if (!ctx || !out || ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!nlen || llen < 6 || !nranges || nlen >= (1u << 24) || llen >= (1u << 24) || nranges >= (1u << 20) || nsyms + 2 > 1024 || npub >= (1u << 20) ||
      !pos_kind || !pos_range || !pos_slot || !pos_sym || !pos_coeff || !range_min || !range_assumed || (nsyms && !syms) || (llen > 6 && (!cs_slot || !cs_sym)) ||
      (npub && (!pub_is_out || !pub_amount || !pub_sym)))
    return fail(ctx, BPPP_ERR_ARG, "trrp_create: bad arguments (at most 1022 distinct reciprocal symbols)");
  for (size_t i = 0; i < nlen; i++)
    if (pos_range[i] >= nranges || (pos_kind[i] & 0xFFu) > 2 || (pos_sym[i] != 0xFFFFFFFFu && pos_sym[i] >= nsyms) || pos_slot[i] >= (uint32_t)TRRP_MAX_SLOTS)
      return fail(ctx, BPPP_ERR_ARG, "trrp_create: position table out of range");
  for (size_t j = 0; j + 6 < llen; j++)
    if (cs_sym[j] >= nsyms || cs_slot[j] >= (uint32_t)TRRP_MAX_SLOTS) return fail(ctx, BPPP_ERR_ARG, "trrp_create: shared-coefficient table out of range");
  for (size_t j = 0; j < npub; j++)
    if (pub_sym[j] >= nsyms) return fail(ctx, BPPP_ERR_ARG, "trrp_create: public-amount table out of range");
  hipSetDevice(ctx->device);
  bppp_trrp *o = new bppp_trrp();
  memset(o, 0, sizeof *o);
  o->ctx = ctx; ctx_retain(ctx);
  o->D = TrrpDims{(uint32_t)nlen, (uint32_t)llen, (uint32_t)nranges, (uint32_t)nsyms, (uint32_t)npub, has_types ? 1u : 0u, flavour ? 1u : 0u};
  bool bad = false;
  auto up = [&](uint32_t **dst, const void *src, size_t bytes) {
    if (!bytes) bytes = 4, src = nullptr;
    if (hipMalloc(dst, bytes) != hipSuccess) { bad = true; *dst = nullptr; return; }
    if (src && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) bad = true;
  };
  up(&o->pos_kind, pos_kind, nlen * 4); up(&o->pos_range, pos_range, nlen * 4); up(&o->pos_slot, pos_slot, nlen * 4); up(&o->pos_sym, pos_sym, nlen * 4);
  up(&o->pos_coeff, pos_coeff, nlen * 32); up(&o->range_min, range_min, nranges * 32); up(&o->range_assumed, range_assumed, nranges * 4);
  up(&o->syms, syms, nsyms * 32); up(&o->cs_slot, cs_slot, (llen - 6) * 4); up(&o->cs_sym, cs_sym, (llen - 6) * 4);
  up(&o->pub_is_out, pub_is_out, npub * 4); up(&o->pub_amount, pub_amount, npub * 32); up(&o->pub_sym, pub_sym, npub * 4);
  if (bad) { bppp_trrp_destroy(o); return fail(ctx, BPPP_ERR_HIP, "trrp_create: device allocation or upload failed"); }
  *out = o;
  return BPPP_OK;
}

int bppp_trrp_public_device(bppp_trrp *o, size_t batch, const void *d_challenges, void *d_q, void *d_sp, void *d_pub_norm, void *d_pub_lin_c,
                            void *d_init_scalars) {
  if (!o) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!batch) return BPPP_OK;
  if (!d_challenges || !d_q || !d_sp || !d_pub_norm || !d_pub_lin_c || !d_init_scalars || batch >= (1u << 24)) return fail(ctx, BPPP_ERR_ARG, "trrp_public: bad arguments");
  hipSetDevice(ctx->device);
  // batches that fill the chip: the three-kernel split (small live sets, the per-proof start-up once); its record lives in the context's scratch
  static const int force_split = [] { const char *e = getenv("BPPP_TRRP_SPLIT"); return e ? atoi(e) : -1; }();
  const bool split = force_split >= 0 ? force_split != 0 : batch > 1024;
  if (split) {
    const TrrpRec R = trrp_rec(o->D);
    int rc = ensure_scratch(ctx, (size_t)batch * R.words * 4 + 256); if (rc) return rc;
    uint32_t *rec = (uint32_t *)ctx->ws2;
    hipStream_t st = ctx->stream;
    k_trrp_pre<16><<<dim3((unsigned)((batch + 3) / 4)), dim3(64), 0, st>>>(o->D, (uint32_t)batch, o->syms, (const uint32_t *)d_challenges, rec);
    k_trrp_pos<<<dim3((unsigned)batch), dim3(64), 0, st>>>(o->D, (uint32_t)batch, o->pos_kind, o->pos_range, o->pos_slot, o->pos_sym, o->pos_coeff, o->range_min,
                                                           o->range_assumed, o->pub_is_out, o->pub_amount, o->pub_sym, rec, (uint32_t *)d_q, (uint32_t *)d_sp,
                                                           (uint32_t *)d_pub_norm);
    const uint64_t nl = (uint64_t)batch * (o->D.llen + 4 + o->D.nr);
    k_trrp_lin<<<dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st>>>(o->D, (uint32_t)batch, o->range_assumed, o->cs_slot, o->cs_sym, rec, (uint32_t *)d_pub_lin_c,
                                                                       (uint32_t *)d_init_scalars);
    BPPP_HIP(ctx, hipGetLastError());
    return BPPP_OK;
  }
  // lanes per proof: 32 (two proofs per wavefront) unless the LDS of the proofs of a wavefront would not fit a workgroup; a batch that
  // cannot give every SIMD a wavefront anyway (<= 1024 proofs) takes a whole wavefront per proof: the kernel is then one proof's
  // dependency chain, which 64 lanes walk in fewer steps
  static const int forced_g = [] { const char *e = getenv("BPPP_TRRP_G"); const int v = e ? atoi(e) : 0; return (v == 16 || v == 32 || v == 64) ? v : 0; }();
  int G = forced_g ? forced_g : (batch <= 1024 ? 64 : 32);
  auto words = [&](int g) { return ((size_t)(2 + o->D.nsyms) + 2 * (size_t)g + o->D.nr + 3 * TRRP_MAX_SLOTS) * 10; };      // fr26: ten words per element
  while (G < 64 && words(G) * 4 * (64 / G) > 64 * 1024) G <<= 1;
  const size_t wpp = words(G), lds = wpp * 4 * (64 / G);
  if (lds > 160 * 1024) return fail(ctx, BPPP_ERR_ARG, "trrp_public: too many ranges for one workgroup's LDS");
  const unsigned grid = (unsigned)((batch + 64 / G - 1) / (64 / G));
#define TRRP_LAUNCH(GG)                                                                                                                        \
  k_trrp_public<GG><<<dim3(grid), dim3(64), lds, ctx->stream>>>(o->D, (uint32_t)batch, (uint32_t)wpp, o->pos_kind, o->pos_range, o->pos_slot, o->pos_sym, o->pos_coeff, \
                                                               o->range_min, o->range_assumed, o->syms, o->cs_slot, o->cs_sym, o->pub_is_out, o->pub_amount, o->pub_sym, \
                                                               (const uint32_t *)d_challenges, (uint32_t *)d_q, (uint32_t *)d_sp, (uint32_t *)d_pub_norm,          \
                                                               (uint32_t *)d_pub_lin_c, (uint32_t *)d_init_scalars)
  if (G == 64) {
    if (lds > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_trrp_public<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TRRP_LAUNCH(64);
  } else if (G == 32) TRRP_LAUNCH(32);
  else TRRP_LAUNCH(16);
#undef TRRP_LAUNCH
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

}  // extern "C"
