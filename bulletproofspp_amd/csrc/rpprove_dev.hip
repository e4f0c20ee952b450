// rpprove_dev.hip — the per-proof field algebra and the Fiat-Shamir transcript of the batch range-proof prover, on the device.
//
// proveTRRPM (src/RangeProof/TypedReciprocal.hs:399-446) does O(nrmLen) field work per phase: reciprocals of (e + digit)
// (makePhase2s :185-205), the six error terms (makeErrorTerms :226-243), the public constants (makePublicConsts :246-274) and the
// linear combination of five witnesses; the prover's randomness is hashToScalar prefix . show (app/Main.hs:83-87) and every
// challenge is shaOracle over the whole transcript (app/Main.hs:75-80, src/ZKP.hs:96-101).  With the group operations already on
// the GPU that work, done on 16 host threads, was two thirds of a batch's wall time.  Here:
//   k_rpp_draws         every random scalar of every proof: one SHA-256 per lane
//   k_rpp_rows_dm_m     the scalar rows of dmWit / mWit (blindWitness, src/RangeProof/Internal.hs:130-139) straight into the MSM input
//   k_rpp_phase2        one wavefront per proof: ONE batched inversion (e, r0, e + s for every digit value s, e + type_i), the
//                       reciprocals r, the coefficients c, err7 (:209-211), the row of rWit (blindErrWitness, Internal.hs:142-149)
//   k_rpp_phase3        one wavefront per proof: shared coefficients (:213-216), input coefficients (:325-328), the six error terms,
//                       blindBlindingTerm (Internal.hs:154-196) on one lane, the row of the blinding witness
//   k_rpp_combine       wit = pub + blWit + t mWit + t^2 dmWit + t^3 rWit + 2 t^5 nWitSum (:441-443), written where the lockstep
//                       argument starts from; pub and the argument's linear weights come from k_trrp_public (the verifier's kernel
//                       computes exactly makePublicConsts and makeBpCoeffs)
//   k_rpp_text_prepend / k_rpp_hash   the transcript text of every proof (newest commitment first) grows at its FRONT; a challenge
//                       is SHA-256 (header <> text from the current start)
// rpp_device_prove has two flows over these kernels: with the handle's comb table in place (large batches) the whole proof is ONE
// stream of kernels — commitments stay in HBM, every oracle call reads its points where they lie, the headers of all oracle calls
// are uploaded up front, the argument runs in fixed-basis mode (csrc/nlb.hip) — and the host waits once, at the end; otherwise
// (small batches) commitments go through the registered basis and the point-folding argument with a host round trip per step.
// Fr arithmetic in 8x32 limbs (fe.hip.h); multiply / square as real functions (instruction-cache footprint, as in trrp.hip).
#include <string.h>
#include <thread>
#include <string>
#include <vector>
#include "fe.hip.h"
#include "modinv.hip.h"
#include "rp_internal.hpp"
#include "rphash.hip.h"
#include "comb.hpp"
#include "rpprove_dev.hpp"
#include "rpp_transcript.hpp"
#include "ipb.hpp"
#include "trrp.hpp"

namespace bppp {

__device__ __noinline__ fe pfm(fe a, fe b) { return fe_mul<1>(a, b); }
__device__ __noinline__ fe pfs(fe a) { return fe_sqr<1>(a); }
BPPP_DI fe pf_pow(fe base, uint32_t e) {
  fe acc = fe_one();
  while (e) { if (e & 1u) acc = pfm(acc, base); base = pfs(base); e >>= 1; }
  return acc;
}
BPPP_DI fe pf_small(uint32_t v) { fe r = fe_zero(); r.v[0] = v; return r; }
BPPP_DI fe lget(const uint32_t *p, uint32_t i) { fe r; for (int k = 0; k < 8; k++) r.v[k] = p[i * 8 + k]; return r; }
BPPP_DI void lput(uint32_t *p, uint32_t i, const fe &a) { for (int k = 0; k < 8; k++) p[i * 8 + k] = a.v[k]; }

// (k_rpp_draws: csrc/rpp_transcript.hip)


// ------------------------------------------------------------------------------------------------ phase 1 rows
// dmWit = blindWitness 3 2 msShared ds, mWit = blindWitness 3 1 [] msInline (TypedReciprocal.hs:408-410; Internal.hs:130-139):
// seven blinding entries [a b c d 0 e 0] / [a b c d e 0 0], the first is the scalar on g.  Rows are [sc | lin (llen) | nrm (nlen)].
__global__ void __launch_bounds__(256) k_rpp_rows_dm_m(PDims D, const uint32_t *__restrict__ pos_kind, const uint32_t *__restrict__ pos_range,
                                                       const uint32_t *__restrict__ in_sc, const uint32_t *__restrict__ dig, const uint32_t *__restrict__ mul,
                                                       const uint32_t *__restrict__ mss, const uint32_t *__restrict__ rnd, uint32_t batch,
                                                       uint32_t *__restrict__ rows) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * D.T) return;
  const uint32_t b = (uint32_t)(g / D.T), j = (uint32_t)(g % D.T);
  const uint32_t *r = rnd + (size_t)b * D.nd * 8;
  fe dm = fe_zero(), m = fe_zero();
  if (j == 0) { dm = fe_load(r); m = fe_load(r + 5 * 8); }
  else if (j <= D.llen) {
    const uint32_t i = j - 1;                  // linear index
    // dm: lin[0..5] = [b c d 0 e 0] = draws 1 2 3 - 4 -;  m: lin[0..5] = [b c d e 0 0] = draws 6 7 8 9 - -
    if (i < 3) { dm = fe_load(r + (1 + i) * 8); m = fe_load(r + (6 + i) * 8); }
    else if (i == 3) m = fe_load(r + 9 * 8);
    else if (i == 4) dm = fe_load(r + 4 * 8);
    else if (i >= 6) dm = pf_small(mss[(size_t)b * (D.llen - 6) + (i - 6)]);
  } else {
    const uint32_t i = j - 1 - D.llen, kind = pos_kind[i] & 0xFFu;
    if (kind == K_TYPING) dm = fe_load(in_sc + ((size_t)b * D.nr + pos_range[i]) * 24 + 8);       // getDsMs: the type (TypedReciprocal.hs:74-80)
    else { dm = pf_small(dig[(size_t)b * D.nlen + i]); m = pf_small(mul[(size_t)b * D.nlen + i]); }
  }
  fe_store(rows + (((size_t)b * 2) * D.T + j) * 8, dm);
  fe_store(rows + (((size_t)b * 2 + 1) * D.T + j) * 8, m);
}

// ------------------------------------------------------------------------------------------------ one inversion per wavefront
// vals[0 .. m) in LDS are replaced by their inverses (0 -> 0, batchInverse): every lane owns a contiguous chunk, prefix / suffix
// products over the lanes, one safegcd inversion on lane 0.  pre: m entries of scratch; sa, sb: 64 entries each.
BPPP_DI void wave_batch_invert(uint32_t *vals, uint32_t *pre, uint32_t *sa, uint32_t *sb, uint32_t m, uint32_t t) {
  const uint32_t K = (m + 63) / 64, lo = min(m, t * K), hi = min(m, lo + K);
  fe local = fe_one();
  for (uint32_t i = lo; i < hi; i++) {
    const fe a = lget(vals, i);
    lput(pre, i, local);
    if (!fe_is_zero(a)) local = pfm(local, a);
  }
  lput(sa, t, local); lput(sb, t, local);
  __syncthreads();
  for (int d = 1; d < 64; d <<= 1) {
    const fe pa = lget(sa, t), pb = lget(sb, t);
    const fe oa = (int)t - d >= 0 ? lget(sa, t - d) : fe_one();
    const fe ob = t + d < 64 ? lget(sb, t + d) : fe_one();
    __syncthreads();
    lput(sa, t, pfm(pa, oa)); lput(sb, t, pfm(pb, ob));
    __syncthreads();
  }
  const fe others = pfm(t ? lget(sa, t - 1) : fe_one(), t + 1 < 64 ? lget(sb, t + 1) : fe_one());
  const fe total = lget(sa, 63);
  __syncthreads();
  if (t == 0) lput(sa, 0, fe_modinv<1>(total));
  __syncthreads();
  fe suf = pfm(lget(sa, 0), others);
  for (uint32_t i = hi; i-- > lo;) {
    const fe a = lget(vals, i);
    if (fe_is_zero(a)) continue;                 // stays 0
    lput(vals, i, pfm(suf, lget(pre, i)));
    suf = pfm(suf, a);
  }
  __syncthreads();
}

static constexpr uint32_t INV_E = 0, INV_R0 = 1, INV_DIG = 2;     // layout of the inverse table: e, r0, e + s (s < D.maxb), e + type_i

// ------------------------------------------------------------------------------------------------ phase 2
__global__ void __launch_bounds__(64) k_rpp_phase2(PDims D, TrrpDims TD, const uint32_t *__restrict__ pos_kind, const uint32_t *__restrict__ pos_range,
                                                   const uint32_t *__restrict__ pos_slot, const uint32_t *__restrict__ pos_sym, const uint32_t *__restrict__ syms,
                                                   const uint32_t *__restrict__ in_sc, const uint32_t *__restrict__ dig, const uint32_t *__restrict__ rnd,
                                                   const uint32_t *__restrict__ ch, uint32_t *__restrict__ row_r, uint32_t *__restrict__ ccbuf,
                                                   uint32_t *__restrict__ invtab) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = threadIdx.x, b = blockIdx.x, m = INV_DIG + D.maxb + D.nr;
  uint32_t *vals = lds, *pre = vals + (size_t)m * 8, *sa = pre + (size_t)m * 8, *sb = sa + 64 * 8, *bm = sb + 64 * 8;   // bm: [TRRP_MAX_SLOTS]
  const uint32_t *c = ch + (size_t)b * 56;
  const fe e = fe_load(c), x = fe_load(c + 8), r0 = fe_load(c + 16);
  for (uint32_t i = t; i < m; i += 64) {
    fe a;
    if (i == INV_E) a = e;
    else if (i == INV_R0) a = r0;
    else if (i < INV_DIG + D.maxb) a = fe_add<1>(e, pf_small(i - INV_DIG));
    else a = fe_add<1>(e, fe_load(in_sc + ((size_t)b * D.nr + (i - INV_DIG - D.maxb)) * 24 + 8));      // e + type of input i
    lput(vals, i, a);
  }
  __syncthreads();
  wave_batch_invert(vals, pre, sa, sb, m, t);
  const fe xx = pfs(x), x3 = pfm(xx, x);
  if (t < TRRP_MAX_SLOTS) lput(bm, t, pfm(x3, pf_pow(xx, t)));                                      // makeBaseMap (:349)
  __syncthreads();
  const fe e_inv = lget(vals, INV_E), r0_inv = lget(vals, INV_R0);
  uint32_t *rr_out = row_r + ((size_t)b * D.T + 1 + D.llen) * 8;
  fe acc = fe_zero();
  const uint32_t C = (D.nlen + 63) / 64, plo = min(D.nlen, t * C), phi = min(D.nlen, plo + C);
  for (uint32_t i = plo; i < phi; i++) {
    const uint32_t kf = pos_kind[i], kind = kf & 0xFFu;
    fe rr, cc = fe_zero();
    if (kind == K_TYPING) {
      const uint32_t rg = pos_range[i];
      rr = pfm(fe_load(in_sc + ((size_t)b * D.nr + rg) * 24), lget(vals, INV_DIG + D.maxb + rg));     // v / (e + t)
    } else {
      rr = lget(vals, INV_DIG + dig[(size_t)b * D.nlen + i]);                                      // 1 / (e + d)
      const uint32_t sy = pos_sym[i];
      if (kind == K_INLINE && sy != NO_SYM) {
        const fe si = lget(vals, INV_DIG + syms[(size_t)sy * 8]);                                  // symbols are small integers (1 .. base - 1)
        if (!fe_is_zero(si)) cc = pfm(lget(bm, pos_slot[i]), fe_sub<1>(e_inv, si));
      }
    }
    fe_store(rr_out + (size_t)i * 8, rr);
    fe_store(ccbuf + ((size_t)b * D.nlen + i) * 8, cc);
    acc = fe_add<1>(acc, pfm(rr, cc));
  }
  lput(sa, t, acc);
  __syncthreads();
  for (int d = 32; d >= 1; d >>= 1) {
    if ((int)t < d) lput(sa, t, fe_add<1>(lget(sa, t), lget(sa, t + d)));
    __syncthreads();
  }
  // the row of rWit = blindErrWitness 3 [err7] [] rs (Internal.hs:142-149): blinding entries [a b c 0 d err7 0]
  const uint32_t *r = rnd + (size_t)b * D.nd * 8;
  uint32_t *row = row_r + (size_t)b * D.T * 8;
  if (t == 0) {
    const fe err7 = pfm(r0_inv, fe_neg<1>(fe_dbl<1>(lget(sa, 0))));                                 // err7Term (:209-211)
    fe_store(row, fe_load(r + 10 * 8));
    fe_store(row + 1 * 8, fe_load(r + 11 * 8)); fe_store(row + 2 * 8, fe_load(r + 12 * 8)); fe_store(row + 3 * 8, fe_zero());
    fe_store(row + 4 * 8, fe_load(r + 13 * 8)); fe_store(row + 5 * 8, err7); fe_store(row + 6 * 8, fe_zero());
  }
  for (uint32_t j = 6 + t; j < D.llen; j += 64) fe_store(row + (size_t)(1 + j) * 8, fe_zero());
  for (uint32_t i = t; i < INV_DIG + D.maxb; i += 64) fe_store(invtab + ((size_t)b * (INV_DIG + D.maxb) + i) * 8, lget(vals, i));
}

// ------------------------------------------------------------------------------------------------ phase 3
// aux[b] = [ns_sc, ns_ty, ns_bl] for the combination
__global__ void __launch_bounds__(64) k_rpp_phase3(PDims D, TrrpDims TD, const uint32_t *__restrict__ pos_kind, const uint32_t *__restrict__ pos_range,
                                                   const uint32_t *__restrict__ pos_slot, const uint32_t *__restrict__ pos_coeff,
                                                   const uint32_t *__restrict__ range_assumed, const uint32_t *__restrict__ syms,
                                                   const uint32_t *__restrict__ cs_slot, const uint32_t *__restrict__ cs_sym, const uint32_t *__restrict__ in_sc,
                                                   const uint32_t *__restrict__ dig, const uint32_t *__restrict__ mul, const uint32_t *__restrict__ rnd,
                                                   const uint32_t *__restrict__ ch, const uint32_t *__restrict__ rows_dm_m, const uint32_t *__restrict__ row_r,
                                                   const uint32_t *__restrict__ ccbuf, const uint32_t *__restrict__ invtab, uint32_t *__restrict__ row_bl,
                                                   uint32_t *__restrict__ aux) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = threadIdx.x, b = blockIdx.x;
  uint32_t *sa = lds, *x2 = sa + 64 * 8 * 6, *bm = x2 + (size_t)D.nr * 8, *sh = bm + TRRP_MAX_SLOTS * 8;      // sa: 6 reduction rows; sh: [4] inverses etc.
  const uint32_t *c = ch + (size_t)b * 56;
  const fe e = fe_load(c), x = fe_load(c + 8), q = fe_load(c + 24), xp = fe_load(c + 32), r1 = fe_load(c + 40);
  const uint32_t *it = invtab + (size_t)b * (INV_DIG + D.maxb) * 8;
  const fe e_inv = fe_load(it + INV_E * 8), r0_inv = fe_load(it + INV_R0 * 8);
  fe q0 = pfs(q);                                     // qPowers': powers' (q^2) of the NL norm (NormArgument.hs:148),
  if (TD.flavour) q0 = fe_neg<1>(q0);                 // powers' (-q^2) of the IP one (InnerProductArgument.hs:231)
  if (t == 0) lput(sh, 0, fe_modinv<1>(r1));
  const fe xx = pfs(x), x3 = pfm(xx, x);
  for (uint32_t j = t; j < D.nr; j += 64) lput(x2, j, pf_pow(xx, j + 1));
  if (t < TRRP_MAX_SLOTS) lput(bm, t, pfm(x3, pf_pow(xx, t)));
  __syncthreads();
  const fe r1_inv = lget(sh, 0);
  const uint32_t *r = rnd + (size_t)b * D.nd * 8;
  const uint32_t *bls_lin = r + 14 * 8, *bls_nrm = bls_lin + (size_t)(D.llen - 5) * 8;
  // sum_j sharedCs_j * blsMs_j (makeSharedCoeffs :213-216; blsMs = tail blsLin) and the input-coefficient sums (:325-328, :431-432)
  fe s3 = fe_zero(), nsc = fe_zero(), nty = fe_zero(), nbl = fe_zero();
  for (uint32_t j = t; j + 6 < D.llen; j += 64) {
    const fe si = fe_load(it + (size_t)(INV_DIG + syms[(size_t)cs_sym[j] * 8]) * 8);
    const fe cs = pfm(lget(bm, cs_slot[j]), fe_sub<1>(e_inv, si));
    s3 = fe_add<1>(s3, pfm(cs, fe_load(bls_lin + (size_t)(1 + j) * 8)));
  }
  for (uint32_t i = t; i < D.nr; i += 64) {
    fe ic = range_assumed[i] ? fe_zero() : lget(x2, i);
    if (D.has_types) ic = fe_add<1>(ic, pf_pow(q0, i + 1));
    const uint32_t *in = in_sc + ((size_t)b * D.nr + i) * 24;
    nsc = fe_add<1>(nsc, pfm(ic, fe_load(in))); nty = fe_add<1>(nty, pfm(ic, fe_load(in + 8))); nbl = fe_add<1>(nbl, pfm(ic, fe_load(in + 16)));
  }
  // makeErrorTerms (:226-243) over this lane's chunk of positions
  fe t0 = fe_zero(), t1 = fe_zero(), t2 = fe_zero(), t3 = fe_dbl<1>(s3), t4 = fe_zero(), t5 = fe_zero();
  {
    const uint32_t C = (D.nlen + 63) / 64, plo = min(D.nlen, t * C), phi = min(D.nlen, plo + C);
    fe q2 = pf_pow(q0, plo + 1);
    const uint32_t *rr_in = row_r + ((size_t)b * D.T + 1 + D.llen) * 8;
    for (uint32_t i = plo; i < phi; i++) {
      const uint32_t kf = pos_kind[i], kind = kf & 0xFFu, rg = pos_range[i];
      const bool is_t = kind == K_TYPING;
      fe d, mm, u, v;
      if (is_t) {
        d = fe_load(in_sc + ((size_t)b * D.nr + rg) * 24 + 8); mm = fe_zero();
        u = (kf & F_IA) ? fe_zero() : lget(x2, rg);
        v = (kf & F_IO) ? fe_neg<1>(x) : x;
      } else {
        d = pf_small(dig[(size_t)b * D.nlen + i]); mm = pf_small(mul[(size_t)b * D.nlen + i]);
        u = pfm(lget(x2, rg), fe_load(pos_coeff + (size_t)i * 8));
        v = lget(bm, pos_slot[i]);
      }
      const fe rr = fe_load(rr_in + (size_t)i * 8), cc = fe_load(ccbuf + ((size_t)b * D.nlen + i) * 8), bl = fe_load(bls_nrm + (size_t)i * 8);
      const fe rC = is_t ? pfm(xp, fe_add<1>(u, q2)) : u;
      const fe dC = fe_add<1>(v, pfm(q2, e));
      const fe q2d = pfm(q2, d), q2r = pfm(q2, rr), q2d_dC = fe_add<1>(q2d, dC), q2r_rC = fe_add<1>(q2r, rC), q2bl = pfm(q2, bl), q2m = pfm(q2, mm);
      t0 = fe_add<1>(t0, pfm(q2bl, bl));
      t1 = fe_add<1>(t1, fe_dbl<1>(pfm(q2m, bl)));
      t2 = fe_add<1>(t2, fe_add<1>(pfm(q2m, mm), fe_dbl<1>(pfm(bl, q2d_dC))));
      t3 = fe_add<1>(t3, fe_dbl<1>(fe_add<1>(pfm(bl, q2r_rC), pfm(mm, q2d_dC))));
      t4 = fe_add<1>(t4, fe_add<1>(fe_add<1>(pfm(q2d, d), fe_dbl<1>(pfm(d, dC))), fe_dbl<1>(fe_add<1>(pfm(bl, cc), pfm(mm, q2r_rC)))));
      t5 = fe_add<1>(t5, fe_add<1>(fe_add<1>(pfm(q2r, rr), fe_dbl<1>(pfm(rr, rC))), fe_dbl<1>(pfm(cc, d))));
      q2 = pfm(q2, q0);
    }
  }
  // reductions over the wavefront: six error totals, then the three input sums (the rows of `sa` are reused)
  auto reduce6 = [&](fe &a0, fe &a1, fe &a2, fe &a3, fe &a4, fe &a5) {
    lput(sa, t, a0); lput(sa + 64 * 8, t, a1); lput(sa + 2 * 64 * 8, t, a2); lput(sa + 3 * 64 * 8, t, a3); lput(sa + 4 * 64 * 8, t, a4); lput(sa + 5 * 64 * 8, t, a5);
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
      if ((int)t < d)
        for (int k = 0; k < 6; k++) lput(sa + k * 64 * 8, t, fe_add<1>(lget(sa + k * 64 * 8, t), lget(sa + k * 64 * 8, t + d)));
      __syncthreads();
    }
    a0 = lget(sa, 0); a1 = lget(sa + 64 * 8, 0); a2 = lget(sa + 2 * 64 * 8, 0); a3 = lget(sa + 3 * 64 * 8, 0); a4 = lget(sa + 4 * 64 * 8, 0); a5 = lget(sa + 5 * 64 * 8, 0);
    __syncthreads();
  };
  // t3 already carries 2 * s3 of THIS lane only: s3 is a per-lane partial, so its double was added once per lane — as a sum it is 2 * sum s3: correct
  reduce6(t0, t1, t2, t3, t4, t5);
  fe z0 = fe_zero(), z1 = fe_zero(), z2 = fe_zero();
  reduce6(nsc, nty, nbl, z0, z1, z2);
  uint32_t *row = row_bl + (size_t)b * D.T * 8;
  if (t == 0) {
    fe_store(aux + (size_t)b * 24, nsc); fe_store(aux + (size_t)b * 24 + 8, nty); fe_store(aux + (size_t)b * 24 + 16, nbl);
    // blindBlindingTerm for [mWit, dmWit, rWit] (Internal.hs:154-196), n = 3
    const fe tC = D.has_types ? xp : fe_zero();
    const fe blT = fe_load(bls_lin);
    const fe rs_inv = pfm(r0_inv, r1_inv);
    const fe errs[6] = {t0, t1, t2, t3, t4, t5};
    fe diag[10];
    for (int k = 0; k < 10; k++) diag[k] = fe_zero();
    // table row 0: errs1 = negate ([errs0 - tC blT] ++ rs_inv * errs[1..]) with a zero inserted at position 5
    {
      fe e1[6];
      e1[0] = fe_neg<1>(fe_sub<1>(errs[0], pfm(tC, blT)));
      for (int j = 1; j < 6; j++) e1[j] = fe_neg<1>(pfm(rs_inv, errs[j]));
      for (int j = 0; j < 6; j++) { const int col = j < 5 ? j : 6; diag[0 + col] = fe_add<1>(diag[0 + col], e1[j]); }
    }
    // table rows 1..3: the witnesses m, dm, r: [sc, lin0, -lin1 .. -lin5], addConsts, scaleErrs r1^-1 on entry 4, zero inserted at 5
    for (int a = 0; a < 3; a++) {
      const uint32_t *w = a == 0 ? rows_dm_m + (((size_t)b * 2 + 1) * D.T) * 8 : a == 1 ? rows_dm_m + (((size_t)b * 2) * D.T) * 8 : row_r + (size_t)b * D.T * 8;
      fe rw[7];
      rw[0] = fe_load(w);
      for (int j = 0; j < 6; j++) rw[1 + j] = (a == 2 && j >= 4) ? fe_zero() : fe_load(w + (size_t)(1 + j) * 8);   // the error witness keeps n + 1 = 4 entries
      for (int j = 2; j < 7; j++) rw[j] = fe_neg<1>(rw[j]);
      fe r6[6];
      r6[0] = fe_add<1>(pfm(rs_inv, rw[0]), pfm(pfm(rs_inv, tC), rw[1]));
      for (int j = 1; j < 6; j++) r6[j] = rw[j + 1];
      r6[4] = pfm(r1_inv, r6[4]);
      for (int j = 0; j < 6; j++) { const int col = j < 5 ? j : 6; diag[1 + a + col] = fe_add<1>(diag[1 + a + col], r6[j]); }
    }
    fe be[6];
    be[0] = diag[0]; be[1] = diag[1]; be[2] = diag[2]; be[3] = diag[3]; be[4] = pfm(r1, diag[4]); be[5] = diag[6];   // removeAt 5, take 6, scaleErrs r1
    be[5] = fe_sub<1>(be[5], fe_dbl<1>(nbl));
    fe_store(row, fe_neg<1>(be[0]));
    fe_store(row + 8, blT);
    for (int j = 1; j < 6; j++) fe_store(row + (size_t)(1 + j) * 8, be[j]);
  }
  for (uint32_t j = 1 + t; j + 5 < D.llen; j += 64) fe_store(row + (size_t)(1 + 5 + j) * 8, fe_load(bls_lin + (size_t)j * 8));
  for (uint32_t i = t; i < D.nlen; i += 64) fe_store(row + (size_t)(1 + D.llen + i) * 8, fe_load(bls_nrm + (size_t)i * 8));
}

// ------------------------------------------------------------------------------------------------ combination
// wit = pub + blWit + t mWit + t^2 dmWit + t^3 rWit + 2 t^5 nWitSum  (TypedReciprocal.hs:441-443); element j of [sc | lin | nrm]
__global__ void __launch_bounds__(256) k_rpp_combine(PDims D, uint32_t batch, const uint32_t *__restrict__ ch, const uint32_t *__restrict__ rows_dm_m,
                                                     const uint32_t *__restrict__ row_r, const uint32_t *__restrict__ row_bl, const uint32_t *__restrict__ aux,
                                                     const uint32_t *__restrict__ pub_sp, const uint32_t *__restrict__ pub_norm, uint32_t *__restrict__ out_s,
                                                     uint32_t *__restrict__ out_lx, uint32_t *__restrict__ out_nx) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * D.T) return;
  const uint32_t b = (uint32_t)(g / D.T), j = (uint32_t)(g % D.T);
  const fe t = fe_load(ch + (size_t)b * 56 + 48);
  const fe t2 = fe_sqr<1>(t), t3 = fe_mul<1>(t2, t);
  const fe m = fe_load(rows_dm_m + (((size_t)b * 2 + 1) * D.T + j) * 8), dm = fe_load(rows_dm_m + (((size_t)b * 2) * D.T + j) * 8);
  const fe r = fe_load(row_r + ((size_t)b * D.T + j) * 8), bl = fe_load(row_bl + ((size_t)b * D.T + j) * 8);
  fe v = fe_add<1>(fe_add<1>(bl, fe_mul<1>(t, m)), fe_add<1>(fe_mul<1>(t2, dm), fe_mul<1>(t3, r)));
  if (j <= 2) {
    const fe two_t5 = fe_dbl<1>(fe_mul<1>(fe_sqr<1>(t2), t));
    v = fe_add<1>(v, fe_mul<1>(two_t5, fe_load(aux + (size_t)b * 24 + j * 8)));        // j = 0: ns_sc, lin[0]: ns_ty, lin[1]: ns_bl
  }
  if (j == 0) fe_store(out_s + (size_t)b * 8, fe_add<1>(v, fe_load(pub_sp + (size_t)b * 8)));
  else if (j <= D.llen) fe_store(out_lx + ((size_t)b * D.llen + (j - 1)) * 8, v);
  else {
    const uint32_t i = j - 1 - D.llen;
    fe_store(out_nx + ((size_t)b * D.nlen + i) * 8, fe_add<1>(v, fe_load(pub_norm + ((size_t)b * D.nlen + i) * 8)));
  }
}

// (the transcript kernels and the randomness live in csrc/rpp_transcript.hip)


}  // namespace bppp

using namespace bppp;

// ================================================================================================ host orchestration of one batch
namespace bppp {

// proveBPM for B proofs whose start state lies in HBM, as ONE stream of kernels behind the range-proof phases: the norm-linear argument in
// fixed-basis mode (csrc/nlb.hip) or the inner-product argument (csrc/ipb.hip), by the setup's flavour; both commit through the handle's comb
// table, both take their challenges from `tr` (calls first_call ...).  In (device): s [B], q [B] (makeNorm's r for the inner-product flavour),
// nx [B][nlen], lc / lx [B][llen]; d_resp [k][B][2][16] scratch.  Out (host): resp [B][k][16] LAST round first (Bulletproof.hs:359), the final
// witness; `d_extra` (extra_points affine points, e.g. the range-proof commitments) comes back in the same download.  Synchronises the stream.
int rpp_argument_stream(bppp_rp *rp, RppTranscript &tr, size_t first_call, size_t B, const uint32_t *a_s, const uint32_t *a_q, const uint32_t *a_nx, const uint32_t *a_lc,
                        const uint32_t *a_lx, uint32_t *d_resp, uint64_t *resp_out, uint64_t *wn_out, uint64_t *wl_out, const uint32_t *d_extra, size_t extra_points,
                        std::vector<uint64_t> &extra_out) {
  bppp_ctx *ctx = rp->ctx;
  hipStream_t st = ctx->stream;
  const bppp_rps::Setup &S = rp->st;
  const size_t nlen = S.nlen, llen = S.llen, k = S.rounds;
  int rc = BPPP_OK;
  extra_out.resize(extra_points * 8);
  std::vector<uint64_t> hresp(k * B * 16 + 16);
  if (S.flavour) {
    const size_t wb = ipb_work_bytes(B, nlen, llen), wit = (B * (S.fn + S.fl) * 32 + 255) & ~(size_t)255;
    if (wb + wit + 256 > rp->awork_bytes) {
      BPPP_HIP(ctx, hipStreamSynchronize(st));
      if (rp->awork) BPPP_HIP(ctx, hipFree(rp->awork));
      rp->awork = nullptr; rp->awork_bytes = 0;
      BPPP_HIP(ctx, hipMalloc(&rp->awork, wb + wit + 256));
      rp->awork_bytes = wb + wit + 256;
    }
    uint32_t *d_wn = (uint32_t *)((char *)rp->awork + wb), *d_wl = d_wn + B * S.fn * 8, *d_flag = nullptr;
    rc = ipb_prove_stream(ctx, rp->comb, tr, first_call, B, nlen, llen, k, S.fn, S.fl, a_s, a_q, a_nx, a_lc, a_lx, rp->awork, wb, d_resp, d_wn, d_wl, &d_flag);
    if (rc) return rc == BPPP_ERR_ARG && !*bppp_last_error(ctx) ? fail(ctx, rc, "rp_prove_batch: inner-product argument: bad arguments") : rc;
    uint32_t hflag = 0;
    if (extra_points) BPPP_HIP(ctx, hipMemcpyAsync(extra_out.data(), d_extra, extra_points * 64, hipMemcpyDeviceToHost, st));
    if (k) BPPP_HIP(ctx, hipMemcpyAsync(hresp.data(), d_resp, k * B * 128, hipMemcpyDeviceToHost, st));
    if (S.fn) BPPP_HIP(ctx, hipMemcpyAsync(wn_out, d_wn, B * S.fn * 32, hipMemcpyDeviceToHost, st));
    if (S.fl) BPPP_HIP(ctx, hipMemcpyAsync(wl_out, d_wl, B * S.fl * 32, hipMemcpyDeviceToHost, st));
    BPPP_HIP(ctx, hipMemcpyAsync(&hflag, d_flag, 4, hipMemcpyDeviceToHost, st));
    BPPP_HIP(ctx, hipStreamSynchronize(st));
    if (hflag) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: a round challenge is zero");
  } else {
    bppp_nlb *nlb = nullptr;
    rc = nlb_create_impl(ctx, B, (const uint64_t *)a_s, (const uint64_t *)rp->d_g(), (const uint64_t *)a_q, (const uint64_t *)a_nx, (const uint64_t *)rp->d_G(), nlen,
                         (const uint64_t *)a_lc, (const uint64_t *)a_lx, (const uint64_t *)rp->d_H(), llen, &nlb, true, rp->comb);
    if (rc) return rc;
    for (size_t round = 0; round < k && !rc; round++) {
      uint32_t *xr_dev = d_resp + round * B * 32;
      rc = nlb_round_commit_dev(nlb, xr_dev); if (rc) break;
      rc = tr.call(xr_dev, first_call + round); if (rc) break;
      if (hipGetLastError() != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "rp_prove_batch: round kernels"); break; }
      rc = nlb_round_collapse_dev(nlb, tr.es);
    }
    // everything the files need comes back now: the range-proof commitments, the 2k responses, the final witness
    if (!rc && ((extra_points && hipMemcpyAsync(extra_out.data(), d_extra, extra_points * 64, hipMemcpyDeviceToHost, st) != hipSuccess) ||
                (k && hipMemcpyAsync(hresp.data(), d_resp, k * B * 128, hipMemcpyDeviceToHost, st) != hipSuccess)))
      rc = fail(ctx, BPPP_ERR_HIP, "rp_prove_batch: result download");
    if (!rc) rc = bppp_nlb_get_witness(nlb, wn_out, wl_out, nullptr);        // synchronises the stream
    bppp_nlb_destroy(nlb);
    if (rc) return rc;
  }
  for (size_t b = 0; b < B; b++)
    for (size_t round = 0; round < k; round++) memcpy(resp_out + (b * k + (k - 1 - round)) * 16, &hresp[(round * B + b) * 16], 128);   // responses LAST round first (:359)
  return BPPP_OK;
}

int rpp_device_prove(bppp_rp *rp, const RppHostInputs &in, RppOutputs &out) {
  bppp_ctx *ctx = rp->ctx;
  hipStream_t st = ctx->stream;
  const bppp_rps::Setup &S = rp->st;
  const size_t B = in.batch, nr = S.rds.size(), nlen = S.nlen, llen = S.llen, k = S.rounds, T = 1 + llen + nlen;
  PDims D; D.nlen = (uint32_t)nlen; D.llen = (uint32_t)llen; D.nr = (uint32_t)nr; D.T = (uint32_t)T; D.has_types = S.has_types ? 1u : 0u;
  D.nd = (uint32_t)(14 + (llen - 5) + nlen);
  D.maxb = 16;       // the reciprocal table covers the widest digit base only: at base 16 the phase-2 batch inversion is 18 + nr entries instead of 258 + nr
  for (const bppp_rps::RangeData &rd : S.rds) while (D.maxb < rd.base && D.maxb < 2048) D.maxb <<= 1;       // bases above 2048 take the host-algebra route (csrc/rpprove.hip)
  const bppp_trrp *tb = rp->tabs;
  const TrrpDims TD = tb->D;
  // text capacity per proof: every commitment of the final transcript (4 + nr + 2k points), right-aligned, 16 bytes of slack at the end
  const uint32_t stride = rp->D.text_stride;

  // ---- carve the device workspace
  uint32_t *in_sc = nullptr, *in_pt = nullptr, *dig = nullptr, *mul = nullptr, *mss = nullptr, *rnd = nullptr, *rows_dm_m = nullptr, *row_r = nullptr, *row_bl = nullptr,
           *ccbuf = nullptr, *invtab = nullptr, *aux = nullptr, *ch = nullptr, *es = nullptr, *tstart = nullptr, *ptbuf = nullptr, *a_s = nullptr, *a_q = nullptr,
           *a_lx = nullptr, *a_nx = nullptr, *p_sp = nullptr, *p_norm = nullptr, *p_cs = nullptr, *p_init = nullptr;
  uint8_t *text = nullptr, *prefix = nullptr, *hdrs = nullptr; uint32_t *d_resp = nullptr, *d_com = nullptr, *cscratch = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    Carver cv(pass ? rp->pwork : nullptr, rp->pwork_bytes);
    in_sc = cv.take<uint32_t>(B * nr * 24); in_pt = cv.take<uint32_t>(B * nr * 16);
    dig = cv.take<uint32_t>(B * nlen); mul = cv.take<uint32_t>(B * nlen); mss = cv.take<uint32_t>(B * (llen - 6) + 1);
    rnd = cv.take<uint32_t>(B * (size_t)D.nd * 8);
    rows_dm_m = cv.take<uint32_t>(2 * B * T * 8); row_r = cv.take<uint32_t>(B * T * 8); row_bl = cv.take<uint32_t>(B * T * 8);
    ccbuf = cv.take<uint32_t>(B * nlen * 8); invtab = cv.take<uint32_t>(B * (size_t)(INV_DIG + D.maxb) * 8); aux = cv.take<uint32_t>(B * 24);
    ch = cv.take<uint32_t>(B * 56); es = cv.take<uint32_t>(B * 8); tstart = cv.take<uint32_t>(B);
    ptbuf = cv.take<uint32_t>(B * (2 + nr) * 16);
    a_s = cv.take<uint32_t>(B * 8); a_q = cv.take<uint32_t>(B * 8); a_lx = cv.take<uint32_t>(B * llen * 8); a_nx = cv.take<uint32_t>(B * nlen * 8);
    p_sp = cv.take<uint32_t>(B * 8); p_norm = cv.take<uint32_t>(B * nlen * 8); p_cs = cv.take<uint32_t>(B * llen * 8); p_init = cv.take<uint32_t>(B * (4 + nr) * 8);
    text = cv.take<uint8_t>(B * (size_t)stride + 64); prefix = cv.take<uint8_t>(B * in.prefix_len + 16); hdrs = cv.take<uint8_t>(RppTranscript::hdr_bytes(3 + k) + 16);
    d_resp = cv.take<uint32_t>(k * B * 32 + 16); d_com = cv.take<uint32_t>(4 * B * 16 + 16);
    cscratch = cv.take<uint32_t>(std::max(comb_rows_scratch_bytes(B), comb_scratch_bytes(2 * B)) / 4 + 16);
    if (!pass) { int rc = rpp_ensure_pwork(rp, cv.off); if (rc) return rc; }
  }
  // ---- uploads: inputs, digits, multiplicities, prefixes
  BPPP_HIP(ctx, hipMemcpyAsync(in_sc, in.in_sc, B * nr * 96, hipMemcpyHostToDevice, st));
  BPPP_HIP(ctx, hipMemcpyAsync(dig, in.dig, B * nlen * 4, hipMemcpyHostToDevice, st));
  BPPP_HIP(ctx, hipMemcpyAsync(mul, in.mul, B * nlen * 4, hipMemcpyHostToDevice, st));
  if (llen > 6) BPPP_HIP(ctx, hipMemcpyAsync(mss, in.mss, B * (llen - 6) * 4, hipMemcpyHostToDevice, st));
  if (in.prefix_len) BPPP_HIP(ctx, hipMemcpyAsync(prefix, in.prefix, B * in.prefix_len, hipMemcpyHostToDevice, st));
  // fixed-basis mode (comb table in place): the whole proof is ONE stream of kernels — commitments stay on the device until the end,
  // every oracle call reads its points where they lie, and the headers of all 3 + k oracle calls go up here
  const bool stream_mode = rp->comb != nullptr && !rp->opt.fold_points;
  const size_t cscratch_bytes = std::max(comb_rows_scratch_bytes(B), comb_scratch_bytes(2 * B));
  // A handful of proofs: the oracle moves to the host.  One GPU lane walks the ~160 SHA-256 blocks of a 64by64 transcript in ~0.6 ms
  // (11 times per proof); a host core needs ~50 us, which pays for the round trip of the new points and the challenges as long as
  // the batch is small (BPPP_RP_HOST_ORACLE_MAX, default 64 proofs: 1 proof 5.2 ms against 12.0 ms, 32 proofs 11.2 against 12.7, level at 128; the
  // host hashes with the CPU's SHA extensions, csrc/sha256.hip.h, up to 16 threads).
  const bool host_oracle = stream_mode && B <= rp->opt.host_oracle_prove;
  // the oracle calls of proveTRRPM (TypedReciprocal.hs:412, :421, :438): ([dmCom, mCom] ++ nComs) -> e x r0; [rCom] -> q x' r1; [blCom] -> t
  RppTranscript tr;
  { int rc_ = tr.begin(rp, B, {RppCall{(uint32_t)(2 + nr), 3, 0}, RppCall{1, 3, 3}, RppCall{1, 1, 6}}, k, host_oracle, text, tstart, hdrs, ch, es); if (rc_) return rc_; }
  auto oracle_dev = [&](const uint32_t *pts_dev, size_t call) -> int { return tr.call(pts_dev, call); };
  { int rc_ = rpp_draws(ctx, prefix, in.prefix_len, B, D.nd, rnd); if (rc_) return rc_; }
  { const uint64_t n = (uint64_t)B * T;
    k_rpp_rows_dm_m<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(D, tb->pos_kind, tb->pos_range, in_sc, dig, mul, mss, rnd, (uint32_t)B, rows_dm_m); }
  BPPP_HIP(ctx, hipGetLastError());
  int rc = rpp_commit_inputs(rp, in_sc, B * nr, in_pt); if (rc) return rc;
  BPPP_HIP(ctx, hipMemcpyAsync(out.input_coms, in_pt, B * nr * 64, hipMemcpyDeviceToHost, st));
  if (stream_mode) {
    const size_t lds2 = (2 * (INV_DIG + D.maxb + nr) + 128 + TRRP_MAX_SLOTS) * 32, lds3 = ((size_t)6 * 64 + nr + TRRP_MAX_SLOTS + 4) * 32;
    if (lds2 > 160 * 1024) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: too many ranges for the device prover");
    if (lds2 > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_rpp_phase2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    if (lds3 > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_rpp_phase3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    auto comb = [&](const uint32_t *rows, size_t n, uint32_t *dst, int hint) -> int {
      int r_ = comb_msm(rp->comb, rows, n, dst, st, hint, 0, cscratch, cscratch_bytes);
      return r_ ? fail(ctx, r_, bppp_last_error(rp->comb->ctx)) : BPPP_OK;
    };
    uint32_t *c_dmm = d_com, *c_r = d_com + 2 * B * 16, *c_bl = d_com + 3 * B * 16;
    std::vector<uint64_t> hcom_host;
    // (e, x, r0) <- oracle ([dmCom, mCom] ++ nComs)   (TypedReciprocal.hs:412)
    rc = comb(rows_dm_m, 2 * B, c_dmm, COMB_ROWS_ANY); if (rc) return rc;          // digits and multiplicities: small scalars
    BPPP_HIP(ctx, hipMemcpy2DAsync(ptbuf, (2 + nr) * 64, c_dmm, 128, 128, B, hipMemcpyDeviceToDevice, st));
    BPPP_HIP(ctx, hipMemcpy2DAsync(ptbuf + 32, (2 + nr) * 64, in_pt, nr * 64, nr * 64, B, hipMemcpyDeviceToDevice, st));
    rc = oracle_dev(ptbuf, 0); if (rc) return rc;
    k_rpp_phase2<<<dim3((unsigned)B), dim3(64), lds2, st>>>(D, TD, tb->pos_kind, tb->pos_range, tb->pos_slot, tb->pos_sym, tb->syms, in_sc, dig, rnd, ch, row_r, ccbuf, invtab);
    // (q, x', r1) <- oracle [rCom]
    rc = comb(row_r, B, c_r, COMB_ROWS_DENSE); if (rc) return rc;                  // reciprocals: full width
    rc = oracle_dev(c_r, 1); if (rc) return rc;
    k_rpp_phase3<<<dim3((unsigned)B), dim3(64), lds3, st>>>(D, TD, tb->pos_kind, tb->pos_range, tb->pos_slot, tb->pos_coeff, tb->range_assumed, tb->syms, tb->cs_slot, tb->cs_sym,
                                                            in_sc, dig, mul, rnd, ch, rows_dm_m, row_r, ccbuf, invtab, row_bl, aux);
    // t <- oracle [blCom]; public constants and linear weights by the verifier's kernel; the combined witness
    rc = comb(row_bl, B, c_bl, COMB_ROWS_DENSE); if (rc) return rc;
    rc = oracle_dev(c_bl, 2); if (rc) return rc;
    BPPP_HIP(ctx, hipGetLastError());
    rc = bppp_trrp_public_device(rp->tabs, B, ch, a_q, p_sp, p_norm, p_cs, p_init); if (rc) return rc;
    { const uint64_t n = (uint64_t)B * T;
      k_rpp_combine<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(D, (uint32_t)B, ch, rows_dm_m, row_r, row_bl, aux, p_sp, p_norm, a_s, a_lx, a_nx); }
    BPPP_HIP(ctx, hipGetLastError());
    // proveBPM in lockstep (src/Bulletproof.hs:357-359): the start state taken where it lies in HBM, k rounds queued back to back
    rc = rpp_argument_stream(rp, tr, 3, B, a_s, a_q, a_nx, p_cs, a_lx, d_resp, out.resp, out.wit_norm, out.wit_lin, d_com, 4 * B, hcom_host); if (rc) return rc;
    for (size_t b = 0; b < B; b++) {
      memcpy(out.c_dm + 8 * b, &hcom_host[16 * b], 64); memcpy(out.c_m + 8 * b, &hcom_host[16 * b + 8], 64);
      memcpy(out.c_r + 8 * b, &hcom_host[(2 * B + b) * 8], 64); memcpy(out.c_bl + 8 * b, &hcom_host[(3 * B + b) * 8], 64);
    }
    return BPPP_OK;
  }
  std::vector<uint64_t> com(2 * B * 8);
  rc = rpp_commit_rows(rp, rows_dm_m, 2 * B, com.data()); if (rc) return rc;            // synchronises
  for (size_t b = 0; b < B; b++) { memcpy(out.c_dm + 8 * b, &com[16 * b], 64); memcpy(out.c_m + 8 * b, &com[16 * b + 8], 64); }

  // the oracle of this flow: the new commitments of every proof come from the host ([B][m] affine, in the order the reference conses them)
  auto oracle = [&](const uint64_t *pts_host, size_t call) -> int {
    BPPP_HIP(ctx, hipMemcpyAsync(ptbuf, pts_host, B * tr.calls[call].points * 64, hipMemcpyHostToDevice, st));
    int r_ = tr.call(ptbuf, call); if (r_) return r_;
    BPPP_HIP(ctx, hipStreamSynchronize(st));          // the caller's staging is on the stack / reused
    return BPPP_OK;
  };

  // ---- (e, x, r0) <- oracle ([dmCom, mCom] ++ nComs)   (TypedReciprocal.hs:412)
  {
    std::vector<uint64_t> pts(B * (2 + nr) * 8);
    BPPP_HIP(ctx, hipStreamSynchronize(st));          // out.input_coms has landed
    for (size_t b = 0; b < B; b++) {
      memcpy(&pts[(b * (2 + nr)) * 8], out.c_dm + 8 * b, 64); memcpy(&pts[(b * (2 + nr) + 1) * 8], out.c_m + 8 * b, 64);
      memcpy(&pts[(b * (2 + nr) + 2) * 8], out.input_coms + 8 * nr * b, nr * 64);
    }
    rc = oracle(pts.data(), 0); if (rc) return rc;
  }
  const size_t m2 = INV_DIG + D.maxb + nr;
  const size_t lds2 = (2 * m2 + 128 + TRRP_MAX_SLOTS) * 32;
  if (lds2 > 160 * 1024) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: too many ranges for the device prover");
  if (lds2 > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_rpp_phase2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  k_rpp_phase2<<<dim3((unsigned)B), dim3(64), lds2, st>>>(D, TD, tb->pos_kind, tb->pos_range, tb->pos_slot, tb->pos_sym, tb->syms, in_sc, dig, rnd, ch, row_r, ccbuf, invtab);
  BPPP_HIP(ctx, hipGetLastError());
  rc = rpp_commit_rows(rp, row_r, B, out.c_r); if (rc) return rc;
  // ---- (q, x', r1) <- oracle [rCom]
  rc = oracle(out.c_r, 1); if (rc) return rc;
  const size_t lds3 = ((size_t)6 * 64 + nr + TRRP_MAX_SLOTS + 4) * 32;
  if (lds3 > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_rpp_phase3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
  k_rpp_phase3<<<dim3((unsigned)B), dim3(64), lds3, st>>>(D, TD, tb->pos_kind, tb->pos_range, tb->pos_slot, tb->pos_coeff, tb->range_assumed, tb->syms, tb->cs_slot, tb->cs_sym,
                                                          in_sc, dig, mul, rnd, ch, rows_dm_m, row_r, ccbuf, invtab, row_bl, aux);
  BPPP_HIP(ctx, hipGetLastError());
  rc = rpp_commit_rows(rp, row_bl, B, out.c_bl); if (rc) return rc;
  // ---- t <- oracle [blCom]; public constants and linear weights by the verifier's kernel; the combined witness
  rc = oracle(out.c_bl, 2); if (rc) return rc;
  rc = bppp_trrp_public_device(rp->tabs, B, ch, a_q, p_sp, p_norm, p_cs, p_init); if (rc) return rc;
  { const uint64_t n = (uint64_t)B * T;
    k_rpp_combine<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(D, (uint32_t)B, ch, rows_dm_m, row_r, row_bl, aux, p_sp, p_norm, a_s, a_lx, a_nx); }
  BPPP_HIP(ctx, hipGetLastError());

  // ---- proveBPM in lockstep (src/Bulletproof.hs:357-359), the start state taken where it lies in HBM
  bppp_nlb *nlb = nullptr;
  rc = nlb_create_impl(ctx, B, (const uint64_t *)a_s, (const uint64_t *)rp->d_g(), (const uint64_t *)a_q, (const uint64_t *)a_nx, (const uint64_t *)rp->d_G(), nlen,
                       (const uint64_t *)p_cs, (const uint64_t *)a_lx, (const uint64_t *)rp->d_H(), llen, &nlb, true, nullptr);
  if (rc) return rc;
  std::vector<uint64_t> sX(B * 4), sR(B * 4), X(B * 8), R(B * 8), eh(B * 4), xr(B * 16);
  for (size_t round = 0; round < k && !rc; round++) {
    rc = bppp_nlb_round_commit(nlb, sX.data(), X.data(), sR.data(), R.data());
    if (rc) break;
    const size_t slot = k - 1 - round;                   // responses LAST round first (:359)
    for (size_t b = 0; b < B; b++) {
      memcpy(&xr[16 * b], &X[8 * b], 64); memcpy(&xr[16 * b + 8], &R[8 * b], 64);
      memcpy(out.resp + (b * k + slot) * 16, &xr[16 * b], 128);
    }
    rc = oracle(xr.data(), 3 + round); if (rc) break;
    if (hipMemcpyAsync(eh.data(), es, B * 32, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "rp_prove_batch: challenge download"); break; }
    rc = bppp_nlb_round_collapse(nlb, eh.data());
  }
  if (!rc) rc = bppp_nlb_get_witness(nlb, out.wit_norm, out.wit_lin, nullptr);
  bppp_nlb_destroy(nlb);
  return rc;
}

}  // namespace bppp
