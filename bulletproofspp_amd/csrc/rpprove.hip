// rpprove.hip — batch prover of typed reciprocal range proofs: B proofs of one setup advance in lockstep.
//
// Replaces, with a leading batch dimension, proveM of RangeProof (src/RangeProof.hs:93-97):
//   proveTRRPM                      src/RangeProof/TypedReciprocal.hs:399-446  (phases 1-3: digits and multiplicities, reciprocals,
//                                   blinding — makePhase1s :133-161, makePhase2s :185-205, makeSharedCoeffs :213-216,
//                                   makeErrorTerms :226-243, makePublicConsts :246-274, makeBpCoeffs :391-396)
//   the blinding algebra            src/RangeProof/Internal.hs:118-196 (blindWitness, blindErrWitness, blindBlindingTerm)
//   commitRPW                       src/RangeProof/Internal.hs:45-50
//   proveBPM                        src/Bulletproof.hs:357-359 (the lockstep argument of csrc/nlb.hip)
//   encodeProof'                    src/RangeProof.hs:60-66, src/Encoding.hs:130-134
// Work split: every group operation is on the device — the input commitments through a fixed-base window table of (g, H0, H1)
// (k_rp_commit_inputs: B x #ranges three-term commitments in one launch), the four range-proof commitments of all proofs as
// batched MSMs over the registered basis (2B, B, B instances of 1 + linLen + nrmLen terms; comb MSMs once the handle has its comb
// table, csrc/comb.hip), the argument through csrc/nlb.hip.  The per-proof field algebra (O(nrmLen) multiplications per phase) and
// the transcript hashing (the CLI's shaOracle and hashToScalar, app/Main.hs:64-87) run on the device (csrc/rpprove_dev.hip;
// prove_batch_host below keeps the first version of this file, with both on the host cores, for comparison); the host extracts
// the digits of the plain amounts, stages inputs in pinned memory and writes the files.  Large batches: two half-batches in flight.
#include <string.h>
#include <array>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "ec.hip.h"
#include "comb.hpp"
#include "rp_internal.hpp"
#include "rpprove_dev.hpp"
#include "rpprove_host.hpp"
#include "sha256.hip.h"

namespace bppp {

// ---- input commitments  v g + ty H0 + bl H1  (scalarPairRPW', src/RangeProof/Internal.hs:59-60) by fixed-base windows:
// table[base][w][d - 1] = d 16^w P_base (affine), 3 x 64 x 15 points; a commitment is at most 192 mixed additions, no doubling.
static constexpr int FB_BASES = 3, FB_WIN = 64, FB_DIG = 15;
__global__ void __launch_bounds__(64) k_rp_commit_inputs(const uint32_t *__restrict__ table, const uint32_t *__restrict__ sc, uint64_t n,
                                                         uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzz acc = xyzz_inf();
  for (int base = 0; base < FB_BASES; base++) {
    const fe s = fe_load(sc + (i * FB_BASES + base) * 8);
    if (fe_is_zero(s)) continue;
#pragma unroll 1
    for (int w = 0; w < FB_WIN; w++) {
      uint32_t limb = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) if ((w >> 3) == k) limb = s.v[k];
      const uint32_t d = (limb >> (4 * (w & 7))) & 15u;
      if (d) xyzz_madd(acc, aff_load(table + ((size_t)(base * FB_WIN + w) * FB_DIG + (d - 1)) * 16));
    }
  }
  aff_store(out + i * 16, xyzz_to_aff(acc));
}

}  // namespace bppp

using namespace bppp;
using namespace bppp_host;
using namespace bppp_rpp;


namespace bppp {
// the same oracle for the device prover's small batches (csrc/rpprove_dev.hip): `groups` / `npoints` are one proof's transcript so far,
// pts [m][8] its new commitments; out[count][4].  A host core hashes a 64by64 transcript in ~50 us; one GPU lane needs ~600 us.
void rpp_host_oracle(const std::string &tag, std::vector<std::string> &groups, size_t &npoints, const uint64_t *pts, size_t m, int count, uint64_t *out) {
  std::string g;
  g.reserve(m * 160);
  for (size_t i = 0; i < m; i++) point_text(g, pts + 8 * i);
  groups.push_back(std::move(g));
  npoints += m;
  for (int n = 1; n <= count; n++) {
    Sha256 h;
    const std::string hdr = tag + std::to_string(n) + std::to_string(npoints);
    h.update(hdr.data(), hdr.size());
    for (size_t k = groups.size(); k-- > 0;) h.update(groups[k].data(), groups[k].size());
    uint32_t d[8];
    h.finish(d);
    digest_to_fr(d).store(out + 4 * (n - 1));
  }
}
}  // namespace bppp
namespace {
// witnessTRRP (TypedReciprocal.hs:372-389) + makePhase1s (:133-161) + getDsMs (:74-80): fills d, mi, pv, ms_shared
bool make_witness(const Setup &st, PState &ps, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds) {
  const size_t nr = st.rds.size();
  ps.v.resize(nr); ps.ty.resize(nr); ps.bl.resize(nr);
  std::vector<U256> amt(nr);
  for (size_t i = 0; i < nr; i++) {
    amt[i] = U256::load(amounts + 4 * i);
    if (!scalars_canonical(types + 4 * i, 1) || !scalars_canonical(blinds + 4 * i, 1)) { ps.err = "type / blinding not canonical"; return false; }
    ps.v[i] = bppp_rps::s_mod_n(amt[i]); ps.ty[i] = U256::load(types + 4 * i); ps.bl[i] = U256::load(blinds + 4 * i);
  }
  if (st.has_types) {                                     // amounts of every type must balance (:376-381)
    std::vector<std::pair<U256, U256>> sums;
    auto add = [&](const U256 &ty, const U256 &val, bool neg) {
      for (auto &kv : sums) if (kv.first == ty) { kv.second = neg ? fs(kv.second, val) : fa(kv.second, val); return; }
      sums.emplace_back(ty, neg ? fneg(val) : val);
    };
    for (const auto &pv : st.pubs) add(pv.type, pv.amount, pv.is_output);
    for (size_t i = 0; i < nr; i++) add(ps.ty[i], ps.v[i], st.rds[i].output);
    for (auto &kv : sums) if (!kv.second.is_zero()) { ps.err = "amounts of some type do not balance"; return false; }
  }
  ps.d.assign(st.nlen, U256::zero()); ps.mi.assign(st.nlen, U256::zero()); ps.pv.assign(st.nlen, U256::one());
  // the shared multiplicities are counts of digits: summed as integers (64 ranges x 255 digit values of modular additions per proof were most of
  // this function's time), converted once at the end
  std::vector<uint32_t> &msc = ps.tmp_msc;
  msc.assign(st.llen - 6, 0u);
  std::vector<size_t> base_off(st.m_bases.size(), 0);
  for (size_t k = 1; k < st.m_bases.size(); k++) base_off[k] = base_off[k - 1] + st.m_bases[k - 1] - 1;
  auto moff = [&](uint32_t base) { size_t k = std::lower_bound(st.m_bases.begin(), st.m_bases.end(), base) - st.m_bases.begin(); return base_off[k]; };
  size_t p = 0;
  if (st.has_types)
    for (size_t i = 0; i < nr; i++, p++) { ps.d[p] = ps.ty[i]; ps.pv[p] = ps.v[i]; }
  for (size_t i = 0; i < nr; i++) {
    const RangeData &rd = st.rds[i];
    if (rd.assumed) continue;
    if (bppp_rps::s_lt(amt[i], rd.lo) || !bppp_rps::s_lt(amt[i], rd.hi)) { ps.err = "value outside its range"; return false; }
    std::vector<uint32_t> &ds = ps.tmp_ds, &cnt = ps.tmp_cnt, &ms = ps.tmp_ms;       // reused across ranges and proofs
    bppp_rps::digits_into(rd, bppp_rps::u_sub(amt[i], rd.lo), ds);
    const uint32_t b = rd.base;
    const size_t p0 = st.first_pos[i];
    if (rd.shared) {
      // the digits go to their positions; their multiplicities to the shared table of this base (baseMss, :363-370; the bit's to base 2), one
      // count per non-zero digit (ms aligned with ns = [1 | hasBit] ++ [1 .. b-1], :141-145)
      for (size_t j = 0; j < ds.size(); j++) ps.d[p0 + j] = small(ds[j]);
      if (rd.has_bit) msc[moff(2)] += ds[0];
      const size_t o = moff(b);
      for (size_t j = rd.has_bit ? 1 : 0; j < ds.size(); j++) if (ds[j]) msc[o + ds[j] - 1]++;
    } else {
      // ms aligned with ns = [1 | hasBit] ++ [1 .. b-1]: the bit itself, then how often each non-zero digit value occurs (:141-145)
      cnt.assign(b, 0);
      for (size_t j = rd.has_bit ? 1 : 0; j < ds.size(); j++) cnt[ds[j]]++;
      ms.clear();
      if (rd.has_bit) ms.push_back(ds[0]);
      for (uint32_t s = 1; s < b; s++) ms.push_back(cnt[s]);
      const size_t ln = std::max(ds.size(), ms.size());
      for (size_t j = 0; j < ln; j++) {
        if (j < ds.size()) ps.d[p0 + j] = small(ds[j]);
        if (j < ms.size()) ps.mi[p0 + j] = small(ms[j]);
      }
    }
  }
  ps.ms_shared.resize(st.llen - 6);
  for (size_t j = 0; j < msc.size(); j++) ps.ms_shared[j] = small(msc[j]);
  return true;
}

// blindWitness n = 3 (src/RangeProof/Internal.hs:130-139): five fresh scalars with one zero slot, 7 blinding entries in all
RPW blind_witness(int k, const std::vector<U256> &ls, const std::vector<U256> &ns, Rnd &rnd, size_t llen) {
  U256 b[7];
  for (int i = 0; i < 7; i++) b[i] = U256::zero();
  const int zero_at = 2 * 3 - k;                           // insertAt (2n - k) 0
  int j = 0;
  for (int i = 0; i < 5; i++, j++) { if (j == zero_at) j++; b[j] = rnd.next(); }
  RPW w;
  w.sc = b[0];
  w.lin.assign(llen, U256::zero());
  for (int i = 0; i < 6; i++) w.lin[i] = b[1 + i];
  for (size_t i = 0; i < ls.size(); i++) w.lin[6 + i] = ls[i];
  w.nrm = ns;
  return w;
}
// blindErrWitness n = 3 (:142-149): [a, b, c, 0, d] ++ es, padded to 7
RPW blind_err_witness(const U256 &err7, const std::vector<U256> &ns, Rnd &rnd, size_t llen) {
  U256 b[7];
  for (int i = 0; i < 7; i++) b[i] = U256::zero();
  b[0] = rnd.next(); b[1] = rnd.next(); b[2] = rnd.next(); b[4] = rnd.next(); b[5] = err7;
  RPW w;
  w.sc = b[0];
  w.lin.assign(llen, U256::zero());
  for (int i = 0; i < 6; i++) w.lin[i] = b[1 + i];
  w.nrm = ns;
  return w;
}

// makeBaseMap: sortedBases zipped with x^3, x^5, ... (TypedReciprocal.hs:349)
std::vector<U256> base_map(const Setup &st, const U256 &x) {
  std::vector<U256> out(st.sorted_bases.size());
  U256 c = fm(fm(x, x), x), xx = fm(x, x);
  for (size_t k = 0; k < out.size(); k++) { out[k] = c; c = fm(c, xx); }
  return out;
}

// makePhase2s with the private fields (TypedReciprocal.hs:185-205)
void make_phase2(const Setup &st, PState &ps) {
  const size_t n = st.nlen;
  const std::vector<U256> bm = base_map(st, ps.x);
  const U256 xx = fm(ps.x, ps.x);
  std::vector<U256> xpow(st.rds.size());
  { U256 c = xx; for (size_t j = 0; j < xpow.size(); j++) { xpow[j] = c; c = fm(c, xx); } }     // x^(2 (j + 1))
  ps.u.resize(n); ps.vv.resize(n); ps.rr.resize(n); ps.cc.resize(n);
  std::vector<U256> dens(n), ss(n);
  for (size_t i = 0; i < n; i++) {
    const Pos &p = st.pos[i];
    const uint32_t kind = p.kind & 0xFFu;
    const U256 &xi = xpow[p.range];
    dens[i] = fa(ps.e, ps.d[i]);
    ss[i] = U256::zero();
    if (kind == bppp_rps::POS_TYPING) {
      ps.vv[i] = (p.kind & bppp_rps::POS_F_IO) ? fneg(ps.x) : ps.x;
      ps.u[i] = (p.kind & bppp_rps::POS_F_IA) ? U256::zero() : xi;
    } else {
      ps.vv[i] = bm[st.slot_of(p.radix)];
      ps.u[i] = fm(xi, p.coeff);
      if (kind == bppp_rps::POS_INLINE && p.sym_small) ss[i] = fa(ps.e, small(p.sym_small));
    }
  }
  batch_inv(dens); batch_inv(ss);
  for (size_t i = 0; i < n; i++) {
    ps.rr[i] = fm(ps.pv[i], dens[i]);
    ps.cc[i] = ss[i].is_zero() ? U256::zero() : fm(ps.vv[i], fs(ps.e_inv, ss[i]));
  }
}

// makeSharedCoeffs (:213-216)
std::vector<U256> make_shared_coeffs(const Setup &st, const PState &ps) {
  const std::vector<U256> bm = base_map(st, ps.x);
  std::vector<U256> xs, ss;
  for (uint32_t b : st.m_bases)
    for (uint32_t s = 1; s < b; s++) { xs.push_back(bm[st.slot_of(b)]); ss.push_back(fa(ps.e, small(s))); }
  batch_inv(ss);
  for (size_t i = 0; i < xs.size(); i++) xs[i] = fm(xs[i], fs(ps.e_inv, ss[i]));
  return xs;
}

// inputCoeffs (:325-328)
std::vector<U256> input_coeffs(const Setup &st, const U256 &x, const U256 &q0) {
  const size_t nr = st.rds.size();
  std::vector<U256> out(nr);
  const U256 xx = fm(x, x);
  U256 xp = xx, qp = q0;
  for (size_t i = 0; i < nr; i++) {
    U256 c = st.rds[i].assumed ? U256::zero() : xp;
    if (st.has_types) c = fa(c, qp);
    out[i] = c;
    xp = fm(xp, xx); qp = fm(qp, q0);
  }
  return out;
}

// makeErrorTerms (:226-243) with q2 = q0^(i+1), bl = the norm blinding of position i
void make_error_terms(const Setup &st, const PState &ps, const std::vector<U256> &bls_ms, const std::vector<U256> &bls_nrm, U256 tot[6]) {
  for (int k = 0; k < 6; k++) tot[k] = U256::zero();
  U256 s3 = U256::zero();
  for (size_t i = 0; i < ps.shared_cs.size(); i++) s3 = fa(s3, fm(ps.shared_cs[i], bls_ms[i]));
  tot[3] = fdbl(s3);
  U256 q2 = ps.q0;
  for (size_t i = 0; i < st.nlen; i++) {
    const bool is_t = (st.pos[i].kind & 0xFFu) == bppp_rps::POS_TYPING;
    const U256 &d = ps.d[i], &m = ps.mi[i], &u = ps.u[i], &v = ps.vv[i], &r = ps.rr[i], &c = ps.cc[i], &bl = bls_nrm[i];
    const U256 rC = is_t ? fm(ps.xp, fa(u, q2)) : u;
    const U256 dC = fa(v, fm(q2, ps.e));
    const U256 q2d_dC = fa(fm(q2, d), dC), q2r_rC = fa(fm(q2, r), rC), q2bl = fm(q2, bl), q2m = fm(q2, m);
    tot[0] = fa(tot[0], fm(q2bl, bl));
    tot[1] = fa(tot[1], fdbl(fm(q2m, bl)));
    tot[2] = fa(tot[2], fa(fm(q2m, m), fdbl(fm(bl, q2d_dC))));
    tot[3] = fa(tot[3], fdbl(fa(fm(bl, q2r_rC), fm(m, q2d_dC))));
    tot[4] = fa(tot[4], fa(fa(fm(fm(q2, d), d), fdbl(fm(d, dC))), fdbl(fa(fm(bl, c), fm(m, q2r_rC)))));
    tot[5] = fa(tot[5], fa(fa(fm(fm(q2, r), r), fdbl(fm(r, rC))), fdbl(fm(c, d))));
    q2 = fm(q2, ps.q0);
  }
}

// blindBlindingTerm for three earlier witnesses [mWit, dmWit, rWit] (src/RangeProof/Internal.hs:154-196)
RPW blind_blinding_term(const std::vector<U256> &bls_lin, const std::vector<U256> &bls_nrm, const U256 &tC, const PState &ps, const U256 errs[6],
                        const U256 &input_bl, size_t llen) {
  const int n = 3;
  const U256 blT = bls_lin[0];
  const U256 rs_inv = fm(ps.r0_inv, ps.r1_inv);
  // rows: [sc] ++ the first 2n linear entries; the error witness keeps only n + 1 of them (the rest zero)
  U256 rows[3][7];
  const RPW *w1[2] = {&ps.m, &ps.dm};
  for (int a = 0; a < 2; a++) { rows[a][0] = w1[a]->sc; for (int j = 0; j < 6; j++) rows[a][1 + j] = w1[a]->lin[j]; }
  rows[2][0] = ps.r.sc;
  for (int j = 0; j < 6; j++) rows[2][1 + j] = j < n + 1 ? ps.r.lin[j] : U256::zero();
  for (int a = 0; a < 3; a++) for (int j = 2; j < 7; j++) rows[a][j] = fneg(rows[a][j]);
  // errs1 = negate ([errs0 - tC blT] ++ rs_inv * errs[1..])
  U256 table[4][7];
  U256 e1[6];
  e1[0] = fneg(fs(errs[0], fm(tC, blT)));
  for (int j = 1; j < 6; j++) e1[j] = fneg(fm(rs_inv, errs[j]));
  auto scale_errs = [&](U256 *xs6, const U256 &s) { xs6[n + 1] = fm(s, xs6[n + 1]); };     // scaleErrs (:118-121): only entry n + 1 of six
  auto ins = [&](U256 *dst7, const U256 *src6) { for (int j = 0; j < 5; j++) dst7[j] = src6[j]; dst7[5] = U256::zero(); dst7[6] = src6[5]; };   // insertAt (2n - 1) 0
  ins(table[0], e1);
  for (int a = 0; a < 3; a++) {
    U256 r6[6];
    r6[0] = fa(fm(rs_inv, rows[a][0]), fm(fm(rs_inv, tC), rows[a][1]));                      // addConsts
    for (int j = 1; j < 6; j++) r6[j] = rows[a][j + 1];
    scale_errs(r6, ps.r1_inv);
    ins(table[1 + a], r6);
  }
  U256 diag[10];
  for (int k = 0; k < 10; k++) diag[k] = U256::zero();
  for (int a = 0; a < 4; a++) for (int b = 0; b < 7; b++) diag[a + b] = fa(diag[a + b], table[a][b]);     // sumDiagonals (:104-111)
  U256 be[6];
  for (int k = 0, j = 0; k < 7 && j < 6; k++) { if (k == 5) continue; be[j++] = diag[k]; }              // removeAt (2n - 1), take 2n
  scale_errs(be, ps.r1);
  be[5] = fs(be[5], fdbl(input_bl));
  RPW w;
  w.sc = fneg(be[0]);
  w.lin.assign(llen, U256::zero());
  w.lin[0] = blT;
  for (int j = 1; j < 6; j++) w.lin[j] = be[j];
  for (size_t j = 1; j < bls_lin.size(); j++) w.lin[5 + j] = bls_lin[j];
  w.nrm = bls_nrm;
  return w;
}

// makePublicConsts (TypedReciprocal.hs:246-274): returns sc and the norm vector
void make_public_consts(const Setup &st, const PState &ps, U256 &sc, std::vector<U256> &nrm) {
  const U256 t2 = fm(ps.t, ps.t), t3 = fm(t2, ps.t), t4 = fm(t2, t2), t5 = fm(t4, ps.t), two_t5 = fdbl(t5);
  const U256 xx = fm(ps.x, ps.x);
  U256 z = U256::zero();
  { U256 xp = xx;
    for (size_t j = 0; j < st.rds.size(); j++) { if (!st.rds[j].assumed) z = fa(z, fm(bppp_rps::s_mod_n(st.rds[j].lo), xp)); xp = fm(xp, xx); } }
  z = fneg(fm(two_t5, z));
  if (st.has_types) {
    std::vector<U256> pr(st.pubs.size());
    for (size_t j = 0; j < pr.size(); j++) pr[j] = fa(ps.e, st.pubs[j].type);
    batch_inv(pr);
    U256 sum = U256::zero();
    for (size_t j = 0; j < pr.size(); j++) { const U256 rv = fm(pr[j], st.pubs[j].amount); sum = st.pubs[j].is_output ? fs(sum, rv) : fa(sum, rv); }
    z = fs(z, fm(fm(two_t5, ps.x), sum));
  }
  nrm.resize(st.nlen);
  U256 q2 = ps.q0, qi2 = ps.q0_inv, acc = U256::zero();
  for (size_t i = 0; i < st.nlen; i++) {
    const bool is_t = (st.pos[i].kind & 0xFFu) == bppp_rps::POS_TYPING;
    U256 rC, p2C;
    if (is_t) { rC = fm(ps.xp, fa(fm(qi2, ps.u[i]), U256::one())); p2C = U256::zero(); }
    else { rC = fm(qi2, ps.u[i]); p2C = fa(fdbl(q2), fdbl(fm(ps.e_inv, ps.vv[i]))); }
    const U256 pv = fa(fa(fm(t2, fa(ps.e, fm(qi2, ps.vv[i]))), fm(t3, rC)), fm(t4, fm(qi2, ps.cc[i])));
    acc = fa(acc, fa(fm(q2, fm(pv, pv)), fm(t5, p2C)));
    nrm[i] = pv;
    q2 = fm(q2, ps.q0); qi2 = fm(qi2, ps.q0_inv);
  }
  sc = fa(z, acc);
}

// the fixed-base table of (g, H0, H1): [3][64][15] affine points, built once per setup on the host (2880 additions, one batch inversion)
int build_fixed_table(bppp_rp *rp) {
  if (rp->d_fixed) return BPPP_OK;
  bppp_ctx *ctx = rp->ctx;
  const Mod &Q = FQ();
  std::vector<HJac> jac;
  jac.reserve(FB_BASES * FB_WIN * FB_DIG);
  const uint64_t *bases[3] = {rp->h_g.data(), rp->h_H.data(), rp->h_H.data() + 8};
  for (int b = 0; b < FB_BASES; b++) {
    HJac cur = hj_from_aff(HAff{U256::load(bases[b]), U256::load(bases[b] + 4)});
    for (int w = 0; w < FB_WIN; w++) {
      HJac acc = cur;
      for (int d = 1; d <= FB_DIG; d++) { jac.push_back(acc); acc = hj_add(acc, cur); }
      cur = acc;                                   // 16 * cur
    }
  }
  std::vector<U256> zs(jac.size());
  for (size_t i = 0; i < jac.size(); i++) zs[i] = jac[i].Z;
  batch_minv(zs.data(), zs.size(), Q);
  std::vector<uint64_t> host(jac.size() * 8, 0);
  for (size_t i = 0; i < jac.size(); i++) {
    if (jac[i].inf()) continue;                    // cannot happen for points of prime order; kept as the infinity encoding
    const U256 zi2 = fqmul(zs[i], zs[i]);
    fqmul(jac[i].X, zi2).store(&host[8 * i]);
    fqmul(jac[i].Y, fqmul(zi2, zs[i])).store(&host[8 * i + 4]);
  }
  BPPP_HIP(ctx, hipMalloc(&rp->d_fixed, host.size() * 8));
  BPPP_HIP(ctx, hipMemcpy(rp->d_fixed, host.data(), host.size() * 8, hipMemcpyHostToDevice));
  return BPPP_OK;
}

int ensure_pwork(bppp_rp *rp, size_t bytes) {
  if (bytes <= rp->pwork_bytes) return BPPP_OK;
  bppp_ctx *ctx = rp->ctx;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (rp->pwork) BPPP_HIP(ctx, hipFree(rp->pwork));
  rp->pwork = nullptr; rp->pwork_bytes = 0;
  BPPP_HIP(ctx, hipMalloc(&rp->pwork, bytes + bytes / 8));
  rp->pwork_bytes = bytes + bytes / 8;
  return BPPP_OK;
}

}  // namespace

namespace bppp {
int rpp_ensure_pwork(bppp_rp *rp, size_t bytes) { return ensure_pwork(rp, bytes); }
int rpp_build_fixed_table(bppp_rp *rp) { return build_fixed_table(rp); }
int rpp_commit_inputs(bppp_rp *rp, const uint32_t *d_in_sc, size_t n, uint32_t *d_out) {
  bppp_ctx *ctx = rp->ctx;
  // g, H0, H1 are the first three points of the registered basis: with its comb table a commitment is <= 3 x 17 additions, not 3 x 64
  if (rp->comb) { int rc = comb_lanes(rp->comb, d_in_sc, FB_BASES, n, d_out, ctx->stream); return rc ? fail(ctx, rc, bppp_last_error(rp->comb->ctx)) : BPPP_OK; }
  k_rp_commit_inputs<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>(rp->d_fixed, d_in_sc, (uint64_t)n, d_out);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}
int rpp_commit_rows(bppp_rp *rp, const uint32_t *d_rows, size_t nrows, uint64_t *host_out) {
  if (!rp->comb) return bppp_msm_basis(rp->commit_basis, d_rows, 1 + rp->st.llen + rp->st.nlen, nrows, host_out);
  bppp_ctx *ctx = rp->ctx;
  if (nrows > rp->comb_out_rows) {
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rp->d_comb_out) BPPP_HIP(ctx, hipFree(rp->d_comb_out));
    rp->d_comb_out = nullptr; rp->comb_out_rows = 0;
    BPPP_HIP(ctx, hipMalloc(&rp->d_comb_out, nrows * 64));
    rp->comb_out_rows = nrows;
  }
  int rc = comb_msm(rp->comb, d_rows, nrows, rp->d_comb_out, ctx->stream);      // one wavefront per instance (the fold route's small batches)
  if (rc) return fail(ctx, rc, bppp_last_error(rp->comb->ctx));
  BPPP_HIP(ctx, hipMemcpyAsync(host_out, rp->d_comb_out, nrows * 64, hipMemcpyDeviceToHost, ctx->stream));
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
}  // namespace bppp

// the comb table of the setup's basis, built at the first large batch: the widest window (<= 18 bits) whose table fits the budget —
// 32 GB by default, BPPP_RP_COMB_GB to change it: 27.6 GB at 774 points (c = 16, built in ~0.3 s), 4.1 GB at c = 13 costs ~15 % more
// additions; BPPP_RP_COMB_BITS forces a width, BPPP_RP_NO_COMB keeps the bucket route
int rp_ensure_comb(bppp_rp *rp) {
  if (rp->comb || rp->opt.no_comb || rp->comb_failed) return BPPP_OK;
  // the table is a persistent allocation of tens of GB: never more than the handle's budget, and never more than half of what is
  // free on the device right now (other handles, other processes on the same GPU); a forced width (comb_bits) skips the budget
  size_t budget = rp->opt.comb_budget, free_b = 0, total_b = 0;
  hipSetDevice(rp->ctx->device);
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 2);
  else (void)hipGetLastError();
  int rc = bppp::comb_create(rp->ctx, rp->d_basis, 1 + rp->st.llen + rp->st.nlen, rp->opt.comb_bits, budget, &rp->comb);
  if (rc) {                                  // no room for the table (or no window fits the budget): the bucket route and the point-folding
    rp->comb = nullptr; rp->comb_failed = true;   // argument serve the batch; the reason stays in bppp_last_error, the attempt is not repeated
    (void)hipGetLastError();                 // a failed hipMalloc leaves its error in the runtime's last-error slot: later launch checks must not see it
    return BPPP_OK;
  }
  rp->comb_owned = true;
  return BPPP_OK;
}

static int prove_batch_host(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                            size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files, size_t index_base);

// encodeProof' (src/RangeProof.hs:60-66): commitments file = the input commitments; proof file = final witness scalars (norm, linear),
// then blCom, rCom, dmCom, mCom and the responses
static void encode_batch(const bppp_rp *rp, size_t B, const RppOutputs &o, uint8_t *coms_files, uint8_t *proof_files) {
  const Setup &st = rp->st;
  const RpDims &D = rp->D;
  const size_t nr = st.rds.size(), k = st.rounds;
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<const uint64_t *> pts;
    for (size_t b = lo; b < hi; b++) {
      pts.assign(nr, nullptr);
      for (size_t i = 0; i < nr; i++) pts[i] = o.input_coms + (b * nr + i) * 8;
      encode_points(coms_files + b * D.coms_bytes, pts.data(), nr);
      uint8_t *pf = proof_files + b * D.proof_bytes;
      for (size_t i = 0; i < st.fn; i++) put_field(pf + 32 * i, U256::load(o.wit_norm + (b * st.fn + i) * 4));
      for (size_t i = 0; i < st.fl; i++) put_field(pf + 32 * (st.fn + i), U256::load(o.wit_lin + (b * st.fl + i) * 4));
      pts.assign(4 + 2 * k, nullptr);
      pts[0] = o.c_bl + 8 * b; pts[1] = o.c_r + 8 * b; pts[2] = o.c_dm + 8 * b; pts[3] = o.c_m + 8 * b;
      for (size_t j = 0; j < 2 * k; j++) pts[4 + j] = o.resp + (b * k) * 16 + 8 * j;
      encode_points(pf + 32 * (st.fn + st.fl), pts.data(), 4 + 2 * k);
    }
  });
}



static int prove_batch_one(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                           size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files, size_t index_base);

extern "C" int bppp_rp_prove_batch(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                                   size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files) {
  if (!rp) return BPPP_ERR_ARG;
  bppp_ctx *ctx = rp->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!batch) return BPPP_OK;
  if (!amounts || (!types && rp->st.kind == 0) || !blinds || (prefix_len && !rand_prefix) || !coms_files || !proof_files || batch >= (1u << 20) || prefix_len > 4096)
    return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: bad arguments");
  const size_t comb_min = rp->opt.comb_min;     // default 1024: the table costs ~0.3 s and tens of GB once: worth it for a handle that proves large batches
  // ... or one that has proved that many proofs in smaller batches: with the table in place every batch size is faster (one 64by64 proof:
  // 12 ms against 22 ms; 256: 22 against 48)
  if (!rp->is_twin && !rp->opt.host_algebra) {
    rp->proved_total += batch;
    if (batch >= comb_min || rp->proved_total >= comb_min) { int rc = rp_ensure_comb(rp); if (rc) return rc; }
  }
  // RangeProof.Binary: with the comb table in place the whole proof is a stream of kernels (csrc/brpprove_dev.hip); before that (small
  // batches) and under BPPP_RP_HOST_ALGEBRA the field algebra and the hashing run on the host cores (prove_batch_binary)
  if (rp->st.kind == 1) {
    if (rp->comb && !rp->opt.host_algebra && !rp->opt.fold_points) {
      // two half-batches in flight (as below for the typed-reciprocal proofs): the transcript hashing, the phase and the round kernels of one half —
      // ~14 ms per 1024 proofs of mostly one-lane-per-proof chains — run under the comb additions of the other
      if (batch < rp->opt.split_min_binary || batch < 2 || rp->is_twin || rp->opt.no_split)
        return prove_batch_binary_dev(rp, batch, amounts, blinds, rand_prefix, prefix_len, coms_files, proof_files);
      { int rc = rp_ensure_twin(rp); if (rc) return rc; }
      if (!rp->twin->comb) rp->twin->comb = rp->comb;              // not owned by the twin
      rp->twin->opt = rp->opt;
      const size_t nrb = rp->st.rds.size(), B0 = (batch + 1) / 2, B1 = batch - B0;
      int rc1 = BPPP_OK;
      std::thread second([&] {
        rc1 = prove_batch_binary_dev(rp->twin, B1, amounts + 4 * nrb * B0, blinds + 4 * nrb * B0, rand_prefix ? rand_prefix + prefix_len * B0 : nullptr, prefix_len,
                                     coms_files + (size_t)rp->D.coms_bytes * B0, proof_files + (size_t)rp->D.proof_bytes * B0, B0);
      });
      const int rc0 = prove_batch_binary_dev(rp, B0, amounts, blinds, rand_prefix, prefix_len, coms_files, proof_files);
      second.join();
      if (rc0) return rc0;
      if (rc1) return fail(ctx, rc1, bppp_last_error(rp->twin_ctx));
      return BPPP_OK;
    }
    return prove_batch_binary(rp, batch, amounts, blinds, rand_prefix, prefix_len, coms_files, proof_files);
  }
  // inner-product flavour without a table: the range-proof phases with their field algebra on the host cores, then the lockstep argument of
  // ip_argument_lockstep (no basis change, no point fold: every commitment an MSM over the registered original basis)
  if (rp->st.flavour != 0 && (!rp->comb || rp->opt.fold_points))
    return prove_batch_host(rp, batch, amounts, types, blinds, rand_prefix, prefix_len, coms_files, proof_files, 0);
  // A large batch runs as TWO half-batches in flight, the second on a twin handle with its own context (stream, workspaces, host
  // thread): the proofs are independent, and the host shares of a half (digits, the argument's half-GCDs and round bookkeeping,
  // the challenge round trips) fall under the kernels of the other.  Same bytes out as one batch (tests).
  const size_t split_min = rp->opt.split_min;   // default 4096; measured: 4096 proofs 91-93 ms split against 95-97 ms, but 2048 proofs (128by64) 109 ms split against 104 ms
  if (batch < split_min || batch < 2 || rp->is_twin || rp->opt.no_split)
    return prove_batch_one(rp, batch, amounts, types, blinds, rand_prefix, prefix_len, coms_files, proof_files, 0);
  { int rc = rp_ensure_twin(rp); if (rc) return rc; }
  if (rp->comb && !rp->twin->comb) rp->twin->comb = rp->comb;      // not owned by the twin
  rp->twin->opt = rp->opt;
  const size_t nr = rp->st.rds.size(), B0 = (batch + 1) / 2, B1 = batch - B0;
  int rc1 = BPPP_OK;
  std::thread second([&] {
    rc1 = prove_batch_one(rp->twin, B1, amounts + 4 * nr * B0, types + 4 * nr * B0, blinds + 4 * nr * B0, rand_prefix ? rand_prefix + prefix_len * B0 : nullptr, prefix_len,
                          coms_files + (size_t)rp->D.coms_bytes * B0, proof_files + (size_t)rp->D.proof_bytes * B0, B0);
  });
  const int rc0 = prove_batch_one(rp, B0, amounts, types, blinds, rand_prefix, prefix_len, coms_files, proof_files, 0);
  second.join();
  if (rc0) return rc0;
  if (rc1) return fail(ctx, rc1, bppp_last_error(rp->twin_ctx));
  return BPPP_OK;
}

static int prove_batch_one(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                           size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files, size_t index_base) {
  bppp_ctx *ctx = rp->ctx;
  const Setup &st = rp->st;
  uint32_t max_base = 0;
  for (const RangeData &rd : st.rds) max_base = std::max(max_base, rd.base);
  // the device algebra looks digits up in a per-proof table of reciprocals held in LDS (256 entries, up to 2048 for wider digit bases); bases beyond
  // that (and BPPP_RP_HOST_ALGEBRA=1, kept for comparison) take the host-algebra path: same bytes out
  if (max_base > 2048 || (max_base > 256 && st.rds.size() > 256) || rp->opt.host_algebra || (st.flavour && !rp->comb)) return prove_batch_host(rp, batch, amounts, types, blinds, rand_prefix, prefix_len, coms_files, proof_files, index_base);
  hipSetDevice(ctx->device);
  const size_t B = batch, nr = st.rds.size(), nlen = st.nlen, llen = st.llen, k = st.rounds, T = 1 + llen + nlen;
  if (nr >= (1u << 16)) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: too many ranges");
  { int rc = build_fixed_table(rp); if (rc) return rc; }
  if (!rp->commit_basis) { int rc = bppp_basis_create_device(ctx, rp->d_basis, T, 0, 4096, &rp->commit_basis); if (rc) return rc; }
  const bool timing = rp->opt.timing;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now();
  auto lap = [&](const char *what) { if (timing) { double t = now(); fprintf(stderr, "[rp_prove] %-28s %8.2f ms\n", what, t - t_last); t_last = t; } };
  // ---- the witness on the host: digits and multiplicities are integer work on the plain amounts (TypedReciprocal.hs:125-161)
  // host staging: one pinned grow-only buffer per handle (no page faults on fresh vectors every call; the 50 MB of inputs and the
  // commitments cross PCIe at the pinned rate)
  const size_t n_in_sc = B * nr * 12, n_dig = B * nlen, n_mss = B * (llen - 6) + 1, n_in_pt = B * nr * 8;
  const size_t pin_need = (n_in_sc + n_in_pt) * 8 + (2 * n_dig + n_mss) * 4 + 64;
  if (pin_need > rp->hpin_bytes) {
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rp->hpin) BPPP_HIP(ctx, hipHostFree(rp->hpin));
    rp->hpin = nullptr; rp->hpin_bytes = 0;
    BPPP_HIP(ctx, hipHostMalloc(&rp->hpin, pin_need + pin_need / 8, hipHostMallocDefault));
    rp->hpin_bytes = pin_need + pin_need / 8;
  }
  uint64_t *h_in_sc = (uint64_t *)rp->hpin, *h_in_pt = h_in_sc + n_in_sc;
  uint32_t *dig = (uint32_t *)(h_in_pt + n_in_pt), *mul = dig + n_dig, *mss = mul + n_dig;
  std::atomic<int> failed{-1};
  std::vector<std::string> errs(B);
  rp_parallel(B, [&](size_t lo, size_t hi) {
    PState p;
    for (size_t b = lo; b < hi; b++) {
      if (!make_witness(st, p, amounts + 4 * nr * b, types + 4 * nr * b, blinds + 4 * nr * b)) { failed = (int)b; errs[b] = p.err; continue; }
      for (size_t i = 0; i < nr; i++) { p.v[i].store(&h_in_sc[(b * nr + i) * 12]); p.ty[i].store(&h_in_sc[(b * nr + i) * 12 + 4]); p.bl[i].store(&h_in_sc[(b * nr + i) * 12 + 8]); }
      for (size_t i = 0; i < nlen; i++) {
        const bool typing = (st.pos[i].kind & 0xFFu) == bppp_rps::POS_TYPING;
        dig[b * nlen + i] = typing ? 0u : (uint32_t)p.d[i].w[0];
        mul[b * nlen + i] = (uint32_t)p.mi[i].w[0];
      }
      for (size_t j = 0; j + 6 < llen; j++) mss[b * (llen - 6) + j] = (uint32_t)p.ms_shared[j].w[0];
    }
  });
  if (failed >= 0) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: proof " + std::to_string((size_t)failed + index_base) + ": " + errs[failed]);
  lap("witness digits (host)");
  std::vector<uint64_t> c_dm(B * 8), c_m(B * 8), c_r(B * 8), c_bl(B * 8), resp(B * k * 16), wn(B * st.fn * 4 + 4), wl(B * st.fl * 4 + 4);
  RppHostInputs in{B, h_in_sc, dig, mul, mss, rand_prefix, prefix_len};
  RppOutputs out{h_in_pt, c_dm.data(), c_m.data(), c_r.data(), c_bl.data(), resp.data(), wn.data(), wl.data()};
  { int rc = rpp_device_prove(rp, in, out); if (rc) return rc; }
  lap("phases + argument (device)");
  encode_batch(rp, B, out, coms_files, proof_files);
  lap("encode (host)");
  return BPPP_OK;
}

static int prove_batch_host(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                            size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files, size_t index_base) {
  if (!rp) return BPPP_ERR_ARG;
  bppp_ctx *ctx = rp->ctx;
  if (!batch) return BPPP_OK;
  hipSetDevice(ctx->device);
  hipStream_t stream = ctx->stream;
  const Setup &st = rp->st;
  const size_t B = batch, nr = st.rds.size(), nlen = st.nlen, llen = st.llen, k = st.rounds, T = 1 + llen + nlen;
  if (nr >= (1u << 16)) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: too many ranges");
  { int rc = build_fixed_table(rp); if (rc) return rc; }
  // the basis of commitRPW is fixed per setup: registered once with its fixed-base table (one bucket set for all windows)
  if (!rp->commit_basis) { int rc = bppp_basis_create_device(ctx, rp->d_basis, T, 0, 4096, &rp->commit_basis); if (rc) return rc; }
  // device workspace: [input scalars B nr 3 | input commitments B nr | commitment rows 2B T]
  const size_t in_sc = B * nr * 3 * 32, in_pt = B * nr * 64, rows = 2 * B * T * 32;
  { int rc = ensure_pwork(rp, in_sc + in_pt + rows + 1024); if (rc) return rc; }
  uint32_t *d_in_sc = (uint32_t *)rp->pwork, *d_in_pt = (uint32_t *)((char *)rp->pwork + ((in_sc + 255) & ~(size_t)255)),
           *d_rows = (uint32_t *)((char *)d_in_pt + ((in_pt + 255) & ~(size_t)255));

  // BPPP_RP_TIMING=1: wall time of each phase on stderr (tuning aid)
  const bool timing = rp->opt.timing;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now();
  auto lap = [&](const char *what) { if (timing) { double t = now(); fprintf(stderr, "[rp_prove] %-28s %8.2f ms\n", what, t - t_last); t_last = t; } };
  std::vector<PState> ps(B);
  std::vector<uint64_t> h_in_sc(B * nr * 12), h_in_pt(B * nr * 8), h_rows(2 * B * T * 4), h_com(2 * B * 8);
  std::atomic<int> failed{-1};
  auto put_row = [&](size_t row, const RPW &w) {           // commitRPW's term order: [sc] ++ lin ++ nrm over [g] ++ hs ++ gs
    uint64_t *dst = &h_rows[row * T * 4];
    w.sc.store(dst);
    for (size_t i = 0; i < llen; i++) w.lin[i].store(dst + 4 * (1 + i));
    for (size_t i = 0; i < nlen; i++) w.nrm[i].store(dst + 4 * (1 + llen + i));
  };
  auto commit_rows = [&](size_t nrows) -> int {            // one batched MSM over the registered basis: h_com[row] = commit(row)
    BPPP_HIP(ctx, hipMemcpyAsync(d_rows, h_rows.data(), nrows * T * 32, hipMemcpyHostToDevice, stream));
    return bppp_msm_basis(rp->commit_basis, d_rows, T, nrows, h_com.data());
  };

  // ---- phase 1: witness, dmWit / mWit (TypedReciprocal.hs:402-410)
  rp_parallel(B, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; b++) {
      PState &p = ps[b];
      p.rnd = Rnd{rand_prefix + b * prefix_len, prefix_len, 0};
      if (!make_witness(st, p, amounts + 4 * nr * b, types + 4 * nr * b, blinds + 4 * nr * b)) { failed = (int)b; continue; }
      for (size_t i = 0; i < nr; i++) { p.v[i].store(&h_in_sc[(b * nr + i) * 12]); p.ty[i].store(&h_in_sc[(b * nr + i) * 12 + 4]); p.bl[i].store(&h_in_sc[(b * nr + i) * 12 + 8]); }
      p.dm = blind_witness(2, p.ms_shared, p.d, p.rnd, llen);
      p.m = blind_witness(1, std::vector<U256>(), p.mi, p.rnd, llen);
      put_row(2 * b, p.dm); put_row(2 * b + 1, p.m);
    }
  });
  if (failed >= 0) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: proof " + std::to_string((size_t)failed + index_base) + ": " + ps[failed].err);
  lap("phase 1 host");
  BPPP_HIP(ctx, hipMemcpyAsync(d_in_sc, h_in_sc.data(), in_sc, hipMemcpyHostToDevice, stream));
  {
    const uint64_t n = (uint64_t)B * nr;
    k_rp_commit_inputs<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream>>>(rp->d_fixed, d_in_sc, n, d_in_pt);
    BPPP_HIP(ctx, hipGetLastError());
    BPPP_HIP(ctx, hipMemcpyAsync(h_in_pt.data(), d_in_pt, in_pt, hipMemcpyDeviceToHost, stream));
  }
  { int rc = commit_rows(2 * B); if (rc) return rc; }       // synchronises the stream
  lap("inputs + dm/m commits (GPU)");
  std::vector<uint64_t> c_dm(B * 8), c_m(B * 8), c_r(B * 8), c_bl(B * 8);
  for (size_t b = 0; b < B; b++) { memcpy(&c_dm[8 * b], &h_com[16 * b], 64); memcpy(&c_m[8 * b], &h_com[16 * b + 8], 64); }

  // ---- phase 2: (e, x, r0), reciprocals, rWit (:412-419)
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<const uint64_t *> pts(2 + nr);
    for (size_t b = lo; b < hi; b++) {
      PState &p = ps[b];
      pts[0] = &c_dm[8 * b]; pts[1] = &c_m[8 * b];
      for (size_t i = 0; i < nr; i++) pts[2 + i] = &h_in_pt[(b * nr + i) * 8];
      U256 ch[3];
      oracle(rp->tag, p, pts.data(), pts.size(), 3, ch);
      p.e = ch[0]; p.x = ch[1]; p.r0 = ch[2];
      p.e_inv = finv(p.e); p.r0_inv = finv(p.r0);
      make_phase2(st, p);
      U256 s = U256::zero();
      for (size_t i = 0; i < nlen; i++) s = fa(s, fdbl(fm(p.rr[i], p.cc[i])));
      const U256 err7 = fm(p.r0_inv, fneg(s));             // err7Term (:209-211)
      p.r = blind_err_witness(err7, p.rr, p.rnd, llen);
      put_row(b, p.r);
    }
  });
  lap("phase 2 host");
  { int rc = commit_rows(B); if (rc) return rc; }
  memcpy(c_r.data(), h_com.data(), B * 64);
  lap("r commit (GPU)");

  // ---- phase 3: (q, x', r1), error terms, the blinding commitment (:421-437)
  rp_parallel(B, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; b++) {
      PState &p = ps[b];
      const uint64_t *pt = &c_r[8 * b];
      U256 ch[3];
      oracle(rp->tag, p, &pt, 1, 3, ch);
      p.q = ch[0]; p.xp = ch[1]; p.r1 = ch[2];
      p.q0 = fm(p.q, p.q);                                  // qPowers': powers' (q^2) for the NL norm (NormArgument.hs:148),
      if (st.flavour) p.q0 = fneg(p.q0);                    // powers' (-q^2) for the IP one (InnerProductArgument.hs:231)
      p.q0_inv = finv(p.q0); p.r1_inv = finv(p.r1);
      p.shared_cs = make_shared_coeffs(st, p);
      const U256 tC = st.has_types ? p.xp : U256::zero();
      std::vector<U256> bls_lin(llen - 5), bls_nrm(nlen);
      for (auto &v : bls_lin) v = p.rnd.next();
      for (auto &v : bls_nrm) v = p.rnd.next();
      const std::vector<U256> bls_ms(bls_lin.begin() + 1, bls_lin.end());
      const std::vector<U256> ic = input_coeffs(st, p.x, p.q0);
      p.ns_sc = p.ns_ty = p.ns_bl = U256::zero();
      for (size_t i = 0; i < nr; i++) { p.ns_sc = fa(p.ns_sc, fm(ic[i], p.v[i])); p.ns_ty = fa(p.ns_ty, fm(ic[i], p.ty[i])); p.ns_bl = fa(p.ns_bl, fm(ic[i], p.bl[i])); }
      U256 errs[6];
      make_error_terms(st, p, bls_ms, bls_nrm, errs);
      p.blw = blind_blinding_term(bls_lin, bls_nrm, tC, p, errs, p.ns_bl, llen);
      put_row(b, p.blw);
    }
  });
  lap("phase 3 host");
  { int rc = commit_rows(B); if (rc) return rc; }
  memcpy(c_bl.data(), h_com.data(), B * 64);
  lap("bl commit (GPU)");

  // ---- t, the combined witness and the argument's linear weights (:438-446)
  std::vector<uint64_t> a_s(B * 4), a_q(B * 4), a_nx(B * nlen * 4), a_lc(B * llen * 4), a_lx(B * llen * 4);
  rp_parallel(B, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; b++) {
      PState &p = ps[b];
      const uint64_t *pt = &c_bl[8 * b];
      oracle(rp->tag, p, &pt, 1, 1, &p.t);
      U256 psc; std::vector<U256> pn;
      make_public_consts(st, p, psc, pn);
      const U256 t2 = fm(p.t, p.t), t3 = fm(t2, p.t), t4 = fm(t2, t2), t5 = fm(t4, p.t), t6 = fm(t3, t3), two_t5 = fdbl(t5);
      // wit = pub + blWit + t mWit + t^2 dmWit + t^3 rWit + 2 t^5 nWitSum
      U256 sc = fa(fa(psc, p.blw.sc), fa(fa(fm(p.t, p.m.sc), fm(t2, p.dm.sc)), fa(fm(t3, p.r.sc), fm(two_t5, p.ns_sc))));
      sc.store(&a_s[4 * b]); p.q.store(&a_q[4 * b]);
      for (size_t i = 0; i < llen; i++) {
        U256 v = fa(fa(p.blw.lin[i], fm(p.t, p.m.lin[i])), fa(fm(t2, p.dm.lin[i]), fm(t3, p.r.lin[i])));
        if (i == 0) v = fa(v, fm(two_t5, p.ns_ty));
        if (i == 1) v = fa(v, fm(two_t5, p.ns_bl));
        v.store(&a_lx[(b * llen + i) * 4]);
      }
      for (size_t i = 0; i < nlen; i++)
        fa(fa(pn[i], p.blw.nrm[i]), fa(fa(fm(p.t, p.m.nrm[i]), fm(t2, p.dm.nrm[i])), fm(t3, p.r.nrm[i]))).store(&a_nx[(b * nlen + i) * 4]);
      // makeBpCoeffs (:391-396)
      const U256 rs = fm(p.r0, p.r1), two_t3 = fdbl(t3);
      U256 c6[6] = {st.has_types ? fneg(p.xp) : U256::zero(), fm(rs, p.t), fm(rs, t2), fm(rs, t3), fm(p.r0, t4), fm(rs, t6)};
      for (int i = 0; i < 6; i++) c6[i].store(&a_lc[(b * llen + i) * 4]);
      for (size_t i = 0; i < p.shared_cs.size(); i++) fm(two_t3, p.shared_cs[i]).store(&a_lc[(b * llen + 6 + i) * 4]);
      // the phase vectors are no longer needed
      p.dm = RPW(); p.m = RPW(); p.r = RPW(); p.blw = RPW();
      p.u.clear(); p.vv.clear(); p.rr.clear(); p.cc.clear(); p.d.clear(); p.mi.clear(); p.pv.clear();
    }
  });

  lap("witness combination host");
  // ---- proveBPM in lockstep (src/Bulletproof.hs:357-359)
  std::vector<uint64_t> resp(B * (k ? k : 1) * 16), wn(B * st.fn * 4 + 4), wl(B * st.fl * 4 + 4);
  if (st.flavour) {
    int rc = ip_argument_lockstep(rp, B, k, a_s.data(), a_q.data(), a_nx.data(), a_lc.data(), a_lx.data(), [&](size_t b) -> PState & { return ps[b]; }, resp.data(),
                                  wn.data(), wl.data());
    if (rc) return rc;
    lap("inner-product argument");
  } else {
  bppp_nlb *nlb = nullptr;
  int rc = bppp_nlb_create(ctx, B, a_s.data(), rp->h_g.data(), a_q.data(), a_nx.data(), rp->h_G.data(), nlen, a_lc.data(), a_lx.data(), rp->h_H.data(), llen, &nlb);
  if (rc) return rc;
  lap("nlb_create");
  double t_commit = 0, t_hash = 0, t_collapse = 0;
  std::vector<uint64_t> sX(B * 4), sR(B * 4), X(B * 8), R(B * 8), es(B * 4);
  for (size_t round = 0; round < k && !rc; round++) {
    double ta = now();
    rc = bppp_nlb_round_commit(nlb, sX.data(), X.data(), sR.data(), R.data());
    if (rc) break;
    double tb = now(); t_commit += tb - ta;
    rp_parallel(B, [&](size_t lo, size_t hi) {
      for (size_t b = lo; b < hi; b++) {
        const uint64_t *pts[2] = {&X[8 * b], &R[8 * b]};
        U256 e;
        oracle(rp->tag, ps[b], pts, 2, 1, &e);
        e.store(&es[4 * b]);
        const size_t slot = k - 1 - round;                 // responses LAST round first (:359)
        memcpy(&resp[(b * k + slot) * 16], pts[0], 64); memcpy(&resp[(b * k + slot) * 16 + 8], pts[1], 64);
      }
    });
    double tc = now(); t_hash += tc - tb;
    rc = bppp_nlb_round_collapse(nlb, es.data());
    t_collapse += now() - tc;
  }
  if (timing) fprintf(stderr, "[rp_prove] argument: commits %.2f ms, hashing %.2f ms, collapses %.2f ms\n", t_commit, t_hash, t_collapse);
  t_last = now();
  if (!rc) rc = bppp_nlb_get_witness(nlb, wn.data(), wl.data(), nullptr);
  bppp_nlb_destroy(nlb);
  if (rc) return rc;
  }

  // ---- encodeProof' (RangeProof.hs:60-66): commitments file = the input commitments; proof file = final witness scalars (norm, linear),
  // then blCom, rCom, dmCom, mCom and the responses
  const RpDims &D = rp->D;
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<const uint64_t *> pts;
    for (size_t b = lo; b < hi; b++) {
      pts.assign(nr, nullptr);
      for (size_t i = 0; i < nr; i++) pts[i] = &h_in_pt[(b * nr + i) * 8];
      encode_points(coms_files + b * D.coms_bytes, pts.data(), nr);
      uint8_t *pf = proof_files + b * D.proof_bytes;
      for (size_t i = 0; i < st.fn; i++) put_field(pf + 32 * i, U256::load(&wn[(b * st.fn + i) * 4]));
      for (size_t i = 0; i < st.fl; i++) put_field(pf + 32 * (st.fn + i), U256::load(&wl[(b * st.fl + i) * 4]));
      pts.assign(4 + 2 * k, nullptr);
      pts[0] = &c_bl[8 * b]; pts[1] = &c_r[8 * b]; pts[2] = &c_dm[8 * b]; pts[3] = &c_m[8 * b];
      for (size_t j = 0; j < 2 * k; j++) pts[4 + j] = &resp[(b * k) * 16 + 8 * j];
      encode_points(pf + 32 * (st.fn + st.fl), pts.data(), 4 + 2 * k);
    }
  });
  lap("witness download + encode");
  return BPPP_OK;
}

