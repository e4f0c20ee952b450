// brpprove.hip — RangeProof.Binary's lockstep prover behind bppp_rp_prove_batch: the host-algebra route (round 3) and the host wrapper of the
// device-resident route (csrc/brpprove_dev.hip, round 4).  Both write the same bytes (tests/test_gpu_native_binary.py).
#include <atomic>
#include <chrono>
#include <stdio.h>
#include "rpprove_host.hpp"
#include "rpprove_dev.hpp"

using namespace bppp;
using namespace bppp_rpp;

namespace bppp {

// ------------------------------------------------------------------------------------------------ RangeProof.Binary
// proveM for B binary range proofs of one setup in lockstep: proveBRPM (src/RangeProof/Binary.hs:169-204) then proveBPM
// (src/Bulletproof.hs:357-359) and encodeProof' (src/RangeProof.hs:60-66).  Work split as in prove_batch_host above: every group
// operation on the device — input commitments v g + bl h0 through the fixed-base table of (g, h0, h1), the digit and blinding
// commitments of all proofs as batched MSMs over the registered basis [g | h0 h1 | G], the argument through csrc/nlb.hip — the
// O(nrmLen) field algebra of a proof and its transcript hashing on the host cores (one proof per thread slice).
int prove_batch_binary(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *blinds, const uint8_t *rand_prefix, size_t prefix_len,
                              uint8_t *coms_files, uint8_t *proof_files) {
  bppp_ctx *ctx = rp->ctx;
  hipSetDevice(ctx->device);
  hipStream_t stream = ctx->stream;
  const Setup &st = rp->st;
  const size_t B = batch, nr = st.rds.size(), nlen = st.nlen, nlive = st.nlive, llen = 2, k = st.rounds, T = 1 + llen + nlen;
  const bool timing = rp->opt.timing;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = timing ? now() : 0;
  auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[rp_prove binary] %-28s %8.2f ms\n", what, t - t_last); t_last = t; } };
  { int rc = rpp_build_fixed_table(rp); if (rc) return rc; }
  if (!rp->commit_basis) { int rc = bppp_basis_create_device(ctx, rp->d_basis, T, 0, 4096, &rp->commit_basis); if (rc) return rc; }
  const size_t in_sc = B * nr * 3 * 32, in_pt = B * nr * 64, rows = B * T * 32;
  { int rc = rpp_ensure_pwork(rp, in_sc + in_pt + rows + 1024); if (rc) return rc; }
  uint32_t *d_in_sc = (uint32_t *)rp->pwork, *d_in_pt = (uint32_t *)((char *)rp->pwork + ((in_sc + 255) & ~(size_t)255)),
           *d_rows = (uint32_t *)((char *)d_in_pt + ((in_pt + 255) & ~(size_t)255));
  struct BState {
    std::vector<U256> v, bl, ds, bls, pub_nrm;
    U256 s_bl, l_bl0, bl_bl, q, x, r, t, q0, bl0_sc, lin1, pub_sc;
    PState tr;                                            // transcript text + randomness (oracle() works on a PState)
    std::string err;
  };
  std::vector<BState> ps(B);
  std::vector<uint64_t> h_in_sc(B * nr * 12), h_in_pt(B * nr * 8), h_rows(B * T * 4), h_com(B * 8), c_d(B * 8), c_bl(B * 8);
  std::atomic<int> failed{-1};
  auto put_row = [&](size_t row, const U256 &sc, const U256 &l0, const U256 &l1, const std::vector<U256> &nrm) {
    uint64_t *dst = &h_rows[row * T * 4];
    memset(dst, 0, T * 32);
    sc.store(dst); l0.store(dst + 4); l1.store(dst + 8);
    for (size_t i = 0; i < nrm.size(); i++) nrm[i].store(dst + 4 * (3 + i));
  };
  auto commit_rows = [&]() -> int {
    BPPP_HIP(ctx, hipMemcpyAsync(d_rows, h_rows.data(), B * T * 32, hipMemcpyHostToDevice, stream));
    return bppp_msm_basis(rp->commit_basis, d_rows, T, B, h_com.data());
  };
  // ---- witnessBRP (:158-166) and phase 1 (:171-178): digits, the digit commitment's scalars
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<uint32_t> dg;
    for (size_t b = lo; b < hi; b++) {
      BState &p = ps[b];
      p.tr.rnd = Rnd{rand_prefix + b * prefix_len, prefix_len, 0};
      p.v.resize(nr); p.bl.resize(nr); p.ds.clear();
      U256 vsum = st.net_public;
      bool ok = true;
      for (size_t i = 0; i < nr && ok; i++) {
        const RangeData &rd = st.rds[i];
        const U256 amt = U256::load(amounts + 4 * (b * nr + i));
        p.bl[i] = U256::load(blinds + 4 * (b * nr + i));
        if (!scalars_canonical(blinds + 4 * (b * nr + i), 1)) { p.err = "blinding is not canonical"; ok = false; break; }
        p.v[i] = bppp_rps::s_mod_n(amt);
        vsum = rd.output ? fs(vsum, p.v[i]) : fa(vsum, p.v[i]);
        if (rd.assumed) continue;
        if (bppp_rps::s_lt(amt, rd.lo) || !bppp_rps::s_lt(amt, rd.hi)) { p.err = "value outside its range"; ok = false; break; }
        bppp_rps::digits_binary_into(rd, bppp_rps::u_sub(amt, rd.lo), dg);
        for (uint32_t d : dg) p.ds.push_back(small(d));
      }
      if (ok && !(st.conserve && vsum.is_zero())) { p.err = "a binary witness needs a conserved schema whose amounts balance (Binary.hs:162-164)"; ok = false; }
      if (!ok) { failed = (int)b; continue; }
      for (size_t i = 0; i < nr; i++) {                     // scalarRPW' (Internal.hs:56-57): v g + bl h0
        p.v[i].store(&h_in_sc[(b * nr + i) * 12]); p.bl[i].store(&h_in_sc[(b * nr + i) * 12 + 4]); U256::zero().store(&h_in_sc[(b * nr + i) * 12 + 8]);
      }
      p.s_bl = p.tr.rnd.next(); p.l_bl0 = p.tr.rnd.next();
      put_row(b, p.s_bl, p.l_bl0, U256::zero(), p.ds);
    }
  });
  if (failed >= 0) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: proof " + std::to_string((size_t)failed) + ": " + ps[failed].err);
  lap("witness, digits (host)");
  BPPP_HIP(ctx, hipMemcpyAsync(d_in_sc, h_in_sc.data(), in_sc, hipMemcpyHostToDevice, stream));
  {
    int rc_ = rpp_commit_inputs(rp, d_in_sc, B * nr, d_in_pt); if (rc_) return rc_;      // the comb table when the handle has one, the fixed-base table otherwise
    BPPP_HIP(ctx, hipMemcpyAsync(h_in_pt.data(), d_in_pt, in_pt, hipMemcpyDeviceToHost, stream));
  }
  { int rc = commit_rows(); if (rc) return rc; }            // dCom of every proof; synchronises the stream
  memcpy(c_d.data(), h_com.data(), B * 64);
  lap("input commitments, dCom");
  // ---- (q, x, r), makePublicConsts, the blinding commitment (:179-189)
  const std::vector<bool> is_o = [&] { std::vector<bool> v_; for (const RangeData &rd : st.rds) v_.push_back(rd.output); return v_; }();
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<const uint64_t *> pts(1 + nr);
    for (size_t b = lo; b < hi; b++) {
      BState &p = ps[b];
      pts[0] = &c_d[8 * b];
      for (size_t i = 0; i < nr; i++) pts[1 + i] = &h_in_pt[(b * nr + i) * 8];
      U256 ch[3];
      oracle(rp->tag, p.tr, pts.data(), pts.size(), 3, ch);
      p.q = ch[0]; p.x = ch[1]; p.r = ch[2];
      p.q0 = fm(p.q, p.q);                                  // qPowers': powers' (q^2) for the norm-linear argument (NormArgument.hs:148),
      if (st.flavour) p.q0 = fneg(p.q0);                    // powers' (-q^2) for the inner-product one (InnerProductArgument.hs:231)
      const U256 q0i = finv(p.q0), r_inv = finv(p.r), xx = fm(p.x, p.x), half = finv(small(2));
      // makePublicConsts (:73-98)
      std::vector<U256> x2s(nr);
      { U256 c = xx; for (size_t j = 0; j < nr; j++) { x2s[j] = c; c = fm(c, xx); } }
      U256 z = st.conserve ? fneg(fm(p.x, st.net_public)) : U256::zero();
      for (size_t j = 0; j < nr; j++) if (!st.rds[j].assumed) z = fa(z, fm(bppp_rps::s_mod_n(st.rds[j].lo), x2s[j]));
      U256 sc = fneg(fdbl(z));
      p.pub_nrm.resize(nlive);
      U256 q2 = p.q0, q2i = q0i;
      for (size_t i = 0; i < nlive; i++) {
        const U256 pv = fs(fm(fm(x2s[st.pos[i].range], st.pos[i].coeff), q2i), half);
        sc = fa(sc, fm(q2, fm(pv, pv)));
        p.pub_nrm[i] = pv;
        q2 = fm(q2, p.q0); q2i = fm(q2i, q0i);
      }
      p.pub_sc = sc;
      p.bls.resize(nlen);
      for (auto &v_ : p.bls) v_ = p.tr.rnd.next();
      p.bl_bl = p.tr.rnd.next();
      // makePolyTerms (Internal.hs:69-80) of |bls + T (ds + pub)|^2_q: the constant and the linear coefficient
      U256 w = p.q0, bl0 = U256::zero(), bl1 = U256::zero();
      for (size_t i = 0; i < nlen; i++) {
        bl0 = fa(bl0, fm(w, fm(p.bls[i], p.bls[i])));
        if (i < nlive) bl1 = fa(bl1, fm(w, fm(p.bls[i], fa(p.ds[i], p.pub_nrm[i]))));
        w = fm(w, p.q0);
      }
      p.bl0_sc = bl0;
      p.lin1 = fm(r_inv, fs(p.s_bl, fdbl(bl1)));
      put_row(b, p.bl0_sc, p.bl_bl, p.lin1, p.bls);
    }
  });
  lap("q x r, public consts, bls (host)");
  { int rc = commit_rows(); if (rc) return rc; }
  memcpy(c_bl.data(), h_com.data(), B * 64);
  lap("blCom");
  // ---- t and the argument's witness (:190-201)
  std::vector<uint64_t> a_s(B * 4), a_q(B * 4), a_nx(B * nlen * 4), a_lc(B * llen * 4), a_lx(B * llen * 4);
  rp_parallel(B, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; b++) {
      BState &p = ps[b];
      const uint64_t *pt = &c_bl[8 * b];
      oracle(rp->tag, p.tr, &pt, 1, 1, &p.t);
      const U256 xx = fm(p.x, p.x), two_t = fdbl(p.t);
      U256 x2 = xx, icv = U256::zero(), icb = U256::zero();
      for (size_t j = 0; j < nr; j++) {                     // inputCoeffs (:127-129)
        U256 ic = st.rds[j].assumed ? U256::zero() : x2;
        if (st.conserve) ic = st.rds[j].output ? fs(ic, p.x) : fa(ic, p.x);
        icv = fa(icv, fm(ic, p.v[j])); icb = fa(icb, fm(ic, p.bl[j]));
        x2 = fm(x2, xx);
      }
      // bpWit = blWit + t (pub' + dWit + 2 t sum_j ic_j nWit_j),  pub' = (t pubSc; pubNrm)
      fa(p.bl0_sc, fm(p.t, fa(fa(fm(p.t, p.pub_sc), p.s_bl), fm(two_t, icv)))).store(&a_s[4 * b]);
      p.q.store(&a_q[4 * b]);
      fa(p.bl_bl, fm(p.t, fa(p.l_bl0, fm(two_t, icb)))).store(&a_lx[(b * 2) * 4]);
      p.lin1.store(&a_lx[(b * 2 + 1) * 4]);
      for (size_t i = 0; i < nlen; i++) {
        U256 v_ = p.bls[i];
        if (i < nlive) v_ = fa(v_, fm(p.t, fa(p.pub_nrm[i], p.ds[i])));
        v_.store(&a_nx[(b * nlen + i) * 4]);
      }
      U256::zero().store(&a_lc[(b * 2) * 4]); fm(p.r, p.t).store(&a_lc[(b * 2 + 1) * 4]);      // setupBRP's cs' = [0, r t] (:152)
      p.bls.clear(); p.pub_nrm.clear(); p.ds.clear();
    }
  });
  lap("t, argument witness (host)");
  // ---- proveBPM in lockstep
  std::vector<uint64_t> resp(B * (k ? k : 1) * 16), wn(B * st.fn * 4 + 4), wl(B * st.fl * 4 + 4);
  if (st.flavour) {
    int rc = ip_argument_lockstep(rp, B, k, a_s.data(), a_q.data(), a_nx.data(), a_lc.data(), a_lx.data(), [&](size_t b) -> PState & { return ps[b].tr; }, resp.data(),
                                  wn.data(), wl.data());
    if (rc) return rc;
  } else {
    // (measured and not kept: the fixed-basis mode over a comb table of the binary setup's 4099 points — 21.5 GB at c = 13 — takes the
    // argument of 1024 64x64-bit proofs from 167 to 157 ms: its rounds are host round trips either way)
    bppp_nlb *nlb = nullptr;
    int rc = bppp_nlb_create(ctx, B, a_s.data(), rp->h_g.data(), a_q.data(), a_nx.data(), rp->h_G.data(), nlen, a_lc.data(), a_lx.data(), rp->h_H.data(), llen, &nlb);
    if (rc) return rc;
    std::vector<uint64_t> sX(B * 4), sR(B * 4), X(B * 8), R(B * 8), es(B * 4);
    for (size_t round = 0; round < k && !rc; round++) {
      rc = bppp_nlb_round_commit(nlb, sX.data(), X.data(), sR.data(), R.data());
      if (rc) break;
      rp_parallel(B, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; b++) {
          const uint64_t *pts[2] = {&X[8 * b], &R[8 * b]};
          U256 e;
          oracle(rp->tag, ps[b].tr, pts, 2, 1, &e);
          e.store(&es[4 * b]);
          const size_t slot = k - 1 - round;                 // responses LAST round first (Bulletproof.hs:359)
          memcpy(&resp[(b * k + slot) * 16], pts[0], 64); memcpy(&resp[(b * k + slot) * 16 + 8], pts[1], 64);
        }
      });
      rc = bppp_nlb_round_collapse(nlb, es.data());
    }
    if (!rc) rc = bppp_nlb_get_witness(nlb, wn.data(), wl.data(), nullptr);
    bppp_nlb_destroy(nlb);
    if (rc) return rc;
  }
  lap("argument (lockstep)");
  // ---- encodeProof': commitments file = the input commitments; proof file = final witness scalars, then blCom, dCom and the responses
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
      pts.assign(2 + 2 * k, nullptr);
      pts[0] = &c_bl[8 * b]; pts[1] = &c_d[8 * b];
      for (size_t j = 0; j < 2 * k; j++) pts[2 + j] = &resp[(b * k) * 16 + 8 * j];
      encode_points(pf + 32 * (st.fn + st.fl), pts.data(), 2 + 2 * k);
    }
  });
  return BPPP_OK;
}

// The same proofs with proveBRPM's field algebra, randomness and transcript on the device (csrc/brpprove_dev.hip): the host checks the
// witness (witnessBRP, Binary.hs:158-166), extracts the binary digits of the plain amounts (makeDigits :56-69) and writes the files.
int prove_batch_binary_dev(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *blinds, const uint8_t *rand_prefix, size_t prefix_len,
                                  uint8_t *coms_files, uint8_t *proof_files, size_t index_base) {
  bppp_ctx *ctx = rp->ctx;
  hipSetDevice(ctx->device);
  const Setup &st = rp->st;
  const size_t B = batch, nr = st.rds.size(), nlive = st.nlive, k = st.rounds;
  const bool timing = rp->opt.timing;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = timing ? now() : 0;
  auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[rp_prove binary] %-28s %8.2f ms\n", what, t - t_last); t_last = t; } };
  // pinned staging, grow-only: [in_sc B nr 3 | input commitments B nr | bits B nlive]
  const size_t n_in_sc = B * nr * 12, n_in_pt = B * nr * 8;
  const size_t pin_need = (n_in_sc + n_in_pt) * 8 + B * nlive + 64;
  if (pin_need > rp->hpin_bytes) {
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rp->hpin) BPPP_HIP(ctx, hipHostFree(rp->hpin));
    rp->hpin = nullptr; rp->hpin_bytes = 0;
    BPPP_HIP(ctx, hipHostMalloc(&rp->hpin, pin_need + pin_need / 8, hipHostMallocDefault));
    rp->hpin_bytes = pin_need + pin_need / 8;
  }
  uint64_t *h_in_sc = (uint64_t *)rp->hpin, *h_in_pt = h_in_sc + n_in_sc;
  uint8_t *bits = (uint8_t *)(h_in_pt + n_in_pt);
  std::atomic<int> failed{-1};
  std::vector<std::string> errs(B);
  rp_parallel(B, [&](size_t lo, size_t hi) {
    std::vector<uint32_t> dg;
    for (size_t b = lo; b < hi; b++) {
      U256 vsum = st.net_public;
      const char *err = nullptr;
      size_t p = 0;
      for (size_t i = 0; i < nr && !err; i++) {
        const RangeData &rd = st.rds[i];
        const U256 amt = U256::load(amounts + 4 * (b * nr + i));
        if (!scalars_canonical(blinds + 4 * (b * nr + i), 1)) { err = "blinding is not canonical"; break; }
        const U256 v = bppp_rps::s_mod_n(amt);
        vsum = rd.output ? fs(vsum, v) : fa(vsum, v);
        uint64_t *row = &h_in_sc[(b * nr + i) * 12];          // scalarRPW' (Internal.hs:56-57): v g + bl h0
        v.store(row); memcpy(row + 4, blinds + 4 * (b * nr + i), 32); memset(row + 8, 0, 32);
        if (rd.assumed) continue;
        if (bppp_rps::s_lt(amt, rd.lo) || !bppp_rps::s_lt(amt, rd.hi)) { err = "value outside its range"; break; }
        bppp_rps::digits_binary_into(rd, bppp_rps::u_sub(amt, rd.lo), dg);
        for (uint32_t d : dg) bits[b * nlive + p++] = (uint8_t)d;
      }
      if (!err && !(st.conserve && vsum.is_zero())) err = "a binary witness needs a conserved schema whose amounts balance (Binary.hs:162-164)";
      if (!err && p != nlive) err = "digit count disagrees with the setup";
      if (err) { failed = (int)b; errs[b] = err; }
    }
  });
  if (failed >= 0) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: proof " + std::to_string(index_base + (size_t)failed) + ": " + errs[failed]);
  lap("witness, digits (host)");
  std::vector<uint64_t> c_d(B * 8), c_bl(B * 8), resp(B * (k ? k : 1) * 16), wn(B * st.fn * 4 + 4), wl(B * st.fl * 4 + 4);
  BrpHostInputs in{B, h_in_sc, bits, rand_prefix, prefix_len};
  BrpOutputs out{h_in_pt, c_d.data(), c_bl.data(), resp.data(), wn.data(), wl.data()};
  { int rc = brp_device_prove(rp, in, out); if (rc) return rc; }
  lap("phases + argument (device)");
  // encodeProof': commitments file = the input commitments; proof file = final witness scalars, then blCom, dCom and the responses
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
      pts.assign(2 + 2 * k, nullptr);
      pts[0] = &c_bl[8 * b]; pts[1] = &c_d[8 * b];
      for (size_t j = 0; j < 2 * k; j++) pts[2 + j] = &resp[(b * k) * 16 + 8 * j];
      encode_points(pf + 32 * (st.fn + st.fl), pts.data(), 2 + 2 * k);
    }
  });
  lap("encode (host)");
  return BPPP_OK;
}

}  // namespace bppp
