// ipb_host.hip — the inner-product argument for B proofs in lockstep with its field algebra on the HOST cores (round 3's route): every
// commitment an MSM over the registered original basis on the GPU, the O(nrmLen + linLen) Fr work of a round and the hashing on host threads.
// Kept as the cross-check of the device-resident csrc/ipb.hip (BPPP_RP_HOST_ALGEBRA=1) and for batches before a handle has its comb table.
#include <atomic>
#include "rpprove_host.hpp"

using namespace bppp;
using namespace bppp_rpp;

namespace bppp {
// ------------------------------------------------------------------------------------------------ inner-product flavour, lockstep
// proveBPM (src/Bulletproof.hs:357-359) of src/Bulletproof/InnerProductArgument.hs for B arguments of one shape, WITHOUT one basis change
// and WITHOUT one point fold.  Two observations:
//   (1) makeNorm's basis (:194-206) g'_j = g_2j+1 + r g_2j, h'_j = g_2j+1 - r g_2j enters every commitment linearly:
//       A g'_j + B h'_j = (A + B) g_2j+1 + r (A - B) g_2j  — so a commitment over the transformed basis is an MSM over the ORIGINAL one;
//   (2) collapse (:86-101, :162-170) folds a pair of points with the reduced fraction (a', b') of rho = 1/(q e) (resp. e, 1/e) and scales the
//       scalars by 1/b0 so that the products scalar x point depend on rho only.  Tracking, per original position i, the product coef_i of
//       the rho's of the right halves i fell into, the level-k basis point at position p is sum_{i >> k = p} coef_i P_i and every round
//       commitment is again an MSM over the original points with scalars sc_{i >> k} coef_i — the same group elements as the folding
//       route, hence the same L, R and (the normalisations cancel: nx x and ny y are what getWitness :222-223 emits) the same final witness.
// Per round and proof: O(nrmLen + linLen) Fr multiplications on the host cores, two rows of 1 + linLen + nrmLen scalars; all 2B rows are ONE
// batched MSM over the registered basis [g | H | G] (fixed-base table, one bucket set per instance).  The oracle is the setup's shaOracle.
// in: psv [B] (the PSV scalar), rr [B] (makeNorm's r), nrm [B][nlen], lc / lx [B][llen]; trs: the proofs' transcripts so far
// out: resp [B][k][16] (L, R per round, LAST round first), wn [B][fn], wl [B][fl]
int ip_argument_lockstep(bppp_rp *rp, size_t B, size_t k, const uint64_t *psv_in, const uint64_t *rr, const uint64_t *nrm, const uint64_t *lc_in,
                                const uint64_t *lx_in, const std::function<PState &(size_t)> &tr_of, uint64_t *resp, uint64_t *wn, uint64_t *wl) {
  bppp_ctx *ctx = rp->ctx;
  const Setup &st = rp->st;
  const size_t nlen = st.nlen, llen = st.llen, m0 = (nlen + 1) / 2, T = 1 + llen + nlen;
  struct IpState {
    std::vector<U256> X, Y, LC, LX, cx, cy, cl;       // current vectors (unscaled) and the per-original-position coefficient products
    U256 r, q, qinv, NX, psv, sL, sR;
  };
  std::vector<IpState> sts(B);
  const U256 half = finv(small(2)), four = small(4);
  rp_parallel(B, [&](size_t lo, size_t hi) {
    for (size_t b = lo; b < hi; b++) {
      IpState &p = sts[b];
      p.r = U256::load(rr + 4 * b);
      const U256 r2 = fm(p.r, p.r), r2i = finv(fdbl(p.r));
      p.q = fm(r2, r2); p.qinv = finv(p.q); p.NX = U256::one(); p.psv = U256::load(psv_in + 4 * b);
      p.X.resize(m0); p.Y.resize(m0); p.cx.assign(m0, U256::one()); p.cy.assign(m0, U256::one());
      for (size_t j = 0; j < m0; j++) {                   // makeNorm (:202-203): x' = s0 / (2r) + s1 / 2, y' = -s0 / (2r) + s1 / 2
        const U256 s0 = U256::load(nrm + 4 * (b * nlen + 2 * j)), s1 = 2 * j + 1 < nlen ? U256::load(nrm + 4 * (b * nlen + 2 * j + 1)) : U256::zero();
        const U256 a = fm(r2i, s0), c = fm(half, s1);
        p.X[j] = fa(a, c); p.Y[j] = fs(c, a);
      }
      p.LC.resize(llen); p.LX.resize(llen); p.cl.assign(llen, U256::one());
      for (size_t i = 0; i < llen; i++) { p.LC[i] = U256::load(lc_in + 4 * (b * llen + i)); p.LX[i] = U256::load(lx_in + 4 * (b * llen + i)); }
    }
  });
  { int rc = rpp_ensure_pwork(rp, 2 * B * T * 32 + 1024); if (rc) return rc; }
  uint32_t *d_rows = (uint32_t *)rp->pwork;
  std::vector<uint64_t> h_rows(2 * B * T * 4), h_com(2 * B * 8);
  for (size_t round = 0; round < k; round++) {
    // ---- makeScalarsComs (:70-81, :155-158; BPCompose sums the two sub-arguments, Bulletproof.hs:258-261) and the two rows
    rp_parallel(B, [&](size_t lo, size_t hi) {
      std::vector<U256> lgx, lhy, rgx, rhy, ll, rl;       // opening scalars per CURRENT position: on g', h' and the linear basis
      for (size_t b = lo; b < hi; b++) {
        IpState &p = sts[b];
        const size_t mc = p.X.size(), lcn = p.LX.size();
        const U256 q2 = fm(p.q, p.q);
        lgx.assign(mc + (mc & 1), U256::zero()); lhy = lgx; rgx = lgx; rhy = lgx;
        U256 w = U256::one(), l = U256::zero(), r_ = U256::zero();
        for (size_t t = 0; 2 * t < mc; t++) {
          const U256 xL = p.X[2 * t], yL = p.Y[2 * t], xR = 2 * t + 1 < mc ? p.X[2 * t + 1] : U256::zero(), yR = 2 * t + 1 < mc ? p.Y[2 * t + 1] : U256::zero();
          l = fa(l, fm(w, fm(xL, yR))); r_ = fa(r_, fm(w, fm(xR, yL)));
          lgx[2 * t + 1] = fm(p.qinv, xL); lhy[2 * t] = yR;      // L: IPF (qInv xL) gR yR hL
          rgx[2 * t] = fm(p.q, xR); rhy[2 * t + 1] = yL;         // R: IPF (q xR) gL yL hR
          w = fm(w, q2);
        }
        const U256 kk = fm(four, p.NX);                          // s nx ny with s = 4 (makeNorm), ny = 1 in the unscaled recursion
        U256 sL = mc ? fm(fm(kk, p.q), l) : U256::zero(), sR = mc ? fm(fm(kk, q2), r_) : U256::zero();
        ll.assign(lcn + (lcn & 1), U256::zero()); rl = ll;
        for (size_t t = 0; 2 * t < lcn; t++) {
          const U256 cL = p.LC[2 * t], xL = p.LX[2 * t], cR = 2 * t + 1 < lcn ? p.LC[2 * t + 1] : U256::zero(), xR = 2 * t + 1 < lcn ? p.LX[2 * t + 1] : U256::zero();
          sL = fa(sL, fm(cR, xL)); sR = fa(sR, fm(cL, xR));
          ll[2 * t + 1] = xL; rl[2 * t] = xR;                    // L: LF cR xL gR;  R: LF cL xR gL
        }
        p.sL = sL; p.sR = sR;
        for (int side = 0; side < 2; side++) {
          uint64_t *row = &h_rows[(2 * b + side) * T * 4];
          const std::vector<U256> &gx = side ? rgx : lgx, &hy = side ? rhy : lhy, &lv = side ? rl : ll;
          (side ? sR : sL).store(row);
          for (size_t i = 0; i < llen; i++) fm(lv[i >> round], p.cl[i]).store(row + 4 * (1 + i));
          for (size_t i = 0; i < m0; i++) {
            const size_t pos = i >> round;
            const U256 A = fm(gx[pos], p.cx[i]), Bv = fm(hy[pos], p.cy[i]);
            fm(p.r, fs(A, Bv)).store(row + 4 * (1 + llen + 2 * i));                       // on g_2i
            if (2 * i + 1 < nlen) fa(A, Bv).store(row + 4 * (1 + llen + 2 * i + 1));        // on g_2i+1 (absent for an odd tail: infinity in makeNorm)
          }
        }
      }
    });
    BPPP_HIP(ctx, hipMemcpyAsync(d_rows, h_rows.data(), 2 * B * T * 32, hipMemcpyHostToDevice, ctx->stream));
    { int rc = bppp_msm_basis(rp->commit_basis, d_rows, T, 2 * B, h_com.data()); if (rc) return rc; }
    // ---- the challenge, s += e0 sL + e1 sR with makeEs e = (1/e, e) (:68), collapse
    std::atomic<int> bad{0};
    rp_parallel(B, [&](size_t lo, size_t hi) {
      for (size_t b = lo; b < hi; b++) {
        IpState &p = sts[b];
        const uint64_t *pts[2] = {&h_com[16 * b], &h_com[16 * b + 8]};
        U256 e;
        oracle(rp->tag, tr_of(b), pts, 2, 1, &e);
        if (e.is_zero()) { bad = 1; continue; }
        const U256 ei = finv(e);
        const size_t slot = k - 1 - round;                     // responses LAST round first (Bulletproof.hs:359)
        memcpy(resp + (b * k + slot) * 16, pts[0], 64); memcpy(resp + (b * k + slot) * 16 + 8, pts[1], 64);
        p.psv = fa(p.psv, fa(fm(ei, p.sL), fm(e, p.sR)));
        const size_t mc = p.X.size(), lcn = p.LX.size();
        if (mc) {
          const U256 eq = fm(e, p.q), rhox = fm(p.qinv, ei);
          std::vector<U256> nx((mc + 1) / 2), ny((mc + 1) / 2);
          for (size_t t = 0; 2 * t < mc; t++) {
            const bool has = 2 * t + 1 < mc;
            nx[t] = has ? fa(p.X[2 * t], fm(eq, p.X[2 * t + 1])) : p.X[2 * t];
            ny[t] = has ? fa(p.Y[2 * t], fm(ei, p.Y[2 * t + 1])) : p.Y[2 * t];
          }
          p.X.swap(nx); p.Y.swap(ny);
          for (size_t i = 0; i < m0; i++) if ((i >> round) & 1) { p.cx[i] = fm(p.cx[i], rhox); p.cy[i] = fm(p.cy[i], e); }
          p.NX = fm(p.NX, p.qinv);
          p.q = fm(p.q, p.q); p.qinv = fm(p.qinv, p.qinv);
        }
        if (lcn) {
          std::vector<U256> nc((lcn + 1) / 2), nxl((lcn + 1) / 2);
          for (size_t t = 0; 2 * t < lcn; t++) {
            const bool has = 2 * t + 1 < lcn;
            nc[t] = has ? fa(p.LC[2 * t], fm(ei, p.LC[2 * t + 1])) : p.LC[2 * t];
            nxl[t] = has ? fa(p.LX[2 * t], fm(e, p.LX[2 * t + 1])) : p.LX[2 * t];
          }
          p.LC.swap(nc); p.LX.swap(nxl);
          for (size_t i = 0; i < llen; i++) if ((i >> round) & 1) p.cl[i] = fm(p.cl[i], ei);
        }
      }
    });
    if (bad) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: a round challenge is zero");
  }
  // ---- getWitness: Norm (nx x - ny y, nx x + ny y) (:222-223), Linear nrmlz x (:160)
  for (size_t b = 0; b < B; b++) {
    const IpState &p = sts[b];
    if (2 * p.X.size() != st.fn || p.LX.size() != st.fl) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: the argument ended at an unexpected length");
    for (size_t j = 0; j < p.X.size(); j++) {
      const U256 a = fm(p.NX, p.X[j]);
      fs(a, p.Y[j]).store(wn + 4 * (b * st.fn + 2 * j)); fa(a, p.Y[j]).store(wn + 4 * (b * st.fn + 2 * j + 1));
    }
    for (size_t i = 0; i < p.LX.size(); i++) p.LX[i].store(wl + 4 * (b * st.fl + i));
  }
  return BPPP_OK;
}

}  // namespace bppp
