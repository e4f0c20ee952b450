// foldcore.hip.h — one basis-fold pair per lane:  out = b' * GL + a' * GR  for 129-bit reduced scalars (b', a').
//
// The group element is the reference's `projectivePairIP (b', gL) (a', gR)` (src/Commitment.hs:343-353, reached from
// collapsePoints, src/Bulletproof.hs:213-214): a 129-row double-and-add over the two points.  Any addition chain that
// evaluates b'*GL + a'*GR yields the same canonical affine point, so the device walks the pair in JOINT SPARSE FORM
// (Solinas): signed digits (u_b, u_a) in {-1,0,1}^2 per row, recoded on the host (hostmath.hpp jsf_recode), at most one
// mixed addition per row and on average one every second row (the binary form needs 1 per row on average), against the
// per-lane table {GL, GR, GL+GR, GL-GR}.  The two sums share the denominator (xR - xL), so the table costs ONE inversion.
//
// The table lives in LDS (16 canonical words per point, [entry][word][lane], bank = lane), not in registers: the hot loop
// then holds only the XYZZ accumulator and one addend, and because the addend is picked PER LANE, lanes of one wavefront
// may belong to different folds (different proofs / different scalar pairs): short folds are packed into full wavefronts.
#pragma once
#include "ec.hip.h"

namespace bppp {

static constexpr int FOLD_ROWS = 130;          // JSF of two 129-bit magnitudes has at most 130 digits
static constexpr int FOLD_DIGIT_WORDS = 17;    // 4 bits per row: [1:0] = u_b code, [3:2] = u_a code (0: 0, 1: +1, 3: -1)
static constexpr int FOLD_TAB_WORDS = 4 * 16 * 64;   // LDS words per wavefront

BPPP_DI void fold_tab_put(uint32_t *tab, int entry, uint32_t lane, const aff &p) {   // p canonical (magnitude 1, normalized)
  fe x = fq_to_fe(p.x), y = fq_to_fe(p.y);
#pragma unroll
  for (int k = 0; k < 8; k++) { tab[((entry * 16 + k) << 6) + lane] = x.v[k]; tab[((entry * 16 + 8 + k) << 6) + lane] = y.v[k]; }
}
BPPP_DI aff fold_tab_get(const uint32_t *tab, uint32_t entry, uint32_t lane) {
  fe x, y;
#pragma unroll
  for (int k = 0; k < 8; k++) { x.v[k] = tab[((entry * 16 + k) << 6) + lane]; y.v[k] = tab[((entry * 16 + 8 + k) << 6) + lane]; }
  aff r; r.x = fq_from_fe(x); r.y = fq_from_fe(y);
  return r;
}

// GL, GR: canonical affine (sign of b', a' already folded in by aff_cneg, so y may have magnitude 2); `dig` points at this
// lane's 17 digit words (any address space; wave-uniform or per lane); `tab` is the wavefront's LDS table.
// Every lane of the wavefront must call this (inactive pairs pass GL = GR = infinity and any valid `dig`).
BPPP_DI aff fold_pair_jsf(const aff &GL_, const aff &GR_, const uint32_t *dig, uint32_t *tab, uint32_t lane) {
  aff GL, GR;
  GL.x = GL_.x; GL.y = fq_normalize(GL_.y); GR.x = GR_.x; GR.y = fq_normalize(GR_.y);
  const bool linf = aff_is_inf(GL), rinf = aff_is_inf(GR);
  // Table entries S = GL + GR, D = GL - GR.  Chord through GL and (+-)GR: both slopes over the ONE inverted denominator
  // xR - xL.  If xR = xL (GR = +-GL: never for an honest basis, kept so every input follows the group law) one of the two
  // is the tangent at GL (slope 3x^2 / 2y through the same inversion) and the other is infinity.
  fq dx = fq_sub<1>(GR.x, GL.x);                                 // magnitude 3
  const bool same_x = fq_normalizes_to_zero(dx);
  const bool same_y = fq_normalizes_to_zero(fq_sub<1>(GR.y, GL.y));
  fq den = fq_mul_int(GL.y, 2);
  fq tan_num = fq_mul_int(fq_sqr(GL.x), 3);                      // 3
  fq num_s = fq_sub<1>(GR.y, GL.y);                              // 3
  fq num_d = fq_neg<2>(fq_add(GR.y, GL.y));                      // 3
#pragma unroll
  for (int i = 0; i < 10; i++) {
    den.n[i] = same_x ? den.n[i] : dx.n[i];
    num_s.n[i] = same_x ? tan_num.n[i] : num_s.n[i];
    num_d.n[i] = same_x ? tan_num.n[i] : num_d.n[i];
  }
  fq inv = fq_inv(den);
  fq ls = fq_mul(num_s, inv), ld = fq_mul(num_d, inv);
  fq sx = fq_add(GL.x, GR.x);                                    // 2
  aff S, D;
  S.x = fq_sub<2>(fq_sqr(ls), sx);                               // 4
  S.y = fq_normalize(fq_sub<1>(fq_mul(ls, fq_sub<4>(GL.x, S.x)), GL.y));
  S.x = fq_normalize(S.x);
  D.x = fq_sub<2>(fq_sqr(ld), sx);
  D.y = fq_normalize(fq_sub<1>(fq_mul(ld, fq_sub<4>(GL.x, D.x)), GL.y));
  D.x = fq_normalize(D.x);
  {
    aff nGR = aff_cneg(GR, true); nGR.y = fq_normalize(nGR.y);
    const bool s_inf = !linf && !rinf && same_x && !same_y, d_inf = !linf && !rinf && same_x && same_y;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      uint32_t sxv = S.x.n[i], syv = S.y.n[i], dxv = D.x.n[i], dyv = D.y.n[i];
      if (rinf) { sxv = GL.x.n[i]; syv = GL.y.n[i]; dxv = GL.x.n[i]; dyv = GL.y.n[i]; }
      else if (linf) { sxv = GR.x.n[i]; syv = GR.y.n[i]; dxv = nGR.x.n[i]; dyv = nGR.y.n[i]; }
      if (s_inf) { sxv = 0; syv = 0; }
      if (d_inf) { dxv = 0; dyv = 0; }
      S.x.n[i] = sxv; S.y.n[i] = syv; D.x.n[i] = dxv; D.y.n[i] = dyv;
    }
  }
  fold_tab_put(tab, 0, lane, GL); fold_tab_put(tab, 1, lane, GR); fold_tab_put(tab, 2, lane, S); fold_tab_put(tab, 3, lane, D);
  // no barrier: every lane reads back only what it wrote itself

  xyzz acc = xyzz_inf();
  uint32_t w = 0;
  for (int row = FOLD_ROWS - 1; row >= 0; row--) {
    if (row == FOLD_ROWS - 1 || (row & 7) == 7) w = dig[row >> 3];
    const uint32_t nib = (w >> (4 * (row & 7))) & 15u;
    acc = xyzz_dbl(acc);
    if (nib) {
      const uint32_t ub = nib & 3u, ua = nib >> 2;
      // u_b GL + u_a GR  =  +-GL | +-GR | +-(GL+GR) | +-(GL-GR)
      const uint32_t entry = ua == 0 ? 0u : ub == 0 ? 1u : ub == ua ? 2u : 3u;
      const bool neg = entry == 1 ? ua == 3u : ub == 3u;
      aff T = fold_tab_get(tab, entry, lane);
      T = aff_cneg(T, neg);
      xyzz_madd(acc, T);
    }
  }
  return xyzz_to_aff(acc);
}

}  // namespace bppp
