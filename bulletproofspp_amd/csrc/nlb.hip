// nlb.hip — B norm-linear arguments (NL flavour) proved in LOCKSTEP: every kernel of a round carries the batch dimension.
//
// Same functions as nl.hip (proveRoundM, src/Bulletproof.hs:346-355; makeScalarsComs, src/Bulletproof/NormArgument.hs:113-118,
// :56-59; collapse :123-129, :64-71) applied to `batch` independent proofs of one shape at once.  One proof's round is far
// too small to fill 256 CUs (its MSMs have < 800 terms and its 130-row basis fold is a ~1 ms dependency chain), so a
// single proof is latency-bound; B proofs share the launches: 2B round commitments are ONE batched MSM (each proof's X and R
// share that proof's basis), B x ceil(n/2) basis folds are ONE launch with per-proof reduced scalars, and the Fr vector work
// is one workgroup per proof.  The bases diverge after the first collapse (different challenges), so they are stored per proof.
//
// Two routes.  The POINT-FOLDING route above is the general one (bppp_nlb_create; small range-proof batches).  With a comb table of
// the starting basis attached (csrc/comb.hip; the range-proof prover's large batches) the argument runs in FIXED-BASIS mode: no
// point is ever folded — the fold coefficients are multiplied into the scalars and every round commitment is a comb MSM over the
// original points — and the per-proof round state (q, q^-1, n, l, s, sX, sR) lives in HBM, so a round is a stream of kernels
// (nlb_round_commit_dev / nlb_round_collapse_dev; the bppp_nlb_round_* entry points wrap them).  Same group elements either way.
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "ctx.hpp"
#include "comb.hpp"
#include "foldcore.hip.h"
#include "fr26.hip.h"
#include "hostmath.hpp"

namespace bppp {
int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);

BPPP_DI fe frb_pow(fe base, uint32_t e) {
  fe acc = fe_one();
  while (e) { if (e & 1u) acc = fe_mul<1>(acc, base); base = fe_sqr<1>(base); e >>= 1; }
  return acc;
}
BPPP_DI void block_sum4(fe v[4], uint32_t *lds) {
  const int t = threadIdx.x;
  for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) lds[(t * 4 + k) * 8 + i] = v[k].v[i];
  __syncthreads();
  for (int d = (int)blockDim.x >> 1; d >= 1; d >>= 1) {      // blockDim.x is a power of two (64 .. 256)
    if (t < d) {
      for (int k = 0; k < 4; k++) {
        fe x, y;
        for (int i = 0; i < 8; i++) { x.v[i] = lds[(t * 4 + k) * 8 + i]; y.v[i] = lds[((t + d) * 4 + k) * 8 + i]; }
        x = fe_add<1>(x, y);
        for (int i = 0; i < 8; i++) lds[(t * 4 + k) * 8 + i] = x.v[i];
      }
    }
    __syncthreads();
  }
  if (t == 0) for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) v[k].v[i] = lds[k * 8 + i];
}

// one workgroup per proof: the four round sums (norm sX', sR'; linear sX, sR) and the X / R opening scalars over the
// even-padded basis slices (layout as in nl.hip): scX = [q xR, qinv xL ... | xR, xL ...], scR = [0, xR ... | 0, xR ...]
__global__ void __launch_bounds__(256) k_nlb_round(const uint32_t *__restrict__ x, const uint32_t *__restrict__ lc, const uint32_t *__restrict__ lx,
                                                   uint32_t n, uint32_t l, uint32_t xstride, uint32_t lstride, const uint32_t *__restrict__ qs /*[B][2]: q, qinv*/,
                                                   uint32_t T, uint32_t *__restrict__ sc /*[2B][T]*/, uint32_t *__restrict__ sums /*[B][4]*/) {
  // Fr in 10 x 26-bit limbs (fr26.hip.h: 413 instructions per multiplication against 785); canonical 8 x 32 values in memory
  __shared__ uint32_t lds[256 * 4 * 8];
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const uint32_t ne = n + (n & 1), np = (n + 1) / 2, lp = (l + 1) / 2;
  const fr q = fr_load(qs + (size_t)b * 16), qinv = fr_load(qs + (size_t)b * 16 + 8);
  const fr q2 = fr_sqr(q), q4 = fr_sqr(q2);
  const uint32_t bs = blockDim.x;                  // 64 .. 256: late rounds have a handful of pairs per proof
  auto powu = [](fr base, uint32_t e) { fr acc = fr_one(); while (e) { if (e & 1u) acc = fr_mul(acc, base); base = fr_sqr(base); e >>= 1; } return acc; };
  fr w = powu(q4, t);
  const fr step = powu(q4, bs);
  uint32_t *scX = sc + (size_t)(2 * b) * T * 8, *scR = sc + (size_t)(2 * b + 1) * T * 8;
  const uint32_t *xb = x + (size_t)b * xstride * 8, *lcb = lc + (size_t)b * lstride * 8, *lxb = lx + (size_t)b * lstride * 8;
  fr s0 = fr_zero(), s1 = s0, s2 = s0, s3 = s0;
  for (uint32_t j = t; j < np; j += bs) {
    const fe xl8 = fe_load(xb + (size_t)(2 * j) * 8);
    const fe xr8 = (2 * j + 1 < n) ? fe_load(xb + (size_t)(2 * j + 1) * 8) : fe_zero();
    const fr xl = fr_from_fe(xl8), xr = fr_from_fe(xr8);
    const fr wxr = fr_mul(w, xr);
    s0 = fr_addr(s0, fr_mul(wxr, xl));
    s1 = fr_addr(s1, fr_mul(wxr, xr));
    fr_store(scX + (size_t)(2 * j) * 8, fr_mul(q, xr));
    fr_store(scX + (size_t)(2 * j + 1) * 8, fr_mul(qinv, xl));
    fe_store(scR + (size_t)(2 * j) * 8, fe_zero());
    fe_store(scR + (size_t)(2 * j + 1) * 8, xr8);
    w = fr_mul(w, step);
  }
  for (uint32_t j = t; j < lp; j += bs) {
    const bool has = 2 * j + 1 < l;
    const fe xl8 = fe_load(lxb + (size_t)(2 * j) * 8), xr8 = has ? fe_load(lxb + (size_t)(2 * j + 1) * 8) : fe_zero();
    const fr cl = fr_load(lcb + (size_t)(2 * j) * 8), xl = fr_from_fe(xl8);
    const fr cr = has ? fr_load(lcb + (size_t)(2 * j + 1) * 8) : fr_zero();
    const fr xr = fr_from_fe(xr8);
    s2 = fr_addr(s2, fr_add(fr_mul(cl, xr), fr_mul(cr, xl)));
    s3 = fr_addr(s3, fr_mul(cr, xr));
    fe_store(scX + (size_t)(ne + 2 * j) * 8, xr8);
    fe_store(scX + (size_t)(ne + 2 * j + 1) * 8, xl8);
    fe_store(scR + (size_t)(ne + 2 * j) * 8, fe_zero());
    fe_store(scR + (size_t)(ne + 2 * j + 1) * 8, xr8);
  }
  fe s[4] = {fr_to_fe(s0), fr_to_fe(s1), fr_to_fe(s2), fr_to_fe(s3)};
  block_sum4(s, lds);
  if (t == 0) for (int k = 0; k < 4; k++) fe_store(sums + ((size_t)b * 4 + k) * 8, s[k]);
}

// per-proof collapse constants: FoldK for the two basis folds and (u, v) for the three scalar folds
struct CollapseK {
  uint32_t dn[FOLD_DIGIT_WORDS], dl[FOLD_DIGIT_WORDS];   // joint-sparse-form digits of (b', a') for the norm fold and the linear fold
  int nbneg, naneg, lbneg, laneg;
  uint32_t nu[8], nv[8];                 // norm x' = nu xL + nv xR
  uint32_t cu[8], cv[8];                 // linear c' = cu cL + cv cR
  uint32_t lu[8], lv[8];                 // linear x' = lu xL + lv xR
};
BPPP_DI fe fe_of8(const uint32_t *p) { fe r; for (int i = 0; i < 8; i++) r.v[i] = p[i]; return r; }

__global__ void __launch_bounds__(256) k_nlb_fold_scalars(const uint32_t *__restrict__ x, const uint32_t *__restrict__ lc, const uint32_t *__restrict__ lx,
                                                          uint32_t n, uint32_t l, uint32_t xstride, uint32_t lstride, const CollapseK *__restrict__ K,
                                                          uint32_t *__restrict__ xo, uint32_t *__restrict__ lco, uint32_t *__restrict__ lxo) {
  const uint32_t b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  const CollapseK &k = K[b];
  if (j < (n + 1) / 2) {
    const uint32_t *xb = x + (size_t)b * xstride * 8;
    fr r = fr_mul(fr_from_fe(fe_of8(k.nu)), fr_load(xb + (size_t)(2 * j) * 8));
    if (2 * j + 1 < n) r = fr_add(r, fr_mul(fr_from_fe(fe_of8(k.nv)), fr_load(xb + (size_t)(2 * j + 1) * 8)));
    fr_store(xo + ((size_t)b * xstride + j) * 8, r);
  }
  if (j < (l + 1) / 2) {
    const uint32_t *cb = lc + (size_t)b * lstride * 8, *xb = lx + (size_t)b * lstride * 8;
    bool has = 2 * j + 1 < l;
    fr c = fr_mul(fr_from_fe(fe_of8(k.cu)), fr_load(cb + (size_t)(2 * j) * 8));
    fr r = fr_mul(fr_from_fe(fe_of8(k.lu)), fr_load(xb + (size_t)(2 * j) * 8));
    if (has) {
      c = fr_add(c, fr_mul(fr_from_fe(fe_of8(k.cv)), fr_load(cb + (size_t)(2 * j + 1) * 8)));
      r = fr_add(r, fr_mul(fr_from_fe(fe_of8(k.lv)), fr_load(xb + (size_t)(2 * j + 1) * 8)));
    }
    fr_store(lco + ((size_t)b * lstride + j) * 8, c);
    fr_store(lxo + ((size_t)b * lstride + j) * 8, r);
  }
}

// basis folds of all proofs: the pairs [norm | linear] of every proof are numbered consecutively over the whole batch, so
// wavefronts are full in every round (a late round has a handful of pairs per proof); each lane walks the digit schedule of
// its own proof and fold (foldcore.hip.h)
__global__ void __launch_bounds__(64) k_nlb_fold_points(const uint32_t *__restrict__ P, uint32_t n, uint32_t l, uint32_t cap, uint32_t cap_out,
                                                        const CollapseK *__restrict__ K, uint32_t batch, uint32_t *__restrict__ Po) {
  __shared__ uint32_t tab[FOLD_TAB_WORDS];
  const uint32_t n2 = (n + 1) / 2, l2 = (l + 1) / 2, pp = n2 + l2;
  const uint64_t g = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  const bool active = g < (uint64_t)batch * pp;
  const uint32_t b = active ? (uint32_t)(g / pp) : 0, r = active ? (uint32_t)(g % pp) : 0;
  const bool lin = r >= n2;
  const uint32_t cnt = lin ? l : n, j = lin ? r - n2 : r;
  const CollapseK &k = K[b];
  const uint32_t ne = n + (n & 1), ne2 = n2 + (n2 & 1);
  aff GL = aff_inf(), GR = aff_inf();
  if (active) {
    const uint32_t *src = P + ((size_t)b * cap + (lin ? ne : 0)) * 16;
    GL = aff_cneg(aff_load(src + (size_t)(2 * j) * 16), (lin ? k.lbneg : k.nbneg) != 0);
    if (2 * j + 1 < cnt) GR = aff_cneg(aff_load(src + (size_t)(2 * j + 1) * 16), (lin ? k.laneg : k.naneg) != 0);
  }
  aff res = fold_pair_jsf(GL, GR, lin ? k.dl : k.dn, tab, threadIdx.x);
  if (active) aff_store(Po + ((size_t)b * cap_out + (lin ? ne2 : 0) + j) * 16, res);
}
// g (and the infinity padding) of every proof carried into the next layout
__global__ void k_nlb_move_g(const uint32_t *__restrict__ P, uint32_t cap, uint32_t cap_out, uint32_t src_off, uint32_t dst_off, uint32_t *__restrict__ Po) {
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  if (t < 16) Po[((size_t)b * cap_out + dst_off) * 16 + t] = P[((size_t)b * cap + src_off) * 16 + t];
}

// ---- fixed-basis mode (a comb table over the setup's [g | H | G] is attached, csrc/comb.hip): the points are NEVER folded.
// After r folds the basis point at level position p is  G^(r)_p = sum_{i >> r == p} coef_i G_i  with coef_i the product of the
// fold coefficients of the halves index i fell into (left: 1, right: a = e q^-1 for the norm part, a = e for the linear part —
// the unscaled fold, i.e. (a, b) = (x, 1) where the point-folding route uses the reduced fraction of rationalReduceScalar), so a
// round commitment  sum_p sc_p G^(r)_p  is the MSM over the ORIGINAL points with scalars sc_(i >> r) coef_i: one comb MSM per
// round, no point fold, no half-GCD.  The same group elements X, R (hence the same proof bytes) as the folding route.
// full[inst][0] = g's scalar, [1 .. l0] = linear part, [1 + l0 ..] = norm part (the comb table's order)
__global__ void __launch_bounds__(256) k_nlb_expand(const uint32_t *__restrict__ sc, uint32_t Tr, uint32_t ne_r, uint32_t r, const uint32_t *__restrict__ coefn,
                                                    const uint32_t *__restrict__ coefl, uint32_t n0, uint32_t l0, uint32_t *__restrict__ full) {
  const uint32_t inst = blockIdx.y, b = inst >> 1, pos = blockIdx.x * 256 + threadIdx.x, Tc = 1 + l0 + n0;
  if (pos >= Tc) return;
  const uint32_t *row = sc + (size_t)inst * Tr * 8;
  fe v;
  if (pos == 0) v = fe_load(row + (size_t)(Tr - 1) * 8);
  else if (pos <= l0) {
    const uint32_t i = pos - 1;
    v = fe_load(row + (size_t)(ne_r + (i >> r)) * 8);
    if (r && !fe_is_zero(v)) v = fr_to_fe(fr_mul(fr_from_fe(v), fr_load(coefl + ((size_t)b * l0 + i) * 8)));
  } else {
    const uint32_t i = pos - 1 - l0;
    v = fe_load(row + (size_t)(i >> r) * 8);
    if (r && !fe_is_zero(v)) v = fr_to_fe(fr_mul(fr_from_fe(v), fr_load(coefn + ((size_t)b * n0 + i) * 8)));
  }
  fe_store(full + ((size_t)inst * Tc + pos) * 8, v);
}
// coef_i *= a for the indices that are RIGHT halves at fold r; A[b] = (a_norm, a_lin)
__global__ void __launch_bounds__(256) k_nlb_coef_update(uint32_t *__restrict__ coefn, uint32_t *__restrict__ coefl, uint32_t n0, uint32_t l0, uint32_t r,
                                                         const uint32_t *__restrict__ A) {
  const uint32_t b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < n0 && ((i >> r) & 1u)) {
    uint32_t *c = coefn + ((size_t)b * n0 + i) * 8;
    fr_store(c, fr_mul(fr_load(c), fr_load(A + (size_t)b * 16)));
  }
  if (i < l0 && ((i >> r) & 1u)) {
    uint32_t *c = coefl + ((size_t)b * l0 + i) * 8;
    fr_store(c, fr_mul(fr_load(c), fr_load(A + (size_t)b * 16 + 8)));
  }
}
// rows [b][1 + l0 + n0] = [0 | coefl_b | coefn_b]: the fold coefficients of a proof in the comb table's order (the scalars of comb_groups)
__global__ void __launch_bounds__(256) k_nlb_coef_rows(const uint32_t *__restrict__ coefn, const uint32_t *__restrict__ coefl, uint32_t n0, uint32_t l0, uint32_t *__restrict__ rows) {
  const uint32_t b = blockIdx.y, pos = blockIdx.x * 256 + threadIdx.x, Tc = 1 + l0 + n0;
  if (pos >= Tc) return;
  fe v = fe_zero();
  if (pos >= 1 && pos <= l0) v = fe_load(coefl + ((size_t)b * l0 + (pos - 1)) * 8);
  else if (pos > l0) v = fe_load(coefn + ((size_t)b * n0 + (pos - 1 - l0)) * 8);
  fe_store(rows + ((size_t)b * Tc + pos) * 8, v);
}
// slot 0 of every proof's materialised basis: g
__global__ void __launch_bounds__(64) k_nlb_set_g(const uint32_t *__restrict__ g, uint32_t batch, uint32_t stride, uint32_t *__restrict__ pb) {
  const uint32_t b = blockIdx.x * 4 + threadIdx.x / 16, t = threadIdx.x % 16;
  if (b < batch) pb[(size_t)b * stride * 16 + t] = g[t];
}
__global__ void __launch_bounds__(256) k_nlb_fill_one(uint32_t *__restrict__ v, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) fe_store(v + i * 8, fe_one());
}
// Round bookkeeping of the fixed-basis mode, one lane per proof, state resident in HBM (no host round trip inside a round):
// st = [batch][NLB_ST][8]: nn, ln, s, sX, sR;  qs = [batch][2][8]: q, q^-1.
enum { NLB_ST_NN = 0, NLB_ST_LN = 1, NLB_ST_S = 2, NLB_ST_SX = 3, NLB_ST_SR = 4, NLB_ST = 5 };
// after k_nlb_round: sX = 2 n^2 q^3 sX' + (linear sX), sR = n^2 q^4 sR' + (linear sR) (NormArgument.hs:113); they are the scalars on g
__global__ void __launch_bounds__(64) k_nlb_tails(const uint32_t *__restrict__ sums, const uint32_t *__restrict__ qs, uint32_t *__restrict__ stt, uint32_t batch,
                                                  uint32_t n, uint32_t l, uint32_t T, uint32_t *__restrict__ sc) {
  const uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  uint32_t *S = stt + (size_t)b * NLB_ST * 8;
  const fe q = fe_load(qs + (size_t)b * 16), nn = fe_load(S + NLB_ST_NN * 8);
  const fe q2 = fe_sqr<1>(q), q3 = fe_mul<1>(q2, q), q4 = fe_sqr<1>(q2), n2 = fe_sqr<1>(nn);
  fe sX = fe_zero(), sR = fe_zero();
  if (n) {
    sX = fe_mul<1>(fe_mul<1>(fe_add<1>(n2, n2), q3), fe_load(sums + (size_t)b * 32));
    sR = fe_mul<1>(fe_mul<1>(n2, q4), fe_load(sums + (size_t)b * 32 + 8));
  }
  if (l) { sX = fe_add<1>(sX, fe_load(sums + (size_t)b * 32 + 16)); sR = fe_add<1>(sR, fe_load(sums + (size_t)b * 32 + 24)); }
  fe_store(S + NLB_ST_SX * 8, sX); fe_store(S + NLB_ST_SR * 8, sR);
  fe_store(sc + ((size_t)(2 * b) * T + T - 1) * 8, sX); fe_store(sc + ((size_t)(2 * b + 1) * T + T - 1) * 8, sR);
}
// start state from q and s of every proof where they lie in HBM: qs = (q, q^-1), st = (n = 1, l = 1, s, 0, 0)
__global__ void __launch_bounds__(64) k_nlb_init_state(const uint32_t *__restrict__ q, const uint32_t *__restrict__ s, uint32_t batch, uint32_t *__restrict__ qs,
                                                       uint32_t *__restrict__ stt) {
  const uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  const fe qq = fe_load(q + (size_t)b * 8);
  fe_store(qs + (size_t)b * 16, qq); fe_store(qs + (size_t)b * 16 + 8, fe_modinv<1>(qq));      // division steps (modinv.hip.h): ~14 k instructions against ~340 k of the Fermat chain
  uint32_t *S = stt + (size_t)b * NLB_ST * 8;
  fe_store(S + NLB_ST_NN * 8, fe_one()); fe_store(S + NLB_ST_LN * 8, fe_one()); fe_store(S + NLB_ST_S * 8, fe_load(s + (size_t)b * 8));
  fe_store(S + NLB_ST_SX * 8, fe_zero()); fe_store(S + NLB_ST_SR * 8, fe_zero());
}
// the challenge e of a round: s' = s + e sX + (e^2 - 1) sR; the unscaled fold constants; n' = n q^-1, q' = q^2
__global__ void __launch_bounds__(64) k_nlb_collapse_state(const uint32_t *__restrict__ es, uint32_t *__restrict__ qs, uint32_t *__restrict__ stt, uint32_t batch,
                                                           uint32_t n, uint32_t l, CollapseK *__restrict__ K, uint32_t *__restrict__ A) {
  const uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  uint32_t *S = stt + (size_t)b * NLB_ST * 8;
  const fe e = fe_load(es + (size_t)b * 8), one = fe_one();
  const fe e1 = fe_sub<1>(fe_sqr<1>(e), one);
  fe_store(S + NLB_ST_S * 8, fe_add<1>(fe_load(S + NLB_ST_S * 8), fe_add<1>(fe_mul<1>(e, fe_load(S + NLB_ST_SX * 8)), fe_mul<1>(e1, fe_load(S + NLB_ST_SR * 8)))));
  CollapseK &k = K[b];
  if (n) {
    const fe q = fe_load(qs + (size_t)b * 16), qi = fe_load(qs + (size_t)b * 16 + 8);
    for (int i = 0; i < 8; i++) k.nu[i] = one.v[i];
    const fe nv = fe_mul<1>(e, q);
    for (int i = 0; i < 8; i++) k.nv[i] = nv.v[i];
    fe_store(A + (size_t)b * 16, fe_mul<1>(e, qi));
    fe_store(S + NLB_ST_NN * 8, fe_mul<1>(fe_load(S + NLB_ST_NN * 8), qi));
    fe_store(qs + (size_t)b * 16, fe_sqr<1>(q)); fe_store(qs + (size_t)b * 16 + 8, fe_sqr<1>(qi));
  }
  if (l) {
    for (int i = 0; i < 8; i++) { k.cu[i] = one.v[i]; k.cv[i] = e.v[i]; k.lu[i] = one.v[i]; k.lv[i] = e.v[i]; }
    fe_store(A + (size_t)b * 16 + 8, e);
  }
}
}  // namespace bppp

namespace bppp { int msm_batch_dev(bppp_ctx *ctx, const void *d_scalars, const void *d_points, size_t n, size_t batch, int shared_points, int window_bits, uint32_t *d_out); }
using namespace bppp;
using namespace bppp_host;


// per-proof host arithmetic of a round (two half-GCDs and a few Fr products each) spread over the host cores: at B in the
// thousands it is otherwise as long as the round's GPU work
template <class F> static void parallel_ranges(size_t n, F f) {     // f(lo, hi) on disjoint ranges covering [0, n)
  unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 64);
  if (nt <= 1) { f((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++) th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

struct bppp_nlb {
  bppp_ctx *ctx;
  size_t batch, n, l, n0, l0, cap, xstride, lstride;   // cap: allocation per proof; the CURRENT point stride is even(n)+even(l)+1
  uint32_t *x[2], *lx[2], *lc[2], *P[2];
  uint32_t *sc, *sums, *qs;
  CollapseK *dK;
  const bppp::CombTable *comb; // fixed-basis mode: comb over [g | lin | norm]; the points are never folded (P[] is not allocated)
  uint32_t *coefn, *coefl, *full, *dA, *d_out, *stt, *d_es, *cscratch;   // stt: per-proof round state in HBM (k_nlb_tails / k_nlb_collapse_state)
  uint32_t folds;              // completed folds (the level shift of an original index)
  // re-basing (long bases): after `rebase_level` folds the level basis of every proof is materialised once (comb_groups) and the remaining rounds commit
  // by bucket MSMs over these per-proof points — Tm = 1 + l0r + n0r terms instead of 1 + l0 + n0 table walks
  uint32_t rebase_level, folds0; bool rebased; size_t n0r, l0r;
  uint32_t *pbasis, *coefn_r, *coefl_r, *full_r;
  int cur;
  std::vector<U256> q, qinv, nn, ln, s, sX, sR;
};
static size_t evb(size_t v) { return v + (v & 1); }
#define NLB_HIP(o, call)                                                                                   \
  do {                                                                                                     \
    hipError_t _e = (call);                                                                                \
    if (_e != hipSuccess) return bppp::fail((o)->ctx, BPPP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" {

void bppp_nlb_destroy(bppp_nlb *o) {
  if (!o) return;
  bppp_ctx *ctx = o->ctx;                  // kept alive by this handle's reference even after bppp_ctx_destroy
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 2; k++) { hipFree(o->x[k]); hipFree(o->lx[k]); hipFree(o->lc[k]); hipFree(o->P[k]); }
  hipFree(o->sc); hipFree(o->sums); hipFree(o->qs); hipFree(o->dK);
  hipFree(o->pbasis); hipFree(o->coefn_r); hipFree(o->coefl_r); hipFree(o->full_r);
  hipFree(o->coefn); hipFree(o->coefl); hipFree(o->full); hipFree(o->dA); hipFree(o->d_out); hipFree(o->stt); hipFree(o->d_es); hipFree(o->cscratch);
  delete o;
  ctx_release(ctx);
}

// `batch` x makeNormLinearBP' 1 q_b cs_b nss_b ngs lss_b lgs (NormArgument.hs:162) inside makePSV s_b g: the basis (g, G, H) is
// shared by all proofs at the start; scalars are [batch][...] host arrays.
}  // extern "C"
namespace bppp {
// copy one basis (words4 uint4) to every proof's slot
__global__ void __launch_bounds__(256) k_nlb_broadcast(const uint4 *__restrict__ src, uint32_t words4, uint4 *__restrict__ dst, uint32_t batch) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= words4) return;
  uint4 v = src[i];
  for (uint32_t b = blockIdx.y; b < batch; b += gridDim.y) dst[(size_t)b * words4 + i] = v;
}
}  // namespace bppp
namespace bppp {
// on_device: every array argument (scalars AND basis points) is already resident in HBM (the batch range-proof prover builds the
// start state of the argument on the device, csrc/rpprove_dev.hip); otherwise they are host arrays (the C ABI entry point)
int nlb_create_impl(bppp_ctx *ctx, size_t batch, const uint64_t *s, const uint64_t *g_xy, const uint64_t *q, const uint64_t *norm_x,
                    const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy, size_t llen,
                    bppp_nlb **out, bool on_device, const CombTable *comb) {
  if (comb && comb->T != 1 + llen + nlen) comb = nullptr;        // not this basis: the general route
  const hipMemcpyKind KIND = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (!ctx || !out || !s || !g_xy || !q || !batch || ctx_closed(ctx)) return BPPP_ERR_ARG;
  if ((nlen && (!norm_x || !norm_g_xy)) || (llen && (!lin_c || !lin_x || !lin_h_xy)) || nlen + llen == 0 || nlen >= (1u << 24) || llen >= (1u << 24) ||
      batch >= (1u << 20))
    return fail(ctx, BPPP_ERR_ARG, "nlb_create: bad arguments");
  hipSetDevice(ctx->device);
  const Mod &M = FR();
  bppp_nlb *o = new bppp_nlb();
  o->ctx = ctx; ctx_retain(ctx); o->batch = batch; o->n = o->n0 = nlen; o->l = o->l0 = llen; o->cur = 0;
  o->cap = evb(nlen) + evb(llen) + 1; o->xstride = evb(nlen) + 2; o->lstride = evb(llen) + 2;
  for (int k = 0; k < 2; k++) { o->x[k] = o->lx[k] = o->lc[k] = o->P[k] = nullptr; }
  o->sc = o->sums = o->qs = nullptr; o->dK = nullptr;
  o->comb = comb; o->coefn = o->coefl = o->full = o->dA = o->d_out = o->stt = o->d_es = o->cscratch = nullptr; o->folds = 0;
  o->rebase_level = 0; o->folds0 = 0; o->rebased = false; o->n0r = o->l0r = 0; o->pbasis = o->coefn_r = o->coefl_r = o->full_r = nullptr;
  if (comb && on_device) {
    // BPPP_NLB_REBASE=<level> forces it (0: never); by default a basis of >= 2048 points is re-based at the first level of <= 528 points when the batch
    // fills a bucket-MSM launch (>= 384 proofs) (64 x 64-bit binary proofs, 4099 points, 1024 proofs: level 2 / 3 / 4 / 5 = 89 / 76 / 80 / 85 ms against 100 without)
    const char *e = getenv("BPPP_NLB_REBASE");
    auto terms_at = [&](uint32_t L) { return 1 + ((llen + ((size_t)1 << L) - 1) >> L) + ((nlen + ((size_t)1 << L) - 1) >> L); };
    if (e) o->rebase_level = (uint32_t)atoi(e);
    else if (nlen + llen >= 2048 && batch >= 384) {      // 64 / 128 / 256 / 512 binary proofs: 22.3 / 31.5 / 38.7 / 64.9 ms without, 26.5 / 32.6 / 39.2 / 54.2 ms with
      uint32_t L = 1;
      while (L < 20 && terms_at(L) > 528) L++;
      o->rebase_level = L;
    }
    if (o->rebase_level > 20 || 2 * batch <= 4) o->rebase_level = 0;
    if (o->rebase_level) { o->l0r = (llen + ((size_t)1 << o->rebase_level) - 1) >> o->rebase_level; o->n0r = (nlen + ((size_t)1 << o->rebase_level) - 1) >> o->rebase_level; }
  }
  bool bad = false;
  for (int k = 0; k < 2; k++) {
    bad |= hipMalloc(&o->x[k], batch * o->xstride * 32) != hipSuccess || hipMalloc(&o->lx[k], batch * o->lstride * 32) != hipSuccess;
    bad |= hipMalloc(&o->lc[k], batch * o->lstride * 32) != hipSuccess || (!comb && hipMalloc(&o->P[k], batch * o->cap * 64) != hipSuccess);
  }
  bad |= hipMalloc(&o->sc, 2 * batch * o->cap * 32) != hipSuccess || hipMalloc(&o->sums, batch * 4 * 32) != hipSuccess;
  bad |= hipMalloc(&o->qs, batch * 64) != hipSuccess || hipMalloc(&o->dK, batch * sizeof(CollapseK)) != hipSuccess;
  if (comb) {
    const size_t Tc = 1 + llen + nlen;
    bad |= hipMalloc(&o->coefn, batch * std::max<size_t>(nlen, 1) * 32) != hipSuccess || hipMalloc(&o->coefl, batch * std::max<size_t>(llen, 1) * 32) != hipSuccess;
    bad |= hipMalloc(&o->full, 2 * batch * Tc * 32) != hipSuccess || hipMalloc(&o->dA, batch * 64) != hipSuccess || hipMalloc(&o->d_out, 2 * batch * 64) != hipSuccess;
    bad |= hipMalloc(&o->stt, batch * NLB_ST * 32) != hipSuccess || hipMalloc(&o->d_es, batch * 32) != hipSuccess;
    bad |= hipMalloc(&o->cscratch, comb_rows_scratch_bytes(2 * batch)) != hipSuccess;
    if (o->rebase_level) {
      const size_t Tm = 1 + o->l0r + o->n0r;
      bad |= hipMalloc(&o->pbasis, batch * Tm * 64) != hipSuccess || hipMalloc(&o->full_r, 2 * batch * Tm * 32) != hipSuccess;
      bad |= hipMalloc(&o->coefn_r, batch * std::max<size_t>(o->n0r, 1) * 32) != hipSuccess || hipMalloc(&o->coefl_r, batch * std::max<size_t>(o->l0r, 1) * 32) != hipSuccess;
    }
  }
  if (bad) { bppp_nlb_destroy(o); return fail(ctx, BPPP_ERR_HIP, "nlb_create: hipMalloc failed"); }
  hipStream_t st = ctx->stream;
  auto fill = [&]() -> int {                 // any failure below goes through ONE cleanup: the handle is destroyed
  if (!comb) NLB_HIP(o, hipMemsetAsync(o->P[0], 0, batch * o->cap * 64, st));
  NLB_HIP(o, hipMemsetAsync(o->x[0], 0, batch * o->xstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lx[0], 0, batch * o->lstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lc[0], 0, batch * o->lstride * 32, st));
  if (nlen) NLB_HIP(o, hipMemcpy2DAsync(o->x[0], o->xstride * 32, norm_x, nlen * 32, nlen * 32, batch, KIND, st));
  if (llen) {
    NLB_HIP(o, hipMemcpy2DAsync(o->lc[0], o->lstride * 32, lin_c, llen * 32, llen * 32, batch, KIND, st));
    NLB_HIP(o, hipMemcpy2DAsync(o->lx[0], o->lstride * 32, lin_x, llen * 32, llen * 32, batch, KIND, st));
  }
  if (comb) {
    const uint64_t cn = (uint64_t)batch * nlen, cl = (uint64_t)batch * llen;
    if (cn) k_nlb_fill_one<<<dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, st>>>(o->coefn, cn);
    if (cl) k_nlb_fill_one<<<dim3((unsigned)((cl + 255) / 256)), dim3(256), 0, st>>>(o->coefl, cl);
  } else
  {   // the shared starting basis: uploaded once (staged in the not-yet-used second buffer), then one copy per proof (they diverge after round 1)
    uint32_t *stg = o->P[1];
    NLB_HIP(o, hipMemsetAsync(stg, 0, o->cap * 64, st));
    if (nlen) NLB_HIP(o, hipMemcpyAsync(stg, norm_g_xy, nlen * 64, KIND, st));
    if (llen) NLB_HIP(o, hipMemcpyAsync(stg + evb(nlen) * 16, lin_h_xy, llen * 64, KIND, st));
    NLB_HIP(o, hipMemcpyAsync(stg + (evb(nlen) + evb(llen)) * 16, g_xy, 64, KIND, st));
    const size_t words4 = o->cap * 4;     // uint4 per basis
    k_nlb_broadcast<<<dim3((unsigned)((words4 + 255) / 256), (unsigned)std::min<size_t>(batch, 65535)), dim3(256), 0, st>>>(
        (const uint4 *)stg, (uint32_t)words4, (uint4 *)o->P[0], (uint32_t)batch);
  }
  o->q.resize(batch); o->qinv.resize(batch); o->nn.assign(batch, U256::one()); o->ln.assign(batch, U256::one());
  o->s.resize(batch); o->sX.resize(batch); o->sR.resize(batch);
  std::vector<uint64_t> hq, hs;
  if (on_device && comb) {                   // fixed-basis mode with resident inputs: the state is set up by a kernel, nothing comes to the host
    k_nlb_init_state<<<dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st>>>((const uint32_t *)q, (const uint32_t *)s, (uint32_t)batch, o->qs, o->stt);
    NLB_HIP(o, hipMemsetAsync(o->dK, 0, batch * sizeof(CollapseK), st));
    NLB_HIP(o, hipGetLastError());
    return BPPP_OK;
  }
  if (on_device) {                           // the host keeps q, q^-1 and s of every proof (round bookkeeping): bring them over
    hq.resize(4 * batch); hs.resize(4 * batch);
    NLB_HIP(o, hipMemcpyAsync(hq.data(), q, batch * 32, hipMemcpyDeviceToHost, st));
    NLB_HIP(o, hipMemcpyAsync(hs.data(), s, batch * 32, hipMemcpyDeviceToHost, st));
    NLB_HIP(o, hipStreamSynchronize(st));
    q = hq.data(); s = hs.data();
  }
  for (size_t b = 0; b < batch; b++) { o->q[b] = U256::load(q + 4 * b); o->qinv[b] = o->q[b]; o->s[b] = U256::load(s + 4 * b); }
  batch_minv(o->qinv.data(), batch, M);
  if (comb) {                                // the round state lives in HBM from here on
    std::vector<uint64_t> hqs(batch * 8), hst(batch * NLB_ST * 4, 0);
    for (size_t b = 0; b < batch; b++) {
      o->q[b].store(&hqs[8 * b]); o->qinv[b].store(&hqs[8 * b + 4]);
      hst[(b * NLB_ST + NLB_ST_NN) * 4] = 1; hst[(b * NLB_ST + NLB_ST_LN) * 4] = 1;
      o->s[b].store(&hst[(b * NLB_ST + NLB_ST_S) * 4]);
    }
    NLB_HIP(o, hipMemcpyAsync(o->qs, hqs.data(), batch * 64, hipMemcpyHostToDevice, st));
    NLB_HIP(o, hipMemcpyAsync(o->stt, hst.data(), batch * NLB_ST * 32, hipMemcpyHostToDevice, st));
    NLB_HIP(o, hipMemsetAsync(o->dK, 0, batch * sizeof(CollapseK), st));
  }
  NLB_HIP(o, hipStreamSynchronize(st));
  return BPPP_OK;
  };
  if (int rc = fill()) { bppp_nlb_destroy(o); return rc; }
  *out = o;
  return BPPP_OK;
}
}  // namespace bppp
namespace bppp {
bool nlb_fixed_basis(const bppp_nlb *o) { return o && o->comb != nullptr; }
// Fixed-basis mode, first half of a round for every proof, nothing leaves the device: d_XR [batch][2][16] receives X_b, R_b
// (canonical affine).  Asynchronous on the context's stream.
int nlb_round_commit_dev(bppp_nlb *o, uint32_t *d_XR) {
  if (!o || !o->comb || !d_XR) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  hipSetDevice(ctx->device);
  const size_t B = o->batch, ne = evb(o->n), le = evb(o->l), T = ne + le + 1;
  const int c = o->cur;
  hipStream_t st = ctx->stream;
  NLB_HIP(o, hipMemsetAsync(o->sc, 0, 2 * B * T * 32, st));
  // threads per proof: every thread pays ~15-20 multiplications for its powers of q^4 before its first pair, so a batch that fills the chip
  // anyway takes four pairs per thread (a lone proof keeps one pair per thread: its round is a dependency chain)
  unsigned round_threads = 64;
  const size_t pairs = std::max((o->n + 1) / 2, (o->l + 1) / 2), per_thread = B >= 256 ? 4 : 1;
  while (round_threads < 256 && (size_t)round_threads * per_thread < pairs) round_threads <<= 1;
  k_nlb_round<<<dim3((unsigned)B), dim3(round_threads), 0, st>>>(o->x[c], o->lc[c], o->lx[c], (uint32_t)o->n, (uint32_t)o->l, (uint32_t)o->xstride, (uint32_t)o->lstride,
                                                                 o->qs, (uint32_t)T, o->sc, o->sums);
  k_nlb_tails<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st>>>(o->sums, o->qs, o->stt, (uint32_t)B, (uint32_t)o->n, (uint32_t)o->l, (uint32_t)T, o->sc);
  const uint32_t Tc = (uint32_t)(1 + o->l0 + o->n0);
  if (o->rebase_level && !o->rebased && o->folds == o->rebase_level) {
    // the level basis of every proof, once: G'_p = sum_{i >> L == p} coef_i G_i (one table walk over the coefficients), g in slot 0; from here on the
    // coefficients restart at 1 over these points
    const uint32_t Tm = (uint32_t)(1 + o->l0r + o->n0r);
    k_nlb_coef_rows<<<dim3((Tc + 255) / 256, (unsigned)B), dim3(256), 0, st>>>(o->coefn, o->coefl, (uint32_t)o->n0, (uint32_t)o->l0, o->full);
    NLB_HIP(o, hipGetLastError());
    int rcg = comb_groups(o->comb, o->full, B, o->l0, o->n0, (int)o->rebase_level, o->pbasis, Tm, st);
    if (rcg) return fail(ctx, rcg, bppp_last_error(o->comb->ctx));
    k_nlb_set_g<<<dim3((unsigned)((B + 3) / 4)), dim3(64), 0, st>>>(o->comb->tab, (uint32_t)B, Tm, o->pbasis);      // tab[0][0][0] = 1 * P_0 = g
    const uint64_t cn = (uint64_t)B * o->n0r, cl = (uint64_t)B * o->l0r;
    if (cn) k_nlb_fill_one<<<dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, st>>>(o->coefn_r, cn);
    if (cl) k_nlb_fill_one<<<dim3((unsigned)((cl + 255) / 256)), dim3(256), 0, st>>>(o->coefl_r, cl);
    NLB_HIP(o, hipGetLastError());
    o->rebased = true; o->folds0 = o->folds;
  }
  if (o->rebased) {
    const uint32_t Tm = (uint32_t)(1 + o->l0r + o->n0r);
    k_nlb_expand<<<dim3((Tm + 255) / 256, (unsigned)(2 * B)), dim3(256), 0, st>>>(o->sc, (uint32_t)T, (uint32_t)ne, o->folds - o->folds0, o->coefn_r, o->coefl_r,
                                                                                  (uint32_t)o->n0r, (uint32_t)o->l0r, o->full_r);
    NLB_HIP(o, hipGetLastError());
    return msm_batch_dev(ctx, o->full_r, o->pbasis, Tm, 2 * B, 2, 0, d_XR);      // X and R of a proof share that proof's points
  }
  k_nlb_expand<<<dim3((Tc + 255) / 256, (unsigned)(2 * B)), dim3(256), 0, st>>>(o->sc, (uint32_t)T, (uint32_t)ne, o->folds, o->coefn, o->coefl, (uint32_t)o->n0,
                                                                                (uint32_t)o->l0, o->full);
  NLB_HIP(o, hipGetLastError());
  int rc = comb_msm(o->comb, o->full, 2 * B, d_XR, st, COMB_ROWS_PAIRS, 0, o->cscratch, comb_rows_scratch_bytes(2 * B));
  if (rc) return fail(ctx, rc, bppp_last_error(o->comb->ctx));
  return BPPP_OK;
}
// second half: d_es [batch][8] are the challenges, in HBM.  Asynchronous.
int nlb_round_collapse_dev(bppp_nlb *o, const uint32_t *d_es) {
  if (!o || !o->comb || !d_es) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  hipSetDevice(ctx->device);
  const size_t B = o->batch;
  const int c = o->cur, d = 1 - c;
  const size_t n2 = (o->n + 1) / 2, l2 = (o->l + 1) / 2;
  hipStream_t st = ctx->stream;
  k_nlb_collapse_state<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st>>>(d_es, o->qs, o->stt, (uint32_t)B, (uint32_t)o->n, (uint32_t)o->l, o->dK, o->dA);
  NLB_HIP(o, hipMemsetAsync(o->x[d], 0, B * o->xstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lx[d], 0, B * o->lstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lc[d], 0, B * o->lstride * 32, st));
  const size_t cn0 = o->rebased ? o->n0r : o->n0, cl0 = o->rebased ? o->l0r : o->l0;
  const uint32_t maxp = (uint32_t)std::max(n2, l2), maxc = (uint32_t)std::max(cn0, cl0);
  if (maxp) k_nlb_fold_scalars<<<dim3((maxp + 255) / 256, (unsigned)B), dim3(256), 0, st>>>(o->x[c], o->lc[c], o->lx[c], (uint32_t)o->n, (uint32_t)o->l,
                                                                                           (uint32_t)o->xstride, (uint32_t)o->lstride, o->dK, o->x[d], o->lc[d], o->lx[d]);
  k_nlb_coef_update<<<dim3((maxc + 255) / 256, (unsigned)B), dim3(256), 0, st>>>(o->rebased ? o->coefn_r : o->coefn, o->rebased ? o->coefl_r : o->coefl, (uint32_t)(o->n ? cn0 : 0),
                                                                                 (uint32_t)(o->l ? cl0 : 0), o->folds - (o->rebased ? o->folds0 : 0u), o->dA);
  NLB_HIP(o, hipGetLastError());
  o->n = o->n ? n2 : 0; o->l = o->l ? l2 : 0; o->cur = d; o->folds++;
  return BPPP_OK;
}
}  // namespace bppp
extern "C" {
int bppp_nlb_create(bppp_ctx *ctx, size_t batch, const uint64_t *s, const uint64_t g_xy[8], const uint64_t *q, const uint64_t *norm_x,
                    const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy, size_t llen,
                    bppp_nlb **out) {
  return nlb_create_impl(ctx, batch, s, g_xy, q, norm_x, norm_g_xy, nlen, lin_c, lin_x, lin_h_xy, llen, out, false, nullptr);
}

int bppp_nlb_lengths(const bppp_nlb *o, size_t *batch, size_t *nlen, size_t *llen) {
  if (!o || !batch || !nlen || !llen) return BPPP_ERR_ARG;
  *batch = o->batch; *nlen = o->n; *llen = o->l;
  return BPPP_OK;
}

// first half of proveRoundM for every proof: sX, X, sR, R are [batch][4] / [batch][8]
int bppp_nlb_round_commit(bppp_nlb *o, uint64_t *sX, uint64_t *X_xy, uint64_t *sR, uint64_t *R_xy) {
  if (!o || !sX || !X_xy || !sR || !R_xy) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = FR();
  const size_t B = o->batch, ne = evb(o->n), le = evb(o->l), T = ne + le + 1;
  const int c = o->cur;
  hipStream_t st = ctx->stream;
  if (o->comb) {                             // fixed-basis mode: the device routine, results brought to the host
    int rcd = nlb_round_commit_dev(o, o->d_out);
    if (rcd) return rcd;
    std::vector<uint64_t> outs(2 * B * 8), hst(B * NLB_ST * 4);
    NLB_HIP(o, hipMemcpyAsync(outs.data(), o->d_out, 2 * B * 64, hipMemcpyDeviceToHost, st));
    NLB_HIP(o, hipMemcpyAsync(hst.data(), o->stt, B * NLB_ST * 32, hipMemcpyDeviceToHost, st));
    NLB_HIP(o, hipStreamSynchronize(st));
    for (size_t b = 0; b < B; b++) {
      memcpy(X_xy + 8 * b, &outs[16 * b], 64); memcpy(R_xy + 8 * b, &outs[16 * b + 8], 64);
      memcpy(sX + 4 * b, &hst[(b * NLB_ST + NLB_ST_SX) * 4], 32); memcpy(sR + 4 * b, &hst[(b * NLB_ST + NLB_ST_SR) * 4], 32);
    }
    return BPPP_OK;
  }
  std::vector<uint64_t> h(B * 16);
  for (size_t b = 0; b < B; b++) { o->q[b].store(&h[8 * b]); o->qinv[b].store(&h[8 * b + 4]); }
  NLB_HIP(o, hipMemcpyAsync(o->qs, h.data(), B * 64, hipMemcpyHostToDevice, st));
  NLB_HIP(o, hipMemsetAsync(o->sc, 0, 2 * B * T * 32, st));
  unsigned round_threads = 64;
  while (round_threads < 256 && round_threads < std::max((o->n + 1) / 2, (o->l + 1) / 2)) round_threads <<= 1;
  k_nlb_round<<<dim3((unsigned)B), dim3(round_threads), 0, st>>>(o->x[c], o->lc[c], o->lx[c], (uint32_t)o->n, (uint32_t)o->l, (uint32_t)o->xstride, (uint32_t)o->lstride,
                                                      o->qs, (uint32_t)T, o->sc, o->sums);
  std::vector<uint64_t> sums(B * 16);
  NLB_HIP(o, hipMemcpyAsync(sums.data(), o->sums, B * 128, hipMemcpyDeviceToHost, st));
  NLB_HIP(o, hipStreamSynchronize(st));
  std::vector<uint64_t> tails(B * 8);
  parallel_ranges(B, [&](size_t lo, size_t hi) { for (size_t b = lo; b < hi; b++) {
    U256 q = o->q[b], q2 = mmul(q, q, M), q3 = mmul(q2, q, M), q4 = mmul(q2, q2, M), n2 = mmul(o->nn[b], o->nn[b], M);
    U256 sXn = mmul(mmul(madd(n2, n2, M), q3, M), U256::load(&sums[16 * b]), M);         // 2 n^2 q^3 sX'  (NormArgument.hs:113)
    U256 sRn = mmul(mmul(n2, q4, M), U256::load(&sums[16 * b + 4]), M);                   // n^2 q^4 sR'
    o->sX[b] = madd(o->n ? sXn : U256::zero(), o->l ? U256::load(&sums[16 * b + 8]) : U256::zero(), M);
    o->sR[b] = madd(o->n ? sRn : U256::zero(), o->l ? U256::load(&sums[16 * b + 12]) : U256::zero(), M);
    o->sX[b].store(sX + 4 * b); o->sR[b].store(sR + 4 * b);
    o->sX[b].store(&tails[8 * b]); o->sR[b].store(&tails[8 * b + 4]);
  } });
  // the scalar on g is the last term of each instance: instance 2b (X) and 2b+1 (R)
  NLB_HIP(o, hipMemcpy2DAsync(o->sc + (T - 1) * 8, T * 32, tails.data(), 32, 32, 2 * B, hipMemcpyHostToDevice, st));
  std::vector<uint64_t> outs(2 * B * 8);
  int rc = msm_run(ctx, o->sc, o->P[c], T, 2 * B, 2, 0, outs.data());   // X and R of a proof share that proof's basis
  if (rc) return rc;
  for (size_t b = 0; b < B; b++) { memcpy(X_xy + 8 * b, &outs[16 * b], 64); memcpy(R_xy + 8 * b, &outs[16 * b + 8], 64); }
  return BPPP_OK;
}

// second half of proveRoundM for every proof; es is [batch][4]
int bppp_nlb_round_collapse(bppp_nlb *o, const uint64_t *es) {
  if (!o || !es) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = FR();
  const size_t B = o->batch;
  const int c = o->cur, d = 1 - c;
  const size_t ne = evb(o->n), le = evb(o->l), n2 = (o->n + 1) / 2, l2 = (o->l + 1) / 2, ne2 = evb(n2), le2 = evb(l2);
  hipStream_t st = ctx->stream;
  std::vector<CollapseK> K(B);
  auto put8 = [](uint32_t *dst, const U256 &v) { for (int i = 0; i < 8; i++) dst[i] = (uint32_t)(v.w[i / 2] >> (32 * (i & 1))); };
  std::vector<U256> inv(2 * B, U256::zero()), a0l(B), b0n(B), b0l(B);
  // the advanced host state is built beside the current one and committed only after the device work has been issued and has
  // completed: an error return leaves the prover exactly where it was
  std::vector<U256> s_new(o->s), nn_new(o->nn), q_new(o->q), qinv_new(o->qinv), ln_new(o->ln);
  for (size_t b = 0; b < B; b++)
    if (cmp(U256::load(es + 4 * b), M.m) >= 0) return fail(ctx, BPPP_ERR_ARG, "nlb_round_collapse: challenge not canonical");
  if (o->comb) {                             // fixed-basis mode: the device routine on the uploaded challenges
    NLB_HIP(o, hipMemcpyAsync(o->d_es, es, B * 32, hipMemcpyHostToDevice, st));
    int rcd = nlb_round_collapse_dev(o, o->d_es);
    if (rcd) return rcd;
    NLB_HIP(o, hipStreamSynchronize(st));
    return BPPP_OK;
  }
  std::atomic<int> too_big{0};
  parallel_ranges(B, [&](size_t lo, size_t hi) {
   for (size_t b = lo; b < hi; b++) {
    const U256 e = U256::load(es + 4 * b);
    U256 e1 = msub(mmul(e, e, M), U256::one(), M);
    s_new[b] = madd(o->s[b], madd(mmul(e, o->sX[b], M), mmul(e1, o->sR[b], M), M), M);
    memset(&K[b], 0, sizeof(CollapseK));
    if (o->n) {
      auto ab = rational_reduce_scalar(mmul(e, o->qinv[b], M));
      if ((ab.first.m[2] >> 1) || (ab.second.m[2] >> 1) || ab.first.m[3] || ab.second.m[3]) too_big = 1;
      b0n[b] = extract_scalar(ab.second); inv[2 * b] = b0n[b];
      jsf_recode(ab.second.m, ab.first.m, K[b].dn); K[b].nbneg = ab.second.neg; K[b].naneg = ab.first.neg;
    }
    if (o->l) {
      auto ab = rational_reduce_scalar(e);
      if ((ab.first.m[2] >> 1) || (ab.second.m[2] >> 1) || ab.first.m[3] || ab.second.m[3]) too_big = 1;
      a0l[b] = extract_scalar(ab.first); b0l[b] = extract_scalar(ab.second); inv[2 * b + 1] = b0l[b];
      jsf_recode(ab.second.m, ab.first.m, K[b].dl); K[b].lbneg = ab.second.neg; K[b].laneg = ab.first.neg;
    }
   }
   batch_minv(inv.data() + 2 * lo, 2 * (hi - lo), M);              // every b0^-1 of the range with ONE field inversion
   for (size_t b = lo; b < hi; b++) {
    const U256 e = U256::load(es + 4 * b);
    if (o->n) {
      put8(K[b].nu, inv[2 * b]); put8(K[b].nv, mmul(mmul(e, o->q[b], M), inv[2 * b], M));
      nn_new[b] = mmul(mmul(o->nn[b], b0n[b], M), o->qinv[b], M);
      q_new[b] = mmul(o->q[b], o->q[b], M); qinv_new[b] = mmul(o->qinv[b], o->qinv[b], M);
    }
    if (o->l) {
      put8(K[b].cu, b0l[b]); put8(K[b].cv, a0l[b]); put8(K[b].lu, inv[2 * b + 1]); put8(K[b].lv, mmul(e, inv[2 * b + 1], M));
      ln_new[b] = mmul(o->ln[b], b0l[b], M);
    }
   }
  });
  if (too_big) return fail(ctx, BPPP_ERR_ARG, "nlb: reduced scalar exceeds 129 bits");
  NLB_HIP(o, hipMemcpyAsync(o->dK, K.data(), B * sizeof(CollapseK), hipMemcpyHostToDevice, st));
  NLB_HIP(o, hipMemsetAsync(o->P[d], 0, B * o->cap * 64, st));
  NLB_HIP(o, hipMemsetAsync(o->x[d], 0, B * o->xstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lx[d], 0, B * o->lstride * 32, st));
  NLB_HIP(o, hipMemsetAsync(o->lc[d], 0, B * o->lstride * 32, st));
  uint32_t maxp = (uint32_t)std::max(n2, l2);
  if (maxp) {
    k_nlb_fold_scalars<<<dim3((maxp + 255) / 256, (unsigned)B), dim3(256), 0, st>>>(o->x[c], o->lc[c], o->lx[c], (uint32_t)o->n, (uint32_t)o->l, (uint32_t)o->xstride,
                                                                                   (uint32_t)o->lstride, o->dK, o->x[d], o->lc[d], o->lx[d]);
    const uint64_t pairs = (uint64_t)B * (n2 + l2);
    k_nlb_fold_points<<<dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, st>>>(o->P[c], (uint32_t)o->n, (uint32_t)o->l, (uint32_t)(ne + le + 1),
                                                                                (uint32_t)(ne2 + le2 + 1), o->dK, (uint32_t)B, o->P[d]);
  }
  k_nlb_move_g<<<dim3((unsigned)B), dim3(64), 0, st>>>(o->P[c], (uint32_t)(ne + le + 1), (uint32_t)(ne2 + le2 + 1), (uint32_t)(ne + le), (uint32_t)(ne2 + le2), o->P[d]);
  NLB_HIP(o, hipGetLastError());
  NLB_HIP(o, hipStreamSynchronize(st));
  o->s.swap(s_new); o->nn.swap(nn_new); o->q.swap(q_new); o->qinv.swap(qinv_new); o->ln.swap(ln_new);
  o->n = o->n ? n2 : 0; o->l = o->l ? l2 : 0; o->cur = d;
  return BPPP_OK;
}

// getWitness of every proof: norm_w [batch][nlen], lin_w [batch][llen] (current lengths), s [batch][4]
int bppp_nlb_get_witness(bppp_nlb *o, uint64_t *norm_w, uint64_t *lin_w, uint64_t *s) {
  if (!o || (o->n && !norm_w) || (o->l && !lin_w)) return BPPP_ERR_ARG;
  bppp_ctx *ctx = o->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = FR();
  const int c = o->cur;
  hipStream_t st = ctx->stream;
  if (o->n) NLB_HIP(o, hipMemcpy2DAsync(norm_w, o->n * 32, o->x[c], o->xstride * 32, o->n * 32, o->batch, hipMemcpyDeviceToHost, st));
  if (o->l) NLB_HIP(o, hipMemcpy2DAsync(lin_w, o->l * 32, o->lx[c], o->lstride * 32, o->l * 32, o->batch, hipMemcpyDeviceToHost, st));
  std::vector<uint64_t> hst;
  if (o->comb) {                             // fixed-basis mode keeps n, l, s of every proof in HBM
    hst.resize(o->batch * NLB_ST * 4);
    NLB_HIP(o, hipMemcpyAsync(hst.data(), o->stt, o->batch * NLB_ST * 32, hipMemcpyDeviceToHost, st));
  }
  NLB_HIP(o, hipStreamSynchronize(st));
  if (o->comb)
    for (size_t b = 0; b < o->batch; b++) {
      o->nn[b] = U256::load(&hst[(b * NLB_ST + NLB_ST_NN) * 4]); o->ln[b] = U256::load(&hst[(b * NLB_ST + NLB_ST_LN) * 4]);
      o->s[b] = U256::load(&hst[(b * NLB_ST + NLB_ST_S) * 4]);
    }
  for (size_t b = 0; b < o->batch; b++) {
    for (size_t i = 0; i < o->n; i++) mmul(U256::load(norm_w + 4 * (b * o->n + i)), o->nn[b], M).store(norm_w + 4 * (b * o->n + i));
    for (size_t i = 0; i < o->l; i++) mmul(U256::load(lin_w + 4 * (b * o->l + i)), o->ln[b], M).store(lin_w + 4 * (b * o->l + i));
    if (s) o->s[b].store(s + 4 * b);
  }
  return BPPP_OK;
}

}  // extern "C"
