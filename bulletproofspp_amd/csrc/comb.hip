// comb.hip — fixed-base comb for THOUSANDS of MSMs over ONE short basis (the prover: range-proof commitments and every round of the
// lockstep argument, 2 x batch instances of ~775 terms each over the setup's [g | H | G]).
//
// The reference commits with `innerProduct` over the setup's fixed points (commitRPW, src/RangeProof/Internal.hs:45-50; the argument's
// round commitments, src/NormArgument.hs:100-128, over a basis that is a known linear image of the setup's).  With the basis fixed
// per setup and 288 GB of HBM, every multiple a signed c-bit digit can ask for is stored once:
//     tab[w][i][d - 1] = d * 2^(c w) * P_i          w < W = ceil(257 / c),   d = 1 .. 2^(c-1)
// (c = 16: 17 x 774 x 32768 entries of 64 B = 27.6 GB — the widest window whose table stays under the caller's budget is taken;
// c = 13: 4.1 GB), and an MSM is then nothing but one mixed addition per non-zero digit into ONE
// accumulator: no digit sort, no buckets, no bucket reduction, no window combine, no doubling — 17 additions per term against
// 29 + sort + reduction on the bucket route at its best window for this shape (csrc/msm.hip, c = 9).
//
// k_comb_msm: one wavefront per instance, lane l takes one term of every 64; the table entry of the NEXT digit is requested
// before the addition of the current one is issued, so the 64-B gathers (random over the table: HBM, not cache) hide under ~2.8 k
// VALU instructions each.  The 64 lane sums meet in a shuffle tree; lane 0 normalises.  VALU-bound like k_acc_points.
//
// k_comb_msm_rows (round 4): thousands of LONG rows of full-width scalars (the argument's rounds) over a table of tens of GB — lane = instance, the
// wavefront walks (term, window) in lockstep and its 64 gathers fall into ONE table row; k_comb_join_rows adds the partial sums of an instance.
// k_comb_msm_packed: thousands of rows of a few dozen terms (8 or 16 lanes per instance).  k_comb_lanes: one lane per three-term instance.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include "comb.hpp"
#include "ec.hip.h"

namespace bppp {

struct CombK { uint32_t k[9]; };       // sum_w 2^(c-1) 2^(c w): adding it turns signed digits into unsigned c-bit fields (as k_digits)

// bases[w][i] = 2^(c w) P_i: one lane per point walks the chain (c doublings and one normalisation per window)
__global__ void __launch_bounds__(64) k_comb_bases(const uint32_t *__restrict__ pts, uint32_t T, int c, int W, uint32_t *__restrict__ bases) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  aff P = aff_load(pts + (size_t)i * 16);
  aff_store(bases + (size_t)i * 16, P);
  for (int w = 1; w < W; w++) {
    xyzz a = xyzz_dbl_aff(P);
    for (int k = 1; k < c; k++) a = xyzz_dbl(a);
    P = xyzz_to_aff(a);
    aff_store(bases + ((size_t)w * T + i) * 16, P);
  }
}
// one lane per (w, i, chunk): the multiples chunk * CH + 1 .. chunk * CH + CH of bases[w][i], each normalised (canonical affine rows)
__global__ void __launch_bounds__(64) k_comb_multiples(const uint32_t *__restrict__ bases, uint32_t T, int W, uint32_t D, uint32_t CH, uint32_t *__restrict__ tab) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunks = D / CH;
  if (g >= (uint64_t)W * T * chunks) return;
  const uint32_t chunk = (uint32_t)(g % chunks);
  const uint64_t wi = g / chunks;                       // w * T + i
  const aff B = aff_load(bases + wi * 16);
  const uint32_t m0 = chunk * CH;
  xyzz acc = xyzz_inf();
  for (int b = 31 - __builtin_clz(m0 | 1u); b >= 0 && m0; b--) {      // m0 * B, double and add
    acc = xyzz_dbl(acc);
    if ((m0 >> b) & 1u) xyzz_madd(acc, B);
  }
  uint32_t *row = tab + (wi * D + m0) * 16;
  for (uint32_t d = 0; d < CH; d++) { xyzz_madd(acc, B); aff_store(row + (size_t)d * 16, xyzz_to_aff(acc)); }
}

struct CombRaw { uint4 a, b, c, d; };                  // one 64-B table entry as loaded
BPPP_DI aff comb_aff(const CombRaw &r, bool neg) {
  fe x, y;
  x.v[0] = r.a.x; x.v[1] = r.a.y; x.v[2] = r.a.z; x.v[3] = r.a.w; x.v[4] = r.b.x; x.v[5] = r.b.y; x.v[6] = r.b.z; x.v[7] = r.b.w;
  y.v[0] = r.c.x; y.v[1] = r.c.y; y.v[2] = r.c.z; y.v[3] = r.c.w; y.v[4] = r.d.x; y.v[5] = r.d.y; y.v[6] = r.d.z; y.v[7] = r.d.w;
  aff p; p.x = fq_from_fe(x); p.y = fq_from_fe(y);
  return aff_cneg(p, neg);
}

// heavy_first: the instances come as (heavy, light) pairs — the prover's X (every scalar non-zero) and R (half of them) — and the
// launch dispatches all heavy ones first, so the light ones fill the slots that free up instead of leaving a tail of heavy ones
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_comb_msm(const uint32_t *__restrict__ tab, uint32_t T, int c, int W, uint32_t D, CombK K,
                                                 const uint32_t *__restrict__ scalars, uint32_t nterms, uint32_t ninst, int heavy_first, uint32_t parts, uint32_t tparts, int wlen,
                                                 uint32_t *__restrict__ partial, uint32_t *__restrict__ out) {
  // parts > 1 (few instances): `parts` = tparts x (window ranges of wlen windows) wavefronts share one instance — wavefront (tp, wr) takes the
  // term groups tp, tp + tparts, ... and of each term only the digits of windows [wr wlen, wr wlen + wlen) — and leave their sums in
  // `partial` for k_comb_join; otherwise one wavefront per instance writes the result
  const uint32_t lane = threadIdx.x, half = ninst >> 1, blk = blockIdx.x / parts, part = blockIdx.x % parts, tpart = part % tparts;
  const int w0 = (int)(part / tparts) * wlen, w1 = min(W, w0 + wlen);
  const uint32_t inst = !heavy_first ? blk : blk < half ? 2 * blk : 2 * (blk - half) + 1;
  const uint32_t mask = (1u << c) - 1u;
  const uint32_t *sc = scalars + (size_t)inst * nterms * 8;       // the first nterms <= T registered points
  xyzz acc = xyzz_inf();
  CombRaw pend; pend.a = pend.b = pend.c = pend.d = make_uint4(0, 0, 0, 0);
  bool pend_ok = false, pend_neg = false;
  // Lane l takes one term of every group of 64, rotated by 21 per group, and walks ITS terms at its own pace: a lane whose term is
  // zero moves straight on to its next non-zero one instead of idling through the other lanes' 20 digit steps.  With vectors whose
  // zeros follow a power-of-two pattern in the index (the argument's R scalars vanish on every left half) every lane then has the
  // same share, and the wavefront of such an instance takes half the steps.
  uint32_t k0 = w0 < W ? 64u * tpart : nterms, k = tpart, sp[9];      // (an empty window range: nothing to do)
  int w = w1, wend = w1;                                       // w == wend: this lane needs its next term
  bool neg = false, live = true;
  const uint32_t *ti = tab;
  while (__any(live)) {
    if (live && w >= wend) {
      live = false;
      while (k0 < nterms) {
        const uint32_t i = k0 + ((lane + 21u * k) & 63u);
        k0 += 64u * tparts; k += tparts;
        if (i >= nterms) continue;
        const fe s = fe_load(sc + (size_t)i * 8);
        if (fe_is_zero(s)) continue;
        fe t, tmp;
        raw_sub(t, fr_modulus(), s);                           // n - s
        neg = raw_sub(tmp, t, s) != 0;                         // s > n - s: take n - s and the negated point (reduceScalar, Commitment.hs:276-279)
        // a SHORT scalar (range-proof digits, bits, multiplicities) has no digit beyond window ceil(bits / c) (that one only as a carry): its lane
        // moves on after those instead of stepping through all W windows
        uint32_t hw = 1u; int top = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t v = neg ? t.v[q] : s.v[q]; if (v) { hw = v; top = q; } }
        const int nw = (32 * top + (32 - __builtin_clz(hw)) + c - 1) / c + 1;
        if (nw <= w0) continue;                                // nothing of this term in this wavefront's window range
        uint64_t cy = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { cy += (uint64_t)(neg ? t.v[q] : s.v[q]) + K.k[q]; sp[q] = (uint32_t)cy; cy >>= 32; }
        sp[8] = (uint32_t)cy + K.k[8];
        for (int j = 0; j < w0; j++) {                          // skip the windows of the other ranges
#pragma unroll
          for (int q = 0; q < 8; q++) sp[q] = (sp[q] >> c) | (sp[q + 1] << (32 - c));
          sp[8] >>= c;
        }
        ti = tab + (size_t)i * D * 16;
        wend = min(w1, nw);
        w = w0; live = true;
        break;
      }
    }
    CombRaw nxt; nxt.a = nxt.b = nxt.c = nxt.d = make_uint4(0, 0, 0, 0);
    bool ok = false, nneg = false;
    if (live) {
      const int d = (int)(sp[0] & mask) - (int)D;              // signed digit in [-D, D - 1]
#pragma unroll
      for (int q = 0; q < 8; q++) sp[q] = (sp[q] >> c) | (sp[q + 1] << (32 - c));
      sp[8] >>= c;
      ok = d != 0; nneg = (d < 0) != neg;
      if (ok) {
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint4 *e = (const uint4 *)(ti + ((size_t)w * T * D + (mag - 1)) * 16);
        nxt.a = e[0]; nxt.b = e[1]; nxt.c = e[2]; nxt.d = e[3];
      }
      w++;
    }
    if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));     // the previous digit's entry: its load was issued one step ago
    pend = nxt; pend_ok = ok; pend_neg = nneg;
  }
  if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
  for (int dd = 32; dd >= 1; dd >>= 1) {
    xyzz o = xyzz_shfl_down(acc, dd);
    if ((int)lane + dd < 64) xyzz_add(acc, o);
  }
  if (lane == 0) {
    if (parts > 1) xyzz_store(partial + ((size_t)inst * parts + part) * XYZZ_WORDS, acc);
    else aff_store(out + (size_t)inst * 16, xyzz_to_aff(acc));
  }
}
// THOUSANDS of instances of a FEW DOZEN terms (the inner-product prover's rows at the examples/64bit shape: 1 + 6 + 16 = 23 terms): with one wavefront per
// instance 23 of 64 lanes work and the epilogue — a 6-level shuffle tree and one inversion — costs as much as the 17 additions of the walk.  Here LPI lanes
// serve an instance (64 / LPI instances per wavefront), each lane walks ceil(nterms / LPI) terms, a log2(LPI)-level segmented tree joins them and the
// 64 / LPI inversions of a wavefront run side by side.
template <int LPI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_comb_msm_packed(const uint32_t *__restrict__ tab, uint32_t T, int c, int W, uint32_t D, CombK K,
                                                                                                  const uint32_t *__restrict__ scalars, uint32_t nterms, uint32_t ninst,
                                                                                                  uint32_t *__restrict__ out) {
  const uint32_t sub = threadIdx.x % LPI, inst = blockIdx.x * (64 / LPI) + threadIdx.x / LPI;
  const bool active = inst < ninst;
  const uint32_t mask = (1u << c) - 1u;
  const uint32_t *sc = scalars + (size_t)(active ? inst : 0) * nterms * 8;
  xyzz acc = xyzz_inf();
  CombRaw pend; pend.a = pend.b = pend.c = pend.d = make_uint4(0, 0, 0, 0);
  bool pend_ok = false, pend_neg = false;
  for (uint32_t i = sub; i < nterms; i += LPI) {
    fe s = fe_load(sc + (size_t)i * 8);
    if (!active) s = fe_zero();
    const bool nz = !fe_is_zero(s);
    fe t, tmp;
    raw_sub(t, fr_modulus(), s);
    const bool neg = raw_sub(tmp, t, s) != 0;                    // reduceScalar (Commitment.hs:276-279)
    uint32_t sp[9];
    uint64_t cy = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { cy += (uint64_t)(neg ? t.v[q] : s.v[q]) + K.k[q]; sp[q] = (uint32_t)cy; cy >>= 32; }
    sp[8] = (uint32_t)cy + K.k[8];
    const uint32_t *ti = tab + (size_t)i * D * 16;
#pragma unroll 1
    for (int w = 0; w < W; w++) {
      const int d = (int)(sp[0] & mask) - (int)D;
#pragma unroll
      for (int q = 0; q < 8; q++) sp[q] = (sp[q] >> c) | (sp[q + 1] << (32 - c));
      sp[8] >>= c;
      CombRaw nxt; nxt.a = nxt.b = nxt.c = nxt.d = make_uint4(0, 0, 0, 0);
      const bool ok = nz && d != 0, nneg = (d < 0) != neg;
      if (ok) {
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint4 *e = (const uint4 *)(ti + ((size_t)w * T * D + (mag - 1)) * 16);
        nxt.a = e[0]; nxt.b = e[1]; nxt.c = e[2]; nxt.d = e[3];
      }
      if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
      pend = nxt; pend_ok = ok; pend_neg = nneg;
    }
  }
  if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
#pragma unroll
  for (int dd = LPI / 2; dd >= 1; dd >>= 1) {
    xyzz o = xyzz_shfl_down(acc, dd);
    if ((int)sub + dd < LPI) xyzz_add(acc, o);
  }
  if (sub == 0 && active) aff_store(out + (size_t)inst * 16, xyzz_to_aff(acc));
}

// LONG rows of FULL-WIDTH scalars, thousands of them (the lockstep argument's round commitments: 2 x batch rows over the whole basis): lane = INSTANCE.
// The 64 lanes of a wavefront then walk the SAME term and window at the same time, so their 64 gathers fall into ONE table row (D entries of 64 B,
// 256 KB at c = 13) instead of 64 rows a gigabyte apart: benchmarks/gather_locality.hip measures 50 G gathers/s for that pattern at any table size against
// 19-20 G/s for one row per lane once the table is beyond 8 GB (address translation) — and k_comb_msm needs 14-15 G/s of them.
// A wavefront takes `clen` consecutive terms of 64 instances and parks the 64 partial sums for k_comb_join_rows; with `pairs` the even (heavy) instances
// are dispatched before the odd (light) ones, whose scalars vanish on a pattern that is the same for every instance (a wave-uniform skip).
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_comb_msm_rows(const uint32_t *__restrict__ tab, uint32_t T, int c, int W, uint32_t D, CombK K,
                                                 const uint32_t *__restrict__ scalars, uint32_t nterms, uint32_t ninst, int pairs, uint32_t ngroups, uint32_t chunks, uint32_t clen,
                                                 uint32_t *__restrict__ partial) {
  const uint32_t lane = threadIdx.x, mask = (1u << c) - 1u;
  uint32_t inst, chunk;
  if (pairs) {                                                  // ngroups = groups of 64 PAIRS; blocks [0, chunks x ngroups) are the even instances
    const uint32_t par = blockIdx.x / (chunks * ngroups), rem = blockIdx.x % (chunks * ngroups);
    chunk = rem / ngroups; inst = 2u * ((rem % ngroups) * 64u + lane) + par;
  } else { chunk = blockIdx.x / ngroups; inst = (blockIdx.x % ngroups) * 64u + lane; }
  const bool active = inst < ninst;
  const uint32_t *sc = scalars + (size_t)(active ? inst : 0) * nterms * 8;
  const uint32_t i0 = chunk * clen, i1 = min(nterms, i0 + clen);
  xyzz acc = xyzz_inf();
  CombRaw pend; pend.a = pend.b = pend.c = pend.d = make_uint4(0, 0, 0, 0);
  bool pend_ok = false, pend_neg = false;
  for (uint32_t i = i0; i < i1; i++) {
    fe s = fe_load(sc + (size_t)i * 8);
    if (!active) s = fe_zero();
    const bool nz = !fe_is_zero(s);
    if (!__any(nz)) continue;
    fe t, tmp;
    raw_sub(t, fr_modulus(), s);
    const bool neg = raw_sub(tmp, t, s) != 0;                    // reduceScalar (Commitment.hs:276-279)
    uint32_t sp[9];
    uint64_t cy = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { cy += (uint64_t)(neg ? t.v[q] : s.v[q]) + K.k[q]; sp[q] = (uint32_t)cy; cy >>= 32; }
    sp[8] = (uint32_t)cy + K.k[8];
    const uint32_t *ti = tab + (size_t)i * D * 16;
#pragma unroll 1
    for (int w = 0; w < W; w++) {
      const int d = (int)(sp[0] & mask) - (int)D;
#pragma unroll
      for (int q = 0; q < 8; q++) sp[q] = (sp[q] >> c) | (sp[q + 1] << (32 - c));
      sp[8] >>= c;
      CombRaw nxt; nxt.a = nxt.b = nxt.c = nxt.d = make_uint4(0, 0, 0, 0);
      const bool ok = nz && d != 0, nneg = (d < 0) != neg;
      if (ok) {
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint4 *e = (const uint4 *)(ti + ((size_t)w * T * D + (mag - 1)) * 16);
        nxt.a = e[0]; nxt.b = e[1]; nxt.c = e[2]; nxt.d = e[3];
      }
      if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
      pend = nxt; pend_ok = ok; pend_neg = nneg;
    }
  }
  if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
  if (active) xyzz_store(partial + ((size_t)inst * chunks + chunk) * XYZZ_WORDS, acc);
}
// the `parts` partial sums of an instance (any count): LPI lanes serve an instance (64 / LPI instances per wavefront) — lane s adds the partials s, s + LPI, ...,
// a log2(LPI)-level segmented shuffle tree joins the lanes, lane 0 of the segment normalises.  (With 64 lanes per instance whatever the count, the 6-level
// tree — six full-wavefront additions for 63 useful ones — made the join of 8192 x 86 partials 14 % of the round's instructions.)
template <int LPI>
__global__ void __launch_bounds__(64) k_comb_join_rows(const uint32_t *__restrict__ partial, uint32_t parts, uint32_t ninst, uint32_t *__restrict__ out) {
  const uint32_t sub = threadIdx.x % LPI, inst = blockIdx.x * (64 / LPI) + threadIdx.x / LPI;
  const bool active = inst < ninst;
  xyzz acc = xyzz_inf();
  if (active)
    for (uint32_t p = sub; p < parts; p += LPI) {
      xyzz o = xyzz_load(partial + ((size_t)inst * parts + p) * XYZZ_WORDS);
      xyzz_add(acc, o);
    }
#pragma unroll
  for (int dd = LPI / 2; dd >= 1; dd >>= 1) {
    xyzz o = xyzz_shfl_down(acc, dd);
    if ((int)sub + dd < LPI) xyzz_add(acc, o);
  }
  if (sub == 0 && active) aff_store(out + (size_t)inst * 16, xyzz_to_aff(acc));
}

// The level-L basis of every proof, materialised (the lockstep argument's late rounds, csrc/nlb.hip): scalars [ninst][1 + l0 + n0] are the fold
// coefficients of a proof over [g | lin | norm] (g's slot unused); group q of 2^L consecutive points of the lin part, then of the norm part, is summed with
// its coefficients into out[inst][1 + q] (canonical affine; out[inst][0] is not written).  Lane = instance as in k_comb_msm_rows: one wavefront per
// (64 instances, group), nothing to join.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_comb_msm_groups(const uint32_t *__restrict__ tab, uint32_t T, int c, int W, uint32_t D, CombK K,
                                                 const uint32_t *__restrict__ scalars, uint32_t nterms, uint32_t ninst, uint32_t ngroups, uint32_t l0, uint32_t n0, int L, uint32_t l0r,
                                                 uint32_t out_stride, uint32_t *__restrict__ out) {
  const uint32_t lane = threadIdx.x, mask = (1u << c) - 1u;
  const uint32_t q = blockIdx.x / ngroups, inst = (blockIdx.x % ngroups) * 64u + lane;
  const bool active = inst < ninst;
  const uint32_t *sc = scalars + (size_t)(active ? inst : 0) * nterms * 8;
  uint32_t i0, i1;
  if (q < l0r) { i0 = 1u + (q << L); i1 = min(1u + l0, i0 + (1u << L)); }
  else { i0 = 1u + l0 + ((q - l0r) << L); i1 = min(1u + l0 + n0, i0 + (1u << L)); }
  xyzz acc = xyzz_inf();
  CombRaw pend; pend.a = pend.b = pend.c = pend.d = make_uint4(0, 0, 0, 0);
  bool pend_ok = false, pend_neg = false;
  for (uint32_t i = i0; i < i1; i++) {
    fe s = fe_load(sc + (size_t)i * 8);
    if (!active) s = fe_zero();
    const bool nz = !fe_is_zero(s);
    if (!__any(nz)) continue;
    fe t, tmp;
    raw_sub(t, fr_modulus(), s);
    const bool neg = raw_sub(tmp, t, s) != 0;                    // reduceScalar (Commitment.hs:276-279)
    uint32_t sp[9];
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { cy += (uint64_t)(neg ? t.v[k] : s.v[k]) + K.k[k]; sp[k] = (uint32_t)cy; cy >>= 32; }
    sp[8] = (uint32_t)cy + K.k[8];
    const uint32_t *ti = tab + (size_t)i * D * 16;
#pragma unroll 1
    for (int w = 0; w < W; w++) {
      const int d = (int)(sp[0] & mask) - (int)D;
#pragma unroll
      for (int k = 0; k < 8; k++) sp[k] = (sp[k] >> c) | (sp[k + 1] << (32 - c));
      sp[8] >>= c;
      CombRaw nxt; nxt.a = nxt.b = nxt.c = nxt.d = make_uint4(0, 0, 0, 0);
      const bool ok = nz && d != 0, nneg = (d < 0) != neg;
      if (ok) {
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint4 *e = (const uint4 *)(ti + ((size_t)w * T * D + (mag - 1)) * 16);
        nxt.a = e[0]; nxt.b = e[1]; nxt.c = e[2]; nxt.d = e[3];
      }
      if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
      pend = nxt; pend_ok = ok; pend_neg = nneg;
    }
  }
  if (pend_ok) xyzz_madd(acc, comb_aff(pend, pend_neg));
  if (active) aff_store(out + ((size_t)inst * out_stride + 1u + q) * 16, xyzz_to_aff(acc));
}

// MANY instances of a FEW terms each (the prover's input commitments v g + ty H0 + bl H1: batch x #values instances over the first
// three registered points): one LANE per instance walks its terms and digits; zero scalars and zero digits cost nothing
__global__ void __launch_bounds__(64) k_comb_lanes(const uint32_t *__restrict__ tab, uint32_t T, int c, int W, uint32_t D, CombK K,
                                                   const uint32_t *__restrict__ scalars, uint32_t nterms, uint64_t ninst, uint32_t *__restrict__ out) {
  const uint64_t inst = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (inst >= ninst) return;
  const uint32_t mask = (1u << c) - 1u;
  xyzz acc = xyzz_inf();
  for (uint32_t i = 0; i < nterms; i++) {
    const fe s = fe_load(scalars + (inst * nterms + i) * 8);
    if (fe_is_zero(s)) continue;
    fe t, tmp;
    raw_sub(t, fr_modulus(), s);
    const bool neg = raw_sub(tmp, t, s) != 0;
    uint32_t sp[9];
    uint64_t cy = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { cy += (uint64_t)(neg ? t.v[q] : s.v[q]) + K.k[q]; sp[q] = (uint32_t)cy; cy >>= 32; }
    sp[8] = (uint32_t)cy + K.k[8];
    const uint32_t *ti = tab + (size_t)i * D * 16;
#pragma unroll 1
    for (int w = 0; w < W; w++) {
      const int d = (int)(sp[0] & mask) - (int)D;
#pragma unroll
      for (int q = 0; q < 8; q++) sp[q] = (sp[q] >> c) | (sp[q + 1] << (32 - c));
      sp[8] >>= c;
      if (d) {
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint4 *e = (const uint4 *)(ti + ((size_t)w * T * D + (mag - 1)) * 16);
        CombRaw r; r.a = e[0]; r.b = e[1]; r.c = e[2]; r.d = e[3];
        xyzz_madd(acc, comb_aff(r, (d < 0) != neg));
      }
    }
  }
  aff_store(out + inst * 16, xyzz_to_aff(acc));
}

// the partial sums of an instance (parts <= 64): one wavefront adds them and normalises
__global__ void __launch_bounds__(64) k_comb_join(const uint32_t *__restrict__ partial, uint32_t parts, uint32_t *__restrict__ out) {
  const uint32_t inst = blockIdx.x, lane = threadIdx.x;
  xyzz acc = lane < parts ? xyzz_load(partial + ((size_t)inst * parts + lane) * XYZZ_WORDS) : xyzz_inf();
  int top = 1; while (top < (int)parts) top <<= 1;               // parts <= 64
  for (int dd = top >> 1; dd >= 1; dd >>= 1) {
    xyzz o = xyzz_shfl_down(acc, dd);
    if ((int)lane + dd < 64) xyzz_add(acc, o);
  }
  if (lane == 0) aff_store(out + (size_t)inst * 16, xyzz_to_aff(acc));
}

void comb_destroy(CombTable *t) {
  if (!t) return;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  if (t->tab) hipFree(t->tab);
  ctx_release(t->ctx);
  delete t;
}

int comb_create(bppp_ctx *ctx, const uint32_t *d_points, size_t T, int window_bits, size_t budget_bytes, CombTable **out) {
  if (!ctx || !d_points || !T || !out || T >= (1u << 24)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  int c = window_bits;
  auto size_of = [&](int cc) { return (size_t)((257 + cc - 1) / cc) * T * ((size_t)1 << (cc - 1)) * 64; };
  if (!c) { c = 18; while (c > 4 && size_of(c) > budget_bytes) c--; }
  if (c < 4 || c > 18) return fail(ctx, BPPP_ERR_ARG, "comb_create: window_bits must be in [4,18]");
  if (!window_bits && size_of(c) > budget_bytes) return fail(ctx, BPPP_ERR_ARG, "comb_create: no window of 4 bits or more fits the budget (" + std::to_string(size_of(c) >> 20) + " MiB needed)");
  CombTable *t = new CombTable();
  t->ctx = ctx; ctx_retain(ctx); t->T = T; t->c = c; t->W = (257 + c - 1) / c; t->D = 1 << (c - 1); t->tab = nullptr; t->bytes = size_of(c);
  uint32_t *bases = nullptr;
  hipStream_t st = ctx->stream;
  // a failed hipMalloc (or launch) leaves its error in the runtime's last-error slot, which later successful calls do NOT clear on
  // ROCm 7: consume it here, or the next launch check of this context reports a stale out-of-memory
  auto bail = [&](const std::string &m) { (void)hipGetLastError(); if (bases) hipFree(bases); comb_destroy(t); (void)hipGetLastError(); return fail(ctx, BPPP_ERR_HIP, m); };
  if (hipMalloc(&t->tab, t->bytes) != hipSuccess) return bail("comb_create: hipMalloc of the table failed (" + std::to_string(t->bytes >> 20) + " MiB)");
  if (hipMalloc(&bases, (size_t)t->W * T * 64) != hipSuccess) return bail("comb_create: hipMalloc failed");
  k_comb_bases<<<dim3((unsigned)((T + 63) / 64)), dim3(64), 0, st>>>(d_points, (uint32_t)T, c, t->W, bases);
  const uint32_t CH = std::min<uint32_t>(256u, (uint32_t)t->D);
  const uint64_t lanes = (uint64_t)t->W * T * (t->D / CH);
  k_comb_multiples<<<dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st>>>(bases, (uint32_t)T, t->W, (uint32_t)t->D, CH, t->tab);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return bail("comb_create: table kernels failed");
  hipFree(bases);
  *out = t;
  return BPPP_OK;
}

int comb_lanes(const CombTable *t, const uint32_t *d_scalars, size_t nterms, size_t ninst, uint32_t *d_out_aff, hipStream_t st) {
  if (!t || !d_scalars || !d_out_aff || !nterms || nterms > t->T) return BPPP_ERR_ARG;
  if (!ninst) return BPPP_OK;
  CombK K; memset(&K, 0, sizeof K);
  for (int w = 0; w < t->W; w++) { const int bit = w * t->c + t->c - 1; if (bit < 288) K.k[bit >> 5] |= 1u << (bit & 31); }
  k_comb_lanes<<<dim3((unsigned)((ninst + 63) / 64)), dim3(64), 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)nterms, (uint64_t)ninst, d_out_aff);
  if (hipGetLastError() != hipSuccess) return fail(t->ctx, BPPP_ERR_HIP, "comb_lanes: launch failed");
  return BPPP_OK;
}

int comb_groups(const CombTable *t, const uint32_t *d_scalars, size_t ninst, size_t l0, size_t n0, int L, uint32_t *d_out_aff, size_t out_stride, hipStream_t st) {
  if (!t || !d_scalars || !d_out_aff || !ninst || 1 + l0 + n0 != t->T || L < 1 || L > 20) return BPPP_ERR_ARG;
  const size_t l0r = (l0 + ((size_t)1 << L) - 1) >> L, n0r = (n0 + ((size_t)1 << L) - 1) >> L, ngroups = (ninst + 63) / 64;
  if (out_stride < 1 + l0r + n0r || ngroups * (l0r + n0r) >= (1ull << 31)) return BPPP_ERR_ARG;
  CombK K; memset(&K, 0, sizeof K);
  for (int w = 0; w < t->W; w++) { const int bit = w * t->c + t->c - 1; if (bit < 288) K.k[bit >> 5] |= 1u << (bit & 31); }
  k_comb_msm_groups<<<dim3((unsigned)(ngroups * (l0r + n0r))), dim3(64), 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)t->T, (uint32_t)ninst,
                                                                               (uint32_t)ngroups, (uint32_t)l0, (uint32_t)n0, L, (uint32_t)l0r, (uint32_t)out_stride, d_out_aff);
  if (hipGetLastError() != hipSuccess) return fail(t->ctx, BPPP_ERR_HIP, "comb_groups: launch failed");
  return BPPP_OK;
}

int comb_msm(const CombTable *t, const uint32_t *d_scalars, size_t ninst, uint32_t *d_out_aff, hipStream_t st, int rows_hint, size_t nterms, uint32_t *d_scratch,
             size_t scratch_bytes) {
  if (!t || !d_scalars || !d_out_aff || ninst >= (1u << 31) || nterms > t->T) return BPPP_ERR_ARG;
  if (!nterms) nterms = t->T;
  if (!ninst) return BPPP_OK;
  CombK K; memset(&K, 0, sizeof K);
  for (int w = 0; w < t->W; w++) { const int bit = w * t->c + t->c - 1; if (bit < 288) K.k[bit >> 5] |= 1u << (bit & 31); }
  // a few dozen terms per instance, thousands of instances: several instances per wavefront (k_comb_msm_packed)
  if (nterms <= 48 && ninst >= 512 && !t->ctx->tune.comb_no_packed) {
    const unsigned lpi = nterms <= 24 ? 8 : 16;
    const unsigned grid_p = (unsigned)((ninst * lpi + 63) / 64);
    if (lpi == 8) k_comb_msm_packed<8><<<dim3(grid_p), dim3(64), 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)nterms, (uint32_t)ninst, d_out_aff);
    else k_comb_msm_packed<16><<<dim3(grid_p), dim3(64), 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)nterms, (uint32_t)ninst, d_out_aff);
    if (hipGetLastError() != hipSuccess) return fail(t->ctx, BPPP_ERR_HIP, "comb_msm: launch failed");
    return BPPP_OK;
  }
  // long rows of full-width scalars by the thousand (the argument's rounds, the blinded phase rows: rows_hint) over a table beyond the reach of the address translation
  // caches: lane = instance (k_comb_msm_rows), ~16384 wavefronts of `clen` terms each
  if (rows_hint != COMB_ROWS_ANY && d_scratch && ninst >= 512 && nterms >= 256 && t->bytes >= t->ctx->tune.comb_rows_min_bytes) {
    const bool pairs = rows_hint == COMB_ROWS_PAIRS && !(ninst & 1);
    const uint32_t ngroups = (uint32_t)(((pairs ? ninst / 2 : ninst) + 63) / 64), gtot = pairs ? 2 * ngroups : ngroups;
    uint32_t chunks = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)(nterms / 4), (std::min<uint32_t>(t->ctx->tune.comb_rows_waves ? (uint32_t)t->ctx->tune.comb_rows_waves : (uint32_t)COMB_ROWS_WAVES, (uint32_t)COMB_ROWS_WAVES) + gtot - 1) / gtot));
    while (chunks > 1 && (size_t)ninst * chunks * XYZZ_WORDS * 4 > scratch_bytes) chunks--;
    const uint32_t clen = (uint32_t)((nterms + chunks - 1) / chunks);
    chunks = (uint32_t)((nterms + clen - 1) / clen);
    if ((size_t)ninst * chunks * XYZZ_WORDS * 4 <= scratch_bytes) {
      const dim3 grid_r(gtot * chunks), block_r(64);
#define COMB_ROWS_LAUNCH(V) k_comb_msm_rows<V><<<grid_r, block_r, 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)nterms, (uint32_t)ninst, pairs ? 1 : 0, ngroups, chunks, clen, d_scratch)
      if (t->ctx->tune.comb_wpe == 3) COMB_ROWS_LAUNCH(3); else COMB_ROWS_LAUNCH(2);
#undef COMB_ROWS_LAUNCH
      unsigned lpi = 4;                                          // about 12 partials per lane (6: 0.35 ms, 12: 0.29, 24: 0.28 for 8192 x 86 partials)
      while (lpi < 64 && lpi * 12 < chunks) lpi <<= 1;
      const dim3 grid_j((unsigned)((ninst * lpi + 63) / 64));
      switch (lpi) {
        case 4: k_comb_join_rows<4><<<grid_j, block_r, 0, st>>>(d_scratch, chunks, (uint32_t)ninst, d_out_aff); break;
        case 8: k_comb_join_rows<8><<<grid_j, block_r, 0, st>>>(d_scratch, chunks, (uint32_t)ninst, d_out_aff); break;
        case 16: k_comb_join_rows<16><<<grid_j, block_r, 0, st>>>(d_scratch, chunks, (uint32_t)ninst, d_out_aff); break;
        case 32: k_comb_join_rows<32><<<grid_j, block_r, 0, st>>>(d_scratch, chunks, (uint32_t)ninst, d_out_aff); break;
        default: k_comb_join_rows<64><<<grid_j, block_r, 0, st>>>(d_scratch, chunks, (uint32_t)ninst, d_out_aff); break;
      }
      if (hipGetLastError() != hipSuccess) return fail(t->ctx, BPPP_ERR_HIP, "comb_msm: launch failed");
      return BPPP_OK;
    }
  }
  // wavefronts per SIMD the register allocation aims at: 2 (225 VGPRs) measured 2 % ahead of 3 (168); 4 (128) spills and is 2.4 x slower
  const int wpe = t->ctx->tune.comb_wpe ? t->ctx->tune.comb_wpe : 2;
  // few instances: several wavefronts per instance (up to one per group of 64 terms), so that a launch is ~1024 wavefronts wide and
  // its depth is a few additions instead of nterms / 64 x W; needs the caller's scratch for the partial sums
  // (a handful of instances — one proof's X and R — also split each term's W digits over window ranges: 17 + 6 chained additions
  // become 5 + 6, the join one level deeper)
  uint32_t tparts = 1, wsplit = 1;
  const uint32_t groups = (uint32_t)((nterms + 63) / 64);
  // (round 4: also a few thousand LONG instances — the binary prover's 2 x 1024 rows of 4099 terms are exactly one wavefront per slot of the
  // chip, the X rows twice as long as the R rows: the launch then lasts as long as an X row on a half-idle SIMD.  Aim at ~8192 wavefronts.)
  // (the wide target only for long instances, >= 32 groups of 64 terms: a 774-term row walks 13 terms per lane, and splitting the
  // 4096 rows of a half-batch in two costs a join launch for nothing; the round 1-3 rule — about 1024 wavefronts — serves the rest)
  // (measured, norm-linear prover: 4096 / 2048 / 1024 proofs 90.3 / 54.5 / 36.3 ms with the wide target for every shape, 84.6 / 49.4 / 32.9 ms with it for
  // long instances only)
  const size_t target = groups >= 32 ? COMB_SPLIT_BELOW : 1024;
  if (d_scratch && ninst < target && groups > 1) {
    tparts = std::min<uint32_t>(std::min<uint32_t>(groups, 64u), (uint32_t)((target + ninst - 1) / ninst));
    while (tparts > 1 && (size_t)ninst * tparts * XYZZ_WORDS * 4 > scratch_bytes) tparts--;
    if (tparts == groups && !t->ctx->tune.comb_no_wsplit) {
      wsplit = std::min<uint32_t>(std::min<uint32_t>(64u / tparts, 4u), (uint32_t)(1024 / std::max<size_t>(1, ninst * tparts)));
      while (wsplit > 1 && (size_t)ninst * tparts * wsplit * XYZZ_WORDS * 4 > scratch_bytes) wsplit--;
      if (wsplit < 1) wsplit = 1;
    }
  }
  const uint32_t parts = tparts * wsplit;
  const int wlen = (t->W + (int)wsplit - 1) / (int)wsplit;
  const dim3 grid((unsigned)(ninst * parts)), block(64);
  const int hf = (rows_hint == COMB_ROWS_PAIRS && !(ninst & 1)) ? 1 : 0;
#define COMB_LAUNCH(V) k_comb_msm<V><<<grid, block, 0, st>>>(t->tab, (uint32_t)t->T, t->c, t->W, (uint32_t)t->D, K, d_scalars, (uint32_t)nterms, (uint32_t)ninst, hf, parts, tparts, wlen, d_scratch, d_out_aff)
  if (wpe <= 2) COMB_LAUNCH(2); else COMB_LAUNCH(3);
#undef COMB_LAUNCH
  if (parts > 1) k_comb_join<<<dim3((unsigned)ninst), dim3(64), 0, st>>>(d_scratch, parts, d_out_aff);
  if (hipGetLastError() != hipSuccess) return fail(t->ctx, BPPP_ERR_HIP, "comb_msm: launch failed");
  return BPPP_OK;
}

}  // namespace bppp
