// ipb.hip — B inner-product arguments (src/Bulletproof/InnerProductArgument.hs) proved in LOCKSTEP with every vector, every fold coefficient
// and the round state resident in HBM: a round is a stream of kernels, nothing returns to the host before the last one.
//
// proveBPM (src/Bulletproof.hs:357-359) of the inner-product flavour WITHOUT one basis change and WITHOUT one point fold (the two
// observations of csrc/rpprove.hip's ip_argument_lockstep, whose host-core version stays there as the cross-check, BPPP_RP_HOST_ALGEBRA):
//   (1) makeNorm's basis (:194-206) g'_j = g_2j+1 + r g_2j, h'_j = g_2j+1 - r g_2j enters every commitment linearly:
//       A g'_j + B h'_j = (A + B) g_2j+1 + r (A - B) g_2j — a commitment over the transformed basis is an MSM over the ORIGINAL one;
//   (2) collapse (:86-101, :162-170) folds a pair of points with (a reduced fraction of) rho = 1/(q e) resp. e, 1/e.  Tracking, per original
//       position i, the product coef_i of the rho's of the right halves i fell into, the level-k basis point at position p is
//       sum_{i >> k = p} coef_i P_i and every round commitment is an MSM over the original points with scalars sc_{i >> k} coef_i: ONE comb MSM
//       (csrc/comb.hip) of 2B rows per round.  Same group elements L, R as the folding route, hence the same proof bytes (tests).
//   k_ipb_init      makeNorm (:202-203): x' = s0 / (2r) + s1 / 2, y' = s1 / 2 - s0 / (2r); q = r^4 (:199); one inversion per proof
//   k_ipb_round     makeScalarsComs (:70-81, :155-158; BPCompose sums the two sub-arguments, Bulletproof.hs:258-261): the round sums sL, sR
//                   and the opening scalars per CURRENT position
//   k_ipb_expand    the two rows of 1 + linLen + nrmLen scalars over the original basis [g | H | G]
//   k_ipb_collapse  makeEs e = (1/e, e) (:68), s += sL / e + e sR, the vector folds (:86-101, :162-170), the coefficient products
//   k_ipb_witness   getWitness: Norm (nx x - ny y, nx x + ny y) (:222-223), Linear x (:160)
// `lp` lanes per proof in the per-proof kernels — the smallest power of two >= the longer of the two vectors, at least 8 — and 256 / lp proofs per
// workgroup (a single-value proof has 8 pairs and 6 linear entries: a workgroup per proof left 250 lanes idle); Fr in 10 x 26-bit limbs
// (fr26.hip.h), canonical 8 x 32 values in memory.
#include <algorithm>
#include <vector>
#include "comb.hpp"
#include "fr26.hip.h"
#include "modinv.hip.h"
#include "ipb.hpp"

namespace bppp {

// per-proof round state, [B][IPB_ST][8] canonical
enum { IPB_R = 0, IPB_Q = 1, IPB_QI = 2, IPB_NX = 3, IPB_S = 4, IPB_SL = 5, IPB_SR = 6, IPB_ST = 7 };

BPPP_DI fr ipb_pow(fr base, uint32_t e) { fr a = fr_one(); while (e) { if (e & 1u) a = fr_mul(a, base); base = fr_sqr(base); e >>= 1; } return a; }
BPPP_DI fe ipb_half() {                                   // (n + 1) / 2
  const fe n = fr_modulus(); uint32_t carry = 1; fe tt, h;
  for (int i = 0; i < 8; i++) { const uint64_t s = (uint64_t)n.v[i] + carry; tt.v[i] = (uint32_t)s; carry = (uint32_t)(s >> 32); }
  for (int i = 0; i < 8; i++) h.v[i] = (tt.v[i] >> 1) | (i < 7 ? tt.v[i + 1] << 31 : carry << 31);
  return h;
}
// sum of two values per lane over the `lp` lanes of a proof (lp a power of two dividing blockDim.x; the proofs of a workgroup run the same
// steps); result valid on lane 0 of the proof.  lds: blockDim.x * 2 * 8 words
BPPP_DI void ipb_seg_sum2(fe v[2], uint32_t *lds, uint32_t lp) {
  const uint32_t tid = threadIdx.x, t = tid % lp;
  for (int k = 0; k < 2; k++) for (int i = 0; i < 8; i++) lds[(tid * 2 + k) * 8 + i] = v[k].v[i];
  __syncthreads();
  for (uint32_t d = lp >> 1; d >= 1; d >>= 1) {
    if (t < d)
      for (int k = 0; k < 2; k++) {
        fe x, y;
        for (int i = 0; i < 8; i++) { x.v[i] = lds[(tid * 2 + k) * 8 + i]; y.v[i] = lds[((tid + d) * 2 + k) * 8 + i]; }
        x = fe_add<1>(x, y);
        for (int i = 0; i < 8; i++) lds[(tid * 2 + k) * 8 + i] = x.v[i];
      }
    __syncthreads();
  }
  if (t == 0) for (int k = 0; k < 2; k++) for (int i = 0; i < 8; i++) v[k].v[i] = lds[(tid * 2 + k) * 8 + i];
}

struct IpbDims { uint32_t nlen, llen, m0, xs, ls, T, batch, lp; };     // m0 = ceil(nlen / 2) pairs; xs, ls: per-proof strides of the X / Y and LC / LX buffers;
// lp: lanes per proof in the per-proof kernels (a power of two, 8 .. 256; blockDim.x / lp proofs per workgroup: single-value proofs have 8 pairs and 6 linear entries)

__global__ void __launch_bounds__(256) k_ipb_init(IpbDims D, const uint32_t *__restrict__ psv, const uint32_t *__restrict__ rr, const uint32_t *__restrict__ nrm,
                                                  const uint32_t *__restrict__ lc, const uint32_t *__restrict__ lx, uint32_t *__restrict__ X, uint32_t *__restrict__ Y,
                                                  uint32_t *__restrict__ LC, uint32_t *__restrict__ LX, uint32_t *__restrict__ cx, uint32_t *__restrict__ cy,
                                                  uint32_t *__restrict__ cl, uint32_t *__restrict__ stt) {
  const uint32_t b = blockIdx.x * (blockDim.x / D.lp) + threadIdx.x / D.lp, t = threadIdx.x % D.lp, bs = D.lp;
  if (b >= D.batch) return;
  const fe r8 = fe_load(rr + (size_t)b * 8);
  const fe ti8 = fe_modinv<1>(fe_dbl<1>(r8));             // 1 / (2r): every lane of the workgroup walks the same division steps
  const fr ti = fr_from_fe(ti8), half = fr_from_fe(ipb_half());
  for (uint32_t j = t; j < D.m0; j += bs) {
    const fr s0 = fr_load(nrm + ((size_t)b * D.nlen + 2 * j) * 8);
    const fr a = fr_mul(ti, s0);
    fr c = fr_zero();
    if (2 * j + 1 < D.nlen) c = fr_mul(half, fr_load(nrm + ((size_t)b * D.nlen + 2 * j + 1) * 8));
    fr_store(X + ((size_t)b * D.xs + j) * 8, fr_add(a, c));
    fr_store(Y + ((size_t)b * D.xs + j) * 8, fr_sub<1>(c, a));
    fe_store(cx + ((size_t)b * D.m0 + j) * 8, fe_one());
    fe_store(cy + ((size_t)b * D.m0 + j) * 8, fe_one());
  }
  for (uint32_t i = t; i < D.llen; i += bs) {
    fe_store(LC + ((size_t)b * D.ls + i) * 8, fe_load(lc + ((size_t)b * D.llen + i) * 8));
    fe_store(LX + ((size_t)b * D.ls + i) * 8, fe_load(lx + ((size_t)b * D.llen + i) * 8));
    fe_store(cl + ((size_t)b * D.llen + i) * 8, fe_one());
  }
  if (t == 0) {
    uint32_t *S = stt + (size_t)b * IPB_ST * 8;
    const fr r = fr_from_fe(r8), r2 = fr_sqr(r), t2 = fr_add(ti, ti), t4 = fr_sqr(fr_sqr(t2));       // q = r^4 (:199), q^-1 = (2 / (2r))^4
    fe_store(S + IPB_R * 8, r8); fr_store(S + IPB_Q * 8, fr_sqr(r2)); fr_store(S + IPB_QI * 8, t4);
    fe_store(S + IPB_NX * 8, fe_one()); fe_store(S + IPB_S * 8, fe_load(psv + (size_t)b * 8));
    fe_store(S + IPB_SL * 8, fe_zero()); fe_store(S + IPB_SR * 8, fe_zero());
  }
}

// og [B][4][me]: lgx, lhy, rgx, rhy over the current (even-padded) norm positions; ol [B][2][le]: ll, rl over the linear ones
__global__ void __launch_bounds__(256) k_ipb_round(IpbDims D, uint32_t mc, uint32_t lcn, uint32_t me, uint32_t le, const uint32_t *__restrict__ X,
                                                   const uint32_t *__restrict__ Y, const uint32_t *__restrict__ LC, const uint32_t *__restrict__ LX,
                                                   uint32_t *__restrict__ stt, uint32_t *__restrict__ og, uint32_t *__restrict__ ol) {
  __shared__ uint32_t lds[256 * 2 * 8];
  const uint32_t b_raw = blockIdx.x * (blockDim.x / D.lp) + threadIdx.x / D.lp, t = threadIdx.x % D.lp, bs = D.lp;
  const bool live = b_raw < D.batch;                          // a group past the end recomputes the last proof (it must keep up with the barriers) and stores nothing
  const uint32_t b = live ? b_raw : D.batch - 1;
  uint32_t *S = stt + (size_t)b * IPB_ST * 8;
  const fr q = fr_load(S + IPB_Q * 8), qi = fr_load(S + IPB_QI * 8), nx = fr_load(S + IPB_NX * 8);
  const fr q2 = fr_sqr(q);
  const uint32_t *xb = X + (size_t)b * D.xs * 8, *yb = Y + (size_t)b * D.xs * 8, *cb = LC + (size_t)b * D.ls * 8, *lb = LX + (size_t)b * D.ls * 8;
  uint32_t *lgx = og + (size_t)b * 4 * me * 8, *lhy = lgx + (size_t)me * 8, *rgx = lhy + (size_t)me * 8, *rhy = rgx + (size_t)me * 8;
  uint32_t *ll = ol + (size_t)b * 2 * le * 8, *rl = ll + (size_t)le * 8;
  fr l = fr_zero(), r_ = fr_zero();
  const uint32_t np = (mc + 1) / 2, lp = (lcn + 1) / 2;
  if (np) {
    fr w = ipb_pow(q2, t);
    const fr step = ipb_pow(q2, bs);
    for (uint32_t p = t; p < np && live; p += bs) {
      const bool has = 2 * p + 1 < mc;
      const fe xl8 = fe_load(xb + (size_t)(2 * p) * 8), yl8 = fe_load(yb + (size_t)(2 * p) * 8);
      const fe xr8 = has ? fe_load(xb + (size_t)(2 * p + 1) * 8) : fe_zero(), yr8 = has ? fe_load(yb + (size_t)(2 * p + 1) * 8) : fe_zero();
      const fr xl = fr_from_fe(xl8), yl = fr_from_fe(yl8), xr = fr_from_fe(xr8), yr = fr_from_fe(yr8);
      l = fr_addr(l, fr_mul(w, fr_mul(xl, yr)));
      r_ = fr_addr(r_, fr_mul(w, fr_mul(xr, yl)));
      fe_store(lgx + (size_t)(2 * p) * 8, fe_zero()); fr_store(lgx + (size_t)(2 * p + 1) * 8, fr_mul(qi, xl));       // L: IPF (qInv xL) gR yR hL
      fe_store(lhy + (size_t)(2 * p) * 8, yr8); fe_store(lhy + (size_t)(2 * p + 1) * 8, fe_zero());
      fr_store(rgx + (size_t)(2 * p) * 8, fr_mul(q, xr)); fe_store(rgx + (size_t)(2 * p + 1) * 8, fe_zero());        // R: IPF (q xR) gL yL hR
      fe_store(rhy + (size_t)(2 * p) * 8, fe_zero()); fe_store(rhy + (size_t)(2 * p + 1) * 8, yl8);
      w = fr_mul(w, step);
    }
  }
  fr sl = fr_zero(), sr = fr_zero();
  for (uint32_t p = t; p < lp && live; p += bs) {
    const bool has = 2 * p + 1 < lcn;
    const fe xl8 = fe_load(lb + (size_t)(2 * p) * 8), xr8 = has ? fe_load(lb + (size_t)(2 * p + 1) * 8) : fe_zero();
    const fr cl_ = fr_load(cb + (size_t)(2 * p) * 8), cr = has ? fr_load(cb + (size_t)(2 * p + 1) * 8) : fr_zero();
    sl = fr_addr(sl, fr_mul(cr, fr_from_fe(xl8))); sr = fr_addr(sr, fr_mul(cl_, fr_from_fe(xr8)));                    // L: LF cR xL gR;  R: LF cL xR gL
    fe_store(ll + (size_t)(2 * p) * 8, fe_zero()); fe_store(ll + (size_t)(2 * p + 1) * 8, xl8);
    fe_store(rl + (size_t)(2 * p) * 8, xr8); fe_store(rl + (size_t)(2 * p + 1) * 8, fe_zero());
  }
  fe v[2] = {fr_to_fe(l), fr_to_fe(r_)};
  ipb_seg_sum2(v, lds, D.lp);
  __syncthreads();
  fe u[2] = {fr_to_fe(sl), fr_to_fe(sr)};
  ipb_seg_sum2(u, lds, D.lp);
  if (t == 0 && live) {
    const fr kk = fr_mul_int(nx, 4);                      // s nx ny with s = 4 (makeNorm), ny = 1 in the unscaled recursion
    fr sL = fr_from_fe(u[0]), sR = fr_from_fe(u[1]);
    if (mc) {
      sL = fr_add(sL, fr_mul(fr_mul(kk, q), fr_from_fe(v[0])));
      sR = fr_add(sR, fr_mul(fr_mul(kk, q2), fr_from_fe(v[1])));
    }
    fr_store(S + IPB_SL * 8, sL); fr_store(S + IPB_SR * 8, sR);
  }
}

// full[2b + side][0] = sL / sR on g, [1 .. llen] the linear part, [1 + llen ..] the norm part over the ORIGINAL points
__global__ void __launch_bounds__(256) k_ipb_expand(IpbDims D, uint32_t round, uint32_t me, uint32_t le, const uint32_t *__restrict__ stt, const uint32_t *__restrict__ og,
                                                    const uint32_t *__restrict__ ol, const uint32_t *__restrict__ cx, const uint32_t *__restrict__ cy,
                                                    const uint32_t *__restrict__ cl, uint32_t *__restrict__ full) {
  const uint32_t inst = blockIdx.y, b = inst >> 1, side = inst & 1u, pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= D.T) return;
  const uint32_t *S = stt + (size_t)b * IPB_ST * 8;
  fe v;
  if (pos == 0) v = fe_load(S + (side ? IPB_SR : IPB_SL) * 8);
  else if (pos <= D.llen) {
    const uint32_t i = pos - 1;
    v = fe_load(ol + (((size_t)b * 2 + side) * le + (i >> round)) * 8);
    if (round && !fe_is_zero(v)) v = fr_to_fe(fr_mul(fr_from_fe(v), fr_load(cl + ((size_t)b * D.llen + i) * 8)));
  } else {
    const uint32_t i2 = pos - 1 - D.llen, i = i2 >> 1, p = i >> round;
    const uint32_t *gx = og + (((size_t)b * 4 + 2 * side) * me + p) * 8, *hy = gx + (size_t)me * 8;
    const fe g8 = fe_load(gx), h8 = fe_load(hy);
    fr A = fr_from_fe(g8), Bv = fr_from_fe(h8);
    if (round) {
      if (!fe_is_zero(g8)) A = fr_mul(A, fr_load(cx + ((size_t)b * D.m0 + i) * 8));
      if (!fe_is_zero(h8)) Bv = fr_mul(Bv, fr_load(cy + ((size_t)b * D.m0 + i) * 8));
    }
    if (i2 & 1u) v = fr_to_fe(fr_add(A, Bv));                                        // on g_2i+1
    else v = fr_to_fe(fr_mul(fr_load(S + IPB_R * 8), fr_sub<1>(A, Bv)));             // on g_2i
  }
  fe_store(full + ((size_t)inst * D.T + pos) * 8, v);
}

// the challenge of a round; src / dst: the two halves of the double buffers
__global__ void __launch_bounds__(256) k_ipb_collapse(IpbDims D, uint32_t round, uint32_t mc, uint32_t lcn, const uint32_t *__restrict__ es, uint32_t *__restrict__ stt,
                                                      const uint32_t *__restrict__ X, const uint32_t *__restrict__ Y, const uint32_t *__restrict__ LC,
                                                      const uint32_t *__restrict__ LX, uint32_t *__restrict__ Xo, uint32_t *__restrict__ Yo, uint32_t *__restrict__ LCo,
                                                      uint32_t *__restrict__ LXo, uint32_t *__restrict__ cx, uint32_t *__restrict__ cy, uint32_t *__restrict__ cl,
                                                      uint32_t *__restrict__ flag) {
  const uint32_t b_raw = blockIdx.x * (blockDim.x / D.lp) + threadIdx.x / D.lp, t = threadIdx.x % D.lp, bs = D.lp;
  const bool live = b_raw < D.batch;
  const uint32_t b = live ? b_raw : D.batch - 1;
  uint32_t *S = stt + (size_t)b * IPB_ST * 8;
  const fe e8 = fe_load(es + (size_t)b * 8);
  const fe ei8 = fe_modinv<1>(e8);                        // 0 -> 0; a zero challenge fails the batch (flag), as the host version does
  const fr e = fr_from_fe(e8), ei = fr_from_fe(ei8);
  const fr q = fr_load(S + IPB_Q * 8), qi = fr_load(S + IPB_QI * 8);
  const fr eq = fr_mul(e, q), rhox = fr_mul(qi, ei);
  const fe s8 = fe_load(S + IPB_S * 8), sl8 = fe_load(S + IPB_SL * 8), sr8 = fe_load(S + IPB_SR * 8), nx8 = fe_load(S + IPB_NX * 8);
  __syncthreads();                                         // every lane holds the old state before lane 0 advances it
  if (t == 0 && live) {
    if (fe_is_zero(e8)) atomicOr(flag, 1u);
    fr_store(S + IPB_S * 8, fr_add(fr_from_fe(s8), fr_add(fr_mul(ei, fr_from_fe(sl8)), fr_mul(e, fr_from_fe(sr8)))));
    if (mc) { fr_store(S + IPB_NX * 8, fr_mul(fr_from_fe(nx8), qi)); fr_store(S + IPB_Q * 8, fr_sqr(q)); fr_store(S + IPB_QI * 8, fr_sqr(qi)); }
  }
  const uint32_t *xb = X + (size_t)b * D.xs * 8, *yb = Y + (size_t)b * D.xs * 8;
  for (uint32_t p = t; 2 * p < mc && live; p += bs) {
    fr nx = fr_load(xb + (size_t)(2 * p) * 8), ny = fr_load(yb + (size_t)(2 * p) * 8);
    if (2 * p + 1 < mc) {
      nx = fr_add(nx, fr_mul(eq, fr_load(xb + (size_t)(2 * p + 1) * 8)));
      ny = fr_add(ny, fr_mul(ei, fr_load(yb + (size_t)(2 * p + 1) * 8)));
    }
    fr_store(Xo + ((size_t)b * D.xs + p) * 8, nx); fr_store(Yo + ((size_t)b * D.xs + p) * 8, ny);
  }
  if (mc && live)
    for (uint32_t i = t; i < D.m0; i += bs)
      if ((i >> round) & 1u) {
        uint32_t *a = cx + ((size_t)b * D.m0 + i) * 8, *c = cy + ((size_t)b * D.m0 + i) * 8;
        fr_store(a, fr_mul(fr_load(a), rhox)); fr_store(c, fr_mul(fr_load(c), e));
      }
  const uint32_t *cb = LC + (size_t)b * D.ls * 8, *lb = LX + (size_t)b * D.ls * 8;
  for (uint32_t p = t; 2 * p < lcn && live; p += bs) {
    fr nc = fr_load(cb + (size_t)(2 * p) * 8), nl = fr_load(lb + (size_t)(2 * p) * 8);
    if (2 * p + 1 < lcn) {
      nc = fr_add(nc, fr_mul(ei, fr_load(cb + (size_t)(2 * p + 1) * 8)));
      nl = fr_add(nl, fr_mul(e, fr_load(lb + (size_t)(2 * p + 1) * 8)));
    }
    fr_store(LCo + ((size_t)b * D.ls + p) * 8, nc); fr_store(LXo + ((size_t)b * D.ls + p) * 8, nl);
  }
  if (lcn && live)
    for (uint32_t i = t; i < D.llen; i += bs)
      if ((i >> round) & 1u) { uint32_t *c = cl + ((size_t)b * D.llen + i) * 8; fr_store(c, fr_mul(fr_load(c), ei)); }
}

__global__ void __launch_bounds__(64) k_ipb_witness(IpbDims D, uint32_t batch, uint32_t mc, uint32_t lcn, const uint32_t *__restrict__ stt, const uint32_t *__restrict__ X,
                                                    const uint32_t *__restrict__ Y, const uint32_t *__restrict__ LX, uint32_t *__restrict__ wn, uint32_t *__restrict__ wl) {
  const uint64_t g = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  const uint32_t per = mc + lcn;
  if (g >= (uint64_t)batch * per) return;
  const uint32_t b = (uint32_t)(g / per), j = (uint32_t)(g % per);
  if (j < mc) {
    const fr a = fr_mul(fr_load(stt + ((size_t)b * IPB_ST + IPB_NX) * 8), fr_load(X + ((size_t)b * D.xs + j) * 8)), y = fr_load(Y + ((size_t)b * D.xs + j) * 8);
    fr_store(wn + ((size_t)b * 2 * mc + 2 * j) * 8, fr_sub<1>(a, y));
    fr_store(wn + ((size_t)b * 2 * mc + 2 * j + 1) * 8, fr_add(a, y));
  } else fe_store(wl + ((size_t)b * lcn + (j - mc)) * 8, fe_load(LX + ((size_t)b * D.ls + (j - mc)) * 8));
}

static size_t ev(size_t v) { return v + (v & 1); }

size_t ipb_work_bytes(size_t B, size_t nlen, size_t llen) {
  Carver cv(nullptr, 0);
  const size_t m0 = (nlen + 1) / 2, xs = ev(m0) + 2, ls = ev(llen) + 2, T = 1 + llen + nlen;
  for (int k = 0; k < 2; k++) { cv.take<uint32_t>(B * xs * 8); cv.take<uint32_t>(B * xs * 8); cv.take<uint32_t>(B * ls * 8); cv.take<uint32_t>(B * ls * 8); }
  cv.take<uint32_t>(B * std::max<size_t>(m0, 1) * 8); cv.take<uint32_t>(B * std::max<size_t>(m0, 1) * 8); cv.take<uint32_t>(B * std::max<size_t>(llen, 1) * 8);
  cv.take<uint32_t>(B * IPB_ST * 8); cv.take<uint32_t>(B * 4 * ev(m0) * 8 + 8); cv.take<uint32_t>(B * 2 * ev(llen) * 8 + 8); cv.take<uint32_t>(2 * B * T * 8);
  cv.take<uint32_t>(comb_scratch_bytes(2 * B) / 4 + 16); cv.take<uint32_t>(16);
  return cv.off;
}

int ipb_prove_stream(bppp_ctx *ctx, const CombTable *comb, RppTranscript &tr, size_t first_call, size_t B, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl,
                     const uint32_t *d_psv, const uint32_t *d_rr, const uint32_t *d_nrm, const uint32_t *d_lc, const uint32_t *d_lx, void *work, size_t work_bytes,
                     uint32_t *d_resp, uint32_t *d_wn, uint32_t *d_wl, uint32_t **d_flag_out) {
  if (!ctx || !comb || !B || nlen + llen == 0 || comb->T != 1 + llen + nlen || work_bytes < ipb_work_bytes(B, nlen, llen)) return BPPP_ERR_ARG;
  hipStream_t st = ctx->stream;
  const size_t m0 = (nlen + 1) / 2, xs = ev(m0) + 2, ls = ev(llen) + 2, T = 1 + llen + nlen;
  // lanes per proof: the smallest power of two >= the longer vector (at least 8), 256 / lp proofs per 256-lane workgroup
  uint32_t lp = 8;
  while (lp < 256 && lp < std::max(m0, llen)) lp <<= 1;
  IpbDims D{(uint32_t)nlen, (uint32_t)llen, (uint32_t)m0, (uint32_t)xs, (uint32_t)ls, (uint32_t)T, (uint32_t)B, lp};
  const unsigned pgrid = (unsigned)((B * lp + 255) / 256);   // workgroups of the per-proof kernels
  Carver cv(work, work_bytes);
  uint32_t *X[2], *Y[2], *LC[2], *LX[2];
  for (int i = 0; i < 2; i++) { X[i] = cv.take<uint32_t>(B * xs * 8); Y[i] = cv.take<uint32_t>(B * xs * 8); LC[i] = cv.take<uint32_t>(B * ls * 8); LX[i] = cv.take<uint32_t>(B * ls * 8); }
  uint32_t *cx = cv.take<uint32_t>(B * std::max<size_t>(m0, 1) * 8), *cy = cv.take<uint32_t>(B * std::max<size_t>(m0, 1) * 8), *cl = cv.take<uint32_t>(B * std::max<size_t>(llen, 1) * 8);
  uint32_t *stt = cv.take<uint32_t>(B * IPB_ST * 8), *og = cv.take<uint32_t>(B * 4 * ev(m0) * 8 + 8), *ol = cv.take<uint32_t>(B * 2 * ev(llen) * 8 + 8);
  uint32_t *full = cv.take<uint32_t>(2 * B * T * 8), *cscratch = cv.take<uint32_t>(comb_scratch_bytes(2 * B) / 4 + 16), *flag = cv.take<uint32_t>(16);
  BPPP_HIP(ctx, hipMemsetAsync(flag, 0, 4, st));
  k_ipb_init<<<dim3(pgrid), dim3(256), 0, st>>>(D, d_psv, d_rr, d_nrm, d_lc, d_lx, X[0], Y[0], LC[0], LX[0], cx, cy, cl, stt);
  BPPP_HIP(ctx, hipGetLastError());
  size_t mc = m0, lcn = llen;
  int cur = 0;
  for (size_t round = 0; round < k; round++) {
    const size_t me = ev(mc), le = ev(lcn);
    k_ipb_round<<<dim3(pgrid), dim3(256), 0, st>>>(D, (uint32_t)mc, (uint32_t)lcn, (uint32_t)me, (uint32_t)le, X[cur], Y[cur], LC[cur], LX[cur], stt, og, ol);
    k_ipb_expand<<<dim3((unsigned)((T + 255) / 256), (unsigned)(2 * B)), dim3(256), 0, st>>>(D, (uint32_t)round, (uint32_t)me, (uint32_t)le, stt, og, ol, cx, cy, cl, full);
    BPPP_HIP(ctx, hipGetLastError());
    uint32_t *lr = d_resp + round * B * 32;
    int rc = comb_msm(comb, full, 2 * B, lr, st, false, 0, cscratch, comb_scratch_bytes(2 * B));
    if (rc) return fail(ctx, rc, bppp_last_error(comb->ctx));
    rc = tr.call(lr, first_call + round); if (rc) return rc;
    k_ipb_collapse<<<dim3(pgrid), dim3(256), 0, st>>>(D, (uint32_t)round, (uint32_t)mc, (uint32_t)lcn, tr.es, stt, X[cur], Y[cur], LC[cur],
                                                                                        LX[cur], X[1 - cur], Y[1 - cur], LC[1 - cur], LX[1 - cur], cx, cy, cl, flag);
    BPPP_HIP(ctx, hipGetLastError());
    mc = (mc + 1) / 2; lcn = (lcn + 1) / 2; cur = 1 - cur;
  }
  if (2 * mc != fn || lcn != fl) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: the argument ended at an unexpected length");
  const uint64_t n = (uint64_t)B * (mc + lcn);
  if (n) k_ipb_witness<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(D, (uint32_t)B, (uint32_t)mc, (uint32_t)lcn, stt, X[cur], Y[cur], LX[cur], d_wn, d_wl);
  BPPP_HIP(ctx, hipGetLastError());
  *d_flag_out = flag;
  return BPPP_OK;
}

}  // namespace bppp
