// fold.hip — batched basis fold: out[j] = b * G[2j] + a * G[2j+1] with ONE (a, b) for all pairs.
//
// Replaces `collapsePoints b a gL gR = projectivePairIP (b, gL) (a, gR)` (src/Bulletproof.hs:213-214,
// src/Commitment.hs:343-353) mapped over adjacent pairs by mapHalves (src/Bulletproof.hs:88-90), as
// called from the three `collapse` methods (NormArgument.hs:71, :129; InnerProductArgument.hs:100-101).
//
// The reference runs a 129-row Straus loop per pair and pays one field inversion per pair
// (normalizeBasis on two points, Commitment.hs:347).  Here every lane owns one pair and walks the joint-sparse-form
// schedule of (b', a') against its own four-entry table in LDS (foldcore.hip.h); inputs are already affine.
#include <string.h>
#include "ctx.hpp"
#include "foldcore.hip.h"
#include "hostmath.hpp"

namespace bppp {

struct FoldK { uint32_t dig[FOLD_DIGIT_WORDS]; int bneg, aneg; };

__global__ void __launch_bounds__(64) k_fold_points(const uint32_t *__restrict__ pts, uint32_t n, FoldK K, uint32_t *__restrict__ out) {
  __shared__ uint32_t tab[FOLD_TAB_WORDS];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, np = (n + 1) / 2;
  aff GL = aff_inf(), GR = aff_inf();
  if (j < np) {
    GL = aff_cneg(aff_load(pts + (size_t)(2 * j) * 16), K.bneg != 0);
    if (2 * j + 1 < n) GR = aff_cneg(aff_load(pts + (size_t)(2 * j + 1) * 16), K.aneg != 0);
  }
  aff r = fold_pair_jsf(GL, GR, K.dig, tab, threadIdx.x);
  if (j < np) aff_store(out + (size_t)j * 16, r);
}

// Several independent folds in ONE launch (norm basis, linear basis, ... of one collapse): the pairs of all segments are
// numbered consecutively, so wavefronts are full even when a segment is short; a lane reads the digits of its own segment.
struct FoldSeg { FoldK K; const uint32_t *pts; uint32_t *out; uint32_t n, first_pair; };
struct FoldSegs { FoldSeg s[3]; int nseg; uint32_t total_pairs; };
__global__ void __launch_bounds__(64) k_fold_points_multi(FoldSegs S_) {
  const FoldSegs *S = &S_;
  __shared__ uint32_t tab[FOLD_TAB_WORDS];
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  int si = 0;
  if (S->nseg > 1 && g >= S->s[1].first_pair) si = 1;
  if (S->nseg > 2 && g >= S->s[2].first_pair) si = 2;
  const FoldSeg *sg = &S->s[si];
  const uint32_t j = g - sg->first_pair;
  const bool active = g < S->total_pairs;
  aff GL = aff_inf(), GR = aff_inf();
  if (active) {
    GL = aff_cneg(aff_load(sg->pts + (size_t)(2 * j) * 16), sg->K.bneg != 0);
    if (2 * j + 1 < sg->n) GR = aff_cneg(aff_load(sg->pts + (size_t)(2 * j + 1) * 16), sg->K.aneg != 0);
  }
  aff r = fold_pair_jsf(GL, GR, sg->K.dig, tab, threadIdx.x);
  if (active) aff_store(sg->out + (size_t)j * 16, r);
}

// The same fold for the reference's Eisenstein configuration (SplitScalar (FastPrime p), src/Commitment.hs:293-306; FastInnerProduct
// of the FastPrime instance, :374-398; rationalReducedScalarLength = 65): the reduced scalars are Eisenstein integers
// b' = b0 + b1 w, a' = a0 + a1 w with ~65-bit components and w acts on a point as the endomorphism lambda (x, y) = (beta x, y)
// (cmConj, src/Data/Curve/CM.hs:25-27):   out = b0 GL + b1 lambda GL + a0 GR + a1 lambda GR.
// Each half is a two-scalar joint-sparse-form walk over (P, lambda P) — 66 rows instead of 130, but two of them and one more
// addition: the same work as the integer fold (measured: no faster), so this is the PARITY option for that configuration.
struct FoldEisK { uint32_t digb[FOLD_DIGIT_WORDS], diga[FOLD_DIGIT_WORDS]; int b0neg, b1neg, a0neg, a1neg; };
BPPP_DI aff aff_lambda(const aff &p) {             // (beta x, y); infinity stays infinity
  const fe beta = {{0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu}};
  aff r; r.x = fq_normalize(fq_mul(p.x, fq_from_fe(beta))); r.y = p.y;
  return r;
}
__global__ void __launch_bounds__(64) k_fold_points_eis(const uint32_t *__restrict__ pts, uint32_t n, FoldEisK K, uint32_t *__restrict__ out) {
  __shared__ uint32_t tab[FOLD_TAB_WORDS];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, np = (n + 1) / 2;
  aff GL = aff_inf(), GR = aff_inf();
  if (j < np) {
    GL = aff_load(pts + (size_t)(2 * j) * 16);
    if (2 * j + 1 < n) GR = aff_load(pts + (size_t)(2 * j + 1) * 16);
  }
  const aff A = fold_pair_jsf(aff_cneg(GL, K.b0neg != 0), aff_cneg(aff_lambda(GL), K.b1neg != 0), K.digb, tab, threadIdx.x);
  const aff B = fold_pair_jsf(aff_cneg(GR, K.a0neg != 0), aff_cneg(aff_lambda(GR), K.a1neg != 0), K.diga, tab, threadIdx.x);
  xyzz acc = xyzz_from_aff(A);
  xyzz_madd(acc, B);
  if (j < np) aff_store(out + (size_t)j * 16, xyzz_to_aff(acc));
}
int fold_points_eis_run(bppp_ctx *ctx, const uint64_t b_mag[4], const int b_neg[2], const uint64_t a_mag[4], const int a_neg[2], const void *d_pts, size_t n,
                        void *d_out) {
  if (n == 0) return BPPP_OK;
  if (!d_pts || !d_out || !b_mag || !a_mag || !b_neg || !a_neg || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "fold_points_eis: bad arguments");
  FoldEisK K; memset(&K, 0, sizeof K);
  for (int k = 0; k < 2; k++)
    if ((b_mag[2 * k + 1] >> 2) || (a_mag[2 * k + 1] >> 2)) return fail(ctx, BPPP_ERR_ARG, "fold_points_eis: component exceeds 66 bits");
  const uint64_t b0[3] = {b_mag[0], b_mag[1], 0}, b1[3] = {b_mag[2], b_mag[3], 0}, a0[3] = {a_mag[0], a_mag[1], 0}, a1[3] = {a_mag[2], a_mag[3], 0};
  bppp_host::jsf_recode(b0, b1, K.digb);
  bppp_host::jsf_recode(a0, a1, K.diga);
  K.b0neg = b_neg[0]; K.b1neg = b_neg[1]; K.a0neg = a_neg[0]; K.a1neg = a_neg[1];
  const uint32_t np = (uint32_t)((n + 1) / 2);
  k_fold_points_eis<<<dim3((np + 63) / 64), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_pts, (uint32_t)n, K, (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

// pointX (app/Main.hs:68-72): y = sqrt(x^3 + 7) = (x^3+7)^((p+1)/4) since p = 3 mod 4; even root.
__global__ void __launch_bounds__(64) k_lift_x(const uint32_t *__restrict__ xs, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe xe = fe_load(xs + (size_t)i * 8);
  fe t;
  bool ok = raw_sub(t, xe, fp_modulus()) != 0;      // x < p
  fq x = fq_from_fe(xe);
  fq seven = fq_zero(); seven.n[0] = 7;
  fq rhs = fq_add(fq_mul(fq_sqr(x), x), seven);     // magnitude 2
  fq acc = fq_sqrt_candidate(rhs);                  // rhs^((p+1)/4): 253 squarings + 13 multiplications
  ok = ok && fq_normalizes_to_zero(fq_sub<2>(fq_sqr(acc), rhs));
  fq y = fq_normalize(acc);
  if (y.n[0] & 1u) y = fq_normalize(fq_neg<1>(y));
  aff r; r.x = x; r.y = y;
  if (!ok) r = aff_inf();
  aff_store(out + (size_t)i * 16, r);
}
int lift_x_run(bppp_ctx *ctx, const void *d_x, size_t n, void *d_out) {
  if (n == 0) return BPPP_OK;
  if (!d_x || !d_out || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "lift_x: bad input");
  k_lift_x<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}

static int make_fold_k(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg, FoldK &K) {
  memset(&K, 0, sizeof K);
  if ((b_mag[2] >> 1) || (a_mag[2] >> 1)) return fail(ctx, BPPP_ERR_ARG, "fold_points: reduced scalar exceeds 129 bits");
  bppp_host::jsf_recode(b_mag, a_mag, K.dig);
  K.bneg = b_neg; K.aneg = a_neg;
  return BPPP_OK;
}
// up to three folds in one launch; entries with n == 0 are skipped.  No stream synchronisation.
int fold_points_multi_run(bppp_ctx *ctx, int nseg, const uint64_t *const b_mag[], const int b_neg[], const uint64_t *const a_mag[], const int a_neg[],
                          const void *const d_pts[], const size_t n[], void *const d_out[]) {
  FoldSegs S; memset(&S, 0, sizeof S);
  uint32_t pairs = 0;
  for (int i = 0; i < nseg && i < 3; i++) {
    if (!n[i]) continue;
    if (!d_pts[i] || !d_out[i] || n[i] >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "fold_points_multi: bad segment");
    FoldSeg &sg = S.s[S.nseg];
    int rc = make_fold_k(ctx, b_mag[i], b_neg[i], a_mag[i], a_neg[i], sg.K); if (rc) return rc;
    sg.pts = (const uint32_t *)d_pts[i]; sg.out = (uint32_t *)d_out[i]; sg.n = (uint32_t)n[i]; sg.first_pair = pairs;
    pairs += (uint32_t)((n[i] + 1) / 2);
    S.nseg++;
  }
  if (!S.nseg) return BPPP_OK;
  S.total_pairs = pairs;
  k_fold_points_multi<<<dim3((pairs + 63) / 64), dim3(64), 0, ctx->stream>>>(S);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

int fold_points_run(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg,
                    const void *d_pts, size_t n, void *d_out) {
  if (n == 0) return BPPP_OK;
  if (!d_pts || !d_out || !b_mag || !a_mag) return fail(ctx, BPPP_ERR_ARG, "fold_points: null pointer");
  if (n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "fold_points: n too large");
  FoldK K;
  { int rc = make_fold_k(ctx, b_mag, b_neg, a_mag, a_neg, K); if (rc) return rc; }
  uint32_t np = (uint32_t)((n + 1) / 2);
  k_fold_points<<<dim3((np + 63) / 64), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_pts, (uint32_t)n, K, (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

}  // namespace bppp
