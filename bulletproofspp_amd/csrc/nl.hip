// nl.hip — the norm-linear argument (NL flavour) with its vectors and basis resident in HBM.
//
// Device counterpart of `NormLinear f v s = BPCompose (Norm f) (Linear f)` (src/Bulletproof/NormArgument.hs:153-178,
// src/Bulletproof.hs:225-273) wrapped in a PedersenScalarVector (src/Commitment.hs:487-501), i.e. the object
// proveRoundM / verifyBPM (src/Bulletproof.hs:346-378) work on.  The reference's container hook for this is
// `class BPCollection` ("Intended to allow substituting other containers", src/Bulletproof.hs:27-66).
//
//   bppp_nl_create            makeNormLinearBP' (NormArgument.hs:162) + makePSV (Commitment.hs:490-491)
//   bppp_nl_round_commit      makeScalarsComs (NormArgument.hs:113-118, :56-59; Bulletproof.hs:258-261) and the two
//                             `commit`s of proveRoundM (Bulletproof.hs:348-350) as ONE batched MSM over a shared basis
//   bppp_nl_round_collapse    makeEs + scalar update + collapse (Bulletproof.hs:352-354; NormArgument.hs:123-129, :64-71)
//   bppp_nl_get_witness       getWitness (NormArgument.hs:121, :62; Bulletproof.hs:264)
//   bppp_nl_verify            expandChallenges (NormArgument.hs:73-81, :131-145; Bulletproof.hs:268-269) + the single
//                             commit of verifyBPM over `verifyWith` (Bulletproof.hs:362-368, :375-377)
//
// The Fiat-Shamir oracle stays with the caller (it is injected in the reference too: src/ZKP.hs:73-77): per
// round 2 x 64 B (X, R) go to the host and one 32-B challenge comes back.
#include <string.h>
#include <vector>
#include "ctx.hpp"
#include "ec.hip.h"
#include "hostmath.hpp"

namespace bppp {
int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);
int fold_points_run(bppp_ctx *, const uint64_t *, int, const uint64_t *, int, const void *, size_t, void *);
int norm_round_sums_run(bppp_ctx *, const void *, size_t, const uint64_t *, uint64_t *, uint64_t *);
int lin_round_sums_run(bppp_ctx *, const void *, const void *, size_t, uint64_t *, uint64_t *);
int fold_scalars_launch(bppp_ctx *, const uint64_t *, const uint64_t *, const void *, size_t, void *);
int fold_points_multi_run(bppp_ctx *, int, const uint64_t *const[], const int[], const uint64_t *const[], const int[], const void *const[], const size_t[], void *const[]);
int tensor_run(bppp_ctx *, const uint64_t *, size_t, const uint64_t *, const uint64_t *, size_t, void *);

struct FrK { uint32_t v[8]; };
static FrK frk(const bppp_host::U256 &x) {
  FrK r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)x.w[i]; r.v[2 * i + 1] = (uint32_t)(x.w[i] >> 32); }
  return r;
}
BPPP_DI fe fe_of(const FrK &a) { fe r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i]; return r; }

// X / R opening scalars of one sub-argument laid over the FULL (even-padded) basis slice:
//   xs[2j] = a * xR_j, xs[2j+1] = b * xL_j          (X opening: NormArgument.hs:117 with a = q, b = qinv; :59 with a = b = 1)
//   rs[2j] = 0,        rs[2j+1] = xR_j               (R opening keeps the ORIGINAL right element, foldXR NormArgument.hs:28)
__global__ void __launch_bounds__(256) k_nl_round_scalars(const uint32_t *__restrict__ x, uint32_t n, int scale, FrK a_, FrK b_,
                                                          uint32_t *__restrict__ xs, uint32_t *__restrict__ rs) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (n + 1) / 2) return;
  fe xl = fe_load(x + (size_t)(2 * j) * 8);
  fe xr = (2 * j + 1 < n) ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
  fe a = xr, b = xl;
  if (scale) { a = fe_mul<1>(fe_of(a_), xr); b = fe_mul<1>(fe_of(b_), xl); }
  fe_store(xs + (size_t)(2 * j) * 8, a);
  fe_store(xs + (size_t)(2 * j + 1) * 8, b);
  fe_store(rs + (size_t)(2 * j) * 8, fe_zero());
  fe_store(rs + (size_t)(2 * j + 1) * 8, xr);
}
// out[i] = pub[i] - (i < nt ? t[i] : 0)      (exp of expandChallenges: NormArgument.hs:142-143, :80-81; zipWithDef' default 0)
__global__ void __launch_bounds__(256) k_sub_scalars(const uint32_t *__restrict__ pub, const uint32_t *__restrict__ t, uint32_t n, uint32_t nt,
                                                     uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe p = fe_load(pub + (size_t)i * 8);
  if (i < nt) p = fe_sub<1>(p, fe_load(t + (size_t)i * 8));
  fe_store(out + (size_t)i * 8, p);
}

}  // namespace bppp

using namespace bppp;
using namespace bppp_host;

struct bppp_nl {
  bppp_ctx *ctx;
  size_t n, l;                 // current norm / linear lengths
  size_t cap;                  // capacity of the point / scalar buffers (initial even-padded size + 1)
  uint32_t *x[2], *lx[2], *lc[2];   // Fr vectors (ping-pong)
  uint32_t *P[2];              // [G (n, padded to even with infinity) | H (l, padded) | g]
  uint32_t *sc;                // scalars of the two round MSMs: [2][cap]
  int cur;
  U256 q, qinv, nn, ln, s, scomp;   // Norm q, q^-1, nrmlz''; Linear nrmlz''; PSV scalar; scalarComp
  U256 sX, sR;                 // kept between round_commit and round_collapse
};

static size_t ev(size_t v) { return v + (v & 1); }
static const Mod &R_() { return FR(); }

#define NL_HIP(nl, call)                                                                                   \
  do {                                                                                                     \
    hipError_t _e = (call);                                                                                \
    if (_e != hipSuccess) return bppp::fail((nl)->ctx, BPPP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" {

void bppp_nl_destroy(bppp_nl *nl) {
  if (!nl) return;
  bppp_ctx *ctx = nl->ctx;                 // kept alive by this handle's reference even after bppp_ctx_destroy
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 2; k++) { hipFree(nl->x[k]); hipFree(nl->lx[k]); hipFree(nl->lc[k]); hipFree(nl->P[k]); }
  hipFree(nl->sc);
  delete nl;
  ctx_release(ctx);
}

int bppp_nl_create(bppp_ctx *ctx, const uint64_t s[4], const uint64_t g_xy[8], const uint64_t q[4], const uint64_t *norm_x,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy,
                   size_t llen, bppp_nl **out) {
  if (!ctx || !out || !s || !g_xy || !q || ctx_closed(ctx)) return BPPP_ERR_ARG;
  if ((nlen && (!norm_x || !norm_g_xy)) || (llen && (!lin_c || !lin_x || !lin_h_xy))) return fail(ctx, BPPP_ERR_ARG, "nl_create: null vector");
  if (nlen + llen == 0 || nlen >= (1u << 30) || llen >= (1u << 30)) return fail(ctx, BPPP_ERR_ARG, "nl_create: bad lengths");
  hipSetDevice(ctx->device);
  bppp_nl *nl = new bppp_nl();
  memset(nl, 0, sizeof *nl);
  nl->ctx = ctx; ctx_retain(ctx); nl->n = nlen; nl->l = llen; nl->cap = ev(nlen) + ev(llen) + 1; nl->cur = 0;
  nl->q = U256::load(q); nl->qinv = minv(nl->q, R_()); nl->nn = U256::one(); nl->ln = U256::one();
  nl->s = U256::load(s); nl->scomp = U256::one();
  hipStream_t st = ctx->stream;
  for (int k = 0; k < 2; k++) {
    if (hipMalloc(&nl->x[k], (ev(nlen) + 2) * 32) != hipSuccess || hipMalloc(&nl->lx[k], (ev(llen) + 2) * 32) != hipSuccess ||
        hipMalloc(&nl->lc[k], (ev(llen) + 2) * 32) != hipSuccess || hipMalloc(&nl->P[k], nl->cap * 64) != hipSuccess) {
      bppp_nl_destroy(nl);
      return fail(ctx, BPPP_ERR_HIP, "nl_create: hipMalloc failed");
    }
  }
  if (hipMalloc(&nl->sc, 2 * nl->cap * 32) != hipSuccess) { bppp_nl_destroy(nl); return fail(ctx, BPPP_ERR_HIP, "nl_create: hipMalloc failed"); }
  auto fill = [&]() -> int {                 // any failure below goes through ONE cleanup: the handle is destroyed
  NL_HIP(nl, hipMemsetAsync(nl->P[0], 0, nl->cap * 64, st));
  if (nlen) {
    NL_HIP(nl, hipMemcpyAsync(nl->x[0], norm_x, nlen * 32, hipMemcpyHostToDevice, st));
    NL_HIP(nl, hipMemcpyAsync(nl->P[0], norm_g_xy, nlen * 64, hipMemcpyHostToDevice, st));
  }
  if (llen) {
    NL_HIP(nl, hipMemcpyAsync(nl->lc[0], lin_c, llen * 32, hipMemcpyHostToDevice, st));
    NL_HIP(nl, hipMemcpyAsync(nl->lx[0], lin_x, llen * 32, hipMemcpyHostToDevice, st));
    NL_HIP(nl, hipMemcpyAsync(nl->P[0] + ev(nlen) * 16, lin_h_xy, llen * 64, hipMemcpyHostToDevice, st));
  }
  NL_HIP(nl, hipMemcpyAsync(nl->P[0] + (ev(nlen) + ev(llen)) * 16, g_xy, 64, hipMemcpyHostToDevice, st));
  NL_HIP(nl, hipStreamSynchronize(st));
  return BPPP_OK;
  };
  if (int rc = fill()) { bppp_nl_destroy(nl); return rc; }
  *out = nl;
  return BPPP_OK;
}

int bppp_nl_lengths(const bppp_nl *nl, size_t *nlen, size_t *llen) {
  if (!nl || !nlen || !llen) return BPPP_ERR_ARG;
  *nlen = nl->n; *llen = nl->l;
  return BPPP_OK;
}

int bppp_nl_round_commit(bppp_nl *nl, uint64_t sX[4], uint64_t X_xy[8], uint64_t sR[4], uint64_t R_xy[8]) {
  if (!nl || !sX || !X_xy || !sR || !R_xy) return BPPP_ERR_ARG;
  bppp_ctx *ctx = nl->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = R_();
  const size_t ne = ev(nl->n), le = ev(nl->l), T = ne + le + 1;
  const int c = nl->cur;
  hipStream_t st = ctx->stream;
  uint32_t *scX = nl->sc, *scR = nl->sc + T * 8;
  // scalar sums (NormArgument.hs:113-118): sX' = sum q^4j xL xR, sR' = sum q^4j xR^2
  U256 q2 = mmul(nl->q, nl->q, M), q3 = mmul(q2, nl->q, M), q4 = mmul(q2, q2, M), n2 = mmul(nl->nn, nl->nn, M);
  uint64_t q4w[4], a[4], b[4];
  q4.store(q4w);
  U256 sXn = U256::zero(), sRn = U256::zero(), sXl = U256::zero(), sRl = U256::zero();
  if (nl->n) {
    int rc = norm_round_sums_run(ctx, nl->x[c], nl->n, q4w, a, b); if (rc) return rc;
    sXn = mmul(mmul(madd(n2, n2, M), q3, M), U256::load(a), M);      // 2 n^2 q^3 sX'
    sRn = mmul(mmul(n2, q4, M), U256::load(b), M);                   // n^2 q^4 sR'
    k_nl_round_scalars<<<dim3((unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, st>>>(nl->x[c], (uint32_t)nl->n, 1, frk(nl->q), frk(nl->qinv), scX, scR);
  }
  if (nl->l) {
    int rc = lin_round_sums_run(ctx, nl->lc[c], nl->lx[c], nl->l, a, b); if (rc) return rc;
    sXl = U256::load(a); sRl = U256::load(b);                        // no n, no q (NormArgument.hs:56-59)
    k_nl_round_scalars<<<dim3((unsigned)((le / 2 + 255) / 256)), dim3(256), 0, st>>>(nl->lx[c], (uint32_t)nl->l, 0, frk(U256::zero()), frk(U256::zero()),
                                                                                     scX + ne * 8, scR + ne * 8);
  }
  nl->sX = madd(sXn, sXl, M); nl->sR = madd(sRn, sRl, M);            // BPCompose.makeScalarsComs (Bulletproof.hs:258-261)
  nl->sX.store(sX); nl->sR.store(sR);
  NL_HIP(nl, hipMemcpyAsync(scX + (ne + le) * 8, sX, 32, hipMemcpyHostToDevice, st));
  NL_HIP(nl, hipMemcpyAsync(scR + (ne + le) * 8, sR, 32, hipMemcpyHostToDevice, st));
  // ac = commit (updatePSV com sX X-opening); bc = commit (updatePSV com sR R-opening)   (Bulletproof.hs:349-350)
  uint64_t outs[16];
  int rc = msm_run(ctx, nl->sc, nl->P[c], T, 2, 1, 0, outs);
  if (rc) return rc;
  memcpy(X_xy, outs, 64); memcpy(R_xy, outs + 8, 64);
  return BPPP_OK;
}

int bppp_nl_round_collapse(bppp_nl *nl, const uint64_t e_[4]) {
  if (!nl || !e_) return BPPP_ERR_ARG;
  bppp_ctx *ctx = nl->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = R_();
  const U256 e = U256::load(e_);
  if (cmp(e, M.m) >= 0) return fail(ctx, BPPP_ERR_ARG, "nl_round_collapse: challenge not canonical");
  const int c = nl->cur, d = 1 - c;
  const size_t ne = ev(nl->n), le = ev(nl->l);
  const size_t n2 = (nl->n + 1) / 2, l2 = (nl->l + 1) / 2, ne2 = ev(n2), le2 = ev(l2);
  hipStream_t st = ctx->stream;
  // makeEs e = (e, e^2 - 1); sc' = s + e0 sX + e1 sR   (NormArgument.hs:109; Bulletproof.hs:352-353)
  U256 e1 = msub(mmul(e, e, M), U256::one(), M);
  nl->s = madd(nl->s, madd(mmul(e, nl->sX, M), mmul(e1, nl->sR, M), M), M);
  NL_HIP(nl, hipMemsetAsync(nl->P[d], 0, nl->cap * 64, st));
  uint64_t u[4], v[4];
  // the basis folds of both sub-arguments go out as ONE launch (their 130-row dependency chains then run side by side)
  std::pair<SInt, SInt> abn, abl;
  const uint64_t *bm[2], *am[2]; int bn[2], an[2]; const void *src[2]; void *dst[2]; size_t cnt[2]; int nseg = 0;
  if (nl->n) {   // Norm.collapse (NormArgument.hs:123-129)
    abn = rational_reduce_scalar(mmul(e, nl->qinv, M));
    U256 b0 = extract_scalar(abn.second), b0i = minv(b0, M);
    b0i.store(u); mmul(mmul(e, nl->q, M), b0i, M).store(v);
    int rc = fold_scalars_launch(ctx, u, v, nl->x[c], nl->n, nl->x[d]); if (rc) return rc;
    bm[nseg] = abn.second.m; bn[nseg] = abn.second.neg; am[nseg] = abn.first.m; an[nseg] = abn.first.neg;
    src[nseg] = nl->P[c]; dst[nseg] = nl->P[d]; cnt[nseg] = nl->n; nseg++;
    nl->nn = mmul(mmul(nl->nn, b0, M), nl->qinv, M);
    nl->q = mmul(nl->q, nl->q, M); nl->qinv = mmul(nl->qinv, nl->qinv, M);
  }
  if (nl->l) {   // Linear.collapse (NormArgument.hs:64-71)
    abl = rational_reduce_scalar(e);
    U256 a0 = extract_scalar(abl.first), b0 = extract_scalar(abl.second), b0i = minv(b0, M);
    b0.store(u); a0.store(v);
    int rc = fold_scalars_launch(ctx, u, v, nl->lc[c], nl->l, nl->lc[d]); if (rc) return rc;
    b0i.store(u); mmul(e, b0i, M).store(v);
    rc = fold_scalars_launch(ctx, u, v, nl->lx[c], nl->l, nl->lx[d]); if (rc) return rc;
    bm[nseg] = abl.second.m; bn[nseg] = abl.second.neg; am[nseg] = abl.first.m; an[nseg] = abl.first.neg;
    src[nseg] = nl->P[c] + ne * 16; dst[nseg] = nl->P[d] + ne2 * 16; cnt[nseg] = nl->l; nseg++;
    nl->ln = mmul(nl->ln, b0, M);
  }
  { int rc = fold_points_multi_run(ctx, nseg, bm, bn, am, an, src, cnt, dst); if (rc) return rc; }
  NL_HIP(nl, hipMemcpyAsync(nl->P[d] + (ne2 + le2) * 16, nl->P[c] + (ne + le) * 16, 64, hipMemcpyDeviceToDevice, st));
  NL_HIP(nl, hipStreamSynchronize(st));
  nl->n = nl->n ? n2 : 0; nl->l = nl->l ? l2 : 0; nl->cur = d;
  return BPPP_OK;
}

// current state for parity checks: raw vectors (NOT multiplied by the normalisation), basis, and the host scalars
int bppp_nl_download(bppp_nl *nl, uint64_t *norm_x, uint64_t *norm_g_xy, uint64_t *lin_c, uint64_t *lin_x, uint64_t *lin_h_xy,
                     uint64_t s[4], uint64_t q[4], uint64_t norm_nrmlz[4], uint64_t lin_nrmlz[4]) {
  if (!nl) return BPPP_ERR_ARG;
  bppp_ctx *ctx = nl->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const int c = nl->cur;
  hipStream_t st = ctx->stream;
  if (nl->n && norm_x) NL_HIP(nl, hipMemcpyAsync(norm_x, nl->x[c], nl->n * 32, hipMemcpyDeviceToHost, st));
  if (nl->n && norm_g_xy) NL_HIP(nl, hipMemcpyAsync(norm_g_xy, nl->P[c], nl->n * 64, hipMemcpyDeviceToHost, st));
  if (nl->l && lin_c) NL_HIP(nl, hipMemcpyAsync(lin_c, nl->lc[c], nl->l * 32, hipMemcpyDeviceToHost, st));
  if (nl->l && lin_x) NL_HIP(nl, hipMemcpyAsync(lin_x, nl->lx[c], nl->l * 32, hipMemcpyDeviceToHost, st));
  if (nl->l && lin_h_xy) NL_HIP(nl, hipMemcpyAsync(lin_h_xy, nl->P[c] + ev(nl->n) * 16, nl->l * 64, hipMemcpyDeviceToHost, st));
  NL_HIP(nl, hipStreamSynchronize(st));
  if (s) nl->s.store(s);
  if (q) nl->q.store(q);
  if (norm_nrmlz) nl->nn.store(norm_nrmlz);
  if (lin_nrmlz) nl->ln.store(lin_nrmlz);
  return BPPP_OK;
}

// getWitness (NormArgument.hs:121, :62 composed by Bulletproof.hs:264): nrmlz * x, times scalarComp (= 1)
int bppp_nl_get_witness(bppp_nl *nl, uint64_t *norm_w, uint64_t *lin_w) {
  if (!nl || (nl->n && !norm_w) || (nl->l && !lin_w)) return BPPP_ERR_ARG;
  int rc = bppp_nl_download(nl, norm_w, nullptr, nullptr, lin_w, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const Mod &M = R_();
  for (size_t i = 0; i < nl->n; i++) mmul(mmul(U256::load(norm_w + 4 * i), nl->nn, M), nl->scomp, M).store(norm_w + 4 * i);
  for (size_t i = 0; i < nl->l; i++) mmul(mmul(U256::load(lin_w + 4 * i), nl->ln, M), nl->scomp, M).store(lin_w + 4 * i);
  return BPPP_OK;
}

// verifyBPM (Bulletproof.hs:370-378) given the challenges (last round first, as the reference holds them):
// out = commit( wit' ++ initCom ++ [e0 X, e1 R] ), the caller checks it is infinity.
int bppp_nl_verify(bppp_ctx *ctx, const uint64_t q_[4], const uint64_t sp_[4], const uint64_t g_xy[8], const uint64_t *pub_norm,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *pub_lin_c, const uint64_t *pub_lin_x, const uint64_t *lin_h_xy,
                   size_t llen, const uint64_t *es, size_t k, const uint64_t *wit_norm, size_t fn, const uint64_t *wit_lin, size_t fl,
                   const uint64_t *init_scalars, const uint64_t *init_points_xy, size_t ninit, const uint64_t *responses_xy, uint64_t out_xy[8]) {
  if (!ctx || !q_ || !sp_ || !g_xy || !out_xy) return BPPP_ERR_ARG;
  if ((nlen && (!pub_norm || !norm_g_xy)) || (llen && (!pub_lin_c || !pub_lin_x || !lin_h_xy)) || (k && (!es || !responses_xy)) ||
      (fn && !wit_norm) || (fl && !wit_lin) || (ninit && (!init_scalars || !init_points_xy)) || k > 30)
    return fail(ctx, BPPP_ERR_ARG, "nl_verify: bad arguments");
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  // the proof-supplied data is untrusted: scalars must be canonical, points on the curve (or the infinity encoding)
  if (!scalars_canonical(q_, 1) || !scalars_canonical(sp_, 1) || !scalars_canonical(pub_norm, nlen) || !scalars_canonical(pub_lin_c, llen) ||
      !scalars_canonical(pub_lin_x, llen) || !scalars_canonical(es, k) || !scalars_canonical(wit_norm, fn) || !scalars_canonical(wit_lin, fl) ||
      !scalars_canonical(init_scalars, ninit))
    return fail(ctx, BPPP_ERR_ARG, "nl_verify: a scalar is not canonical (>= n)");
  if (!points_on_curve(g_xy, 1) || !points_on_curve(norm_g_xy, nlen) || !points_on_curve(lin_h_xy, llen) || !points_on_curve(init_points_xy, ninit) ||
      !points_on_curve(responses_xy, 2 * k))
    return fail(ctx, BPPP_ERR_POINT, "nl_verify: a point is not on the curve");
  hipSetDevice(ctx->device);
  const Mod &M = R_();
  hipStream_t st = ctx->stream;
  const size_t T = nlen + llen + 1 + ninit + 2 * k;
  const size_t tn = fn << k, tl = fl << k;
  // scratch: [scalars T][tensor norm tn][tensor lin tl][pub tmp max(nlen,llen)] as Fr, then points T
  size_t words = (T + tn + tl + (nlen > llen ? nlen : llen) + 4) * 8 + T * 16;
  { int rc0 = ensure_scratch(ctx, words * 4); if (rc0) return rc0; }
  uint32_t *buf = (uint32_t *)ctx->ws2;
  uint32_t *d_sc = buf, *d_tn = d_sc + T * 8, *d_tl = d_tn + tn * 8, *d_pub = d_tl + tl * 8,
           *d_pts = d_pub + (nlen > llen ? nlen : llen) * 8 + 32;
  int rc = BPPP_OK;
  std::vector<uint64_t> qs(4 * (k ? k : 1)), ones(4 * (k ? k : 1), 0), tl_host(4 * (tl ? tl : 1));
  U256 q = U256::load(q_), qp = q;
  for (size_t r = 0; r < k; r++) { qp.store(&qs[4 * r]); qp = mmul(qp, qp, M); ones[4 * r] = 1; }   // iterate (^2) q; qp ends as qF
  do {
    // Norm.expandChallenges (NormArgument.hs:131-145)
    if (fn) { rc = tensor_run(ctx, wit_norm, fn, es, qs.data(), k, d_tn); if (rc) break; }
    if (nlen) {
      if (hipMemcpyAsync(d_pub, pub_norm, nlen * 32, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: upload"); break; }
      k_sub_scalars<<<dim3((unsigned)((nlen + 255) / 256)), dim3(256), 0, st>>>(d_pub, d_tn, (uint32_t)nlen, (uint32_t)(tn < nlen ? tn : nlen), d_sc);
      if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: norm scalars"); break; }
    }
    // Linear.expandChallenges (NormArgument.hs:73-81)
    if (fl) { rc = tensor_run(ctx, wit_lin, fl, es, ones.data(), k, d_tl); if (rc) break; }
    if (llen) {
      if (hipMemcpyAsync(d_pub, pub_lin_x, llen * 32, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: upload"); break; }
      k_sub_scalars<<<dim3((unsigned)((llen + 255) / 256)), dim3(256), 0, st>>>(d_pub, d_tl, (uint32_t)llen, (uint32_t)(tl < llen ? tl : llen), d_sc + nlen * 8);
      if (tl && hipMemcpyAsync(tl_host.data(), d_tl, tl * 32, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: download"); break; }
      if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: lin scalars"); break; }
    }
    // sc = weightedDotZip (powers' qF^2) vs vs  +  dotZip (contract' expEs cs) vs      (NormArgument.hs:135, :76-78)
    U256 sc = U256::zero(), qF2 = mmul(qp, qp, M), w = qF2;
    for (size_t i = 0; i < fn; i++) { U256 v = U256::load(wit_norm + 4 * i); sc = madd(sc, mmul(w, mmul(v, v, M), M), M); w = mmul(w, qF2, M); }
    for (size_t j = 0; j < llen && j < tl; j++) sc = madd(sc, mmul(U256::load(pub_lin_c + 4 * j), U256::load(&tl_host[4 * j]), M), M);
    // wit' = updatePSV basis (sp - sc) chs; then initCom; then e0 X, e1 R per response   (Bulletproof.hs:376, :365-368)
    std::vector<uint64_t> tail(4 * (1 + ninit + 2 * k));
    msub(U256::load(sp_), sc, M).store(&tail[0]);
    if (ninit) memcpy(&tail[4], init_scalars, ninit * 32);
    for (size_t r = 0; r < k; r++) {
      U256 e = U256::load(es + 4 * r);
      e.store(&tail[4 * (1 + ninit + 2 * r)]);
      msub(mmul(e, e, M), U256::one(), M).store(&tail[4 * (1 + ninit + 2 * r + 1)]);
    }
    hipError_t he = hipMemcpyAsync(d_sc + (nlen + llen) * 8, tail.data(), tail.size() * 8, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && nlen) he = hipMemcpyAsync(d_pts, norm_g_xy, nlen * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && llen) he = hipMemcpyAsync(d_pts + nlen * 16, lin_h_xy, llen * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_pts + (nlen + llen) * 16, g_xy, 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && ninit) he = hipMemcpyAsync(d_pts + (nlen + llen + 1) * 16, init_points_xy, ninit * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && k) he = hipMemcpyAsync(d_pts + (nlen + llen + 1 + ninit) * 16, responses_xy, 2 * k * 64, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify: upload of the MSM tail failed"); break; }
    rc = msm_run(ctx, d_sc, d_pts, T, 1, 1, 0, out_xy);
  } while (0);
  hipStreamSynchronize(st);
  return rc;
}

// proveBPM (src/Bulletproof.hs:357-359) entirely behind the ABI: n_rounds x proveRoundM (:346-355) with the caller's
// oracle.  The oracle sees what ZKPT's `oracle` sees (src/ZKP.hs:96-101): the WHOLE transcript, newest commitments first,
// as affine points; it returns the first challenge (`head <$> oracle [ac, bc]`).  `transcript_xy` / `*ntranscript` carry the
// commitments made before the argument starts (e.g. by the range proof) and receive the responses.  responses_xy gets the
// n_rounds (X, R) pairs LAST ROUND FIRST (:359); es likewise.
int bppp_nl_prove(bppp_nl *nl, size_t n_rounds, bppp_oracle_fn oracle, void *user, uint64_t *transcript_xy, size_t *ntranscript,
                  size_t transcript_cap, uint64_t *responses_xy, uint64_t *es) {
  if (!nl || !oracle || !responses_xy || !es || !transcript_xy || !ntranscript) return BPPP_ERR_ARG;
  if (*ntranscript + 2 * n_rounds > transcript_cap) return bppp::fail(nl->ctx, BPPP_ERR_ARG, "nl_prove: transcript buffer too small");
  for (size_t r = 0; r < n_rounds; r++) {
    uint64_t sX[4], sR[4], X[8], R[8], e[4];
    int rc = bppp_nl_round_commit(nl, sX, X, sR, R); if (rc) return rc;
    // cs' = xs ++ cs (ZKP.hs:98): prepend [ac, bc]
    memmove(transcript_xy + 16, transcript_xy, *ntranscript * 64);
    memcpy(transcript_xy, X, 64); memcpy(transcript_xy + 8, R, 64);
    *ntranscript += 2;
    oracle(user, transcript_xy, *ntranscript, e);
    if (cmp(U256::load(e), R_().m) >= 0) return bppp::fail(nl->ctx, BPPP_ERR_ARG, "nl_prove: oracle returned a non-canonical scalar");
    rc = bppp_nl_round_collapse(nl, e); if (rc) return rc;
    size_t slot = n_rounds - 1 - r;                       // fmap (: resps): the newest response goes to the front
    memcpy(responses_xy + 16 * slot, X, 64); memcpy(responses_xy + 16 * slot + 8, R, 64);
    memcpy(es + 4 * slot, e, 32);
  }
  return BPPP_OK;
}

// the challenge derivation of verifyBPM (src/Bulletproof.hs:374): foldrM walks the responses from the right (first round
// first) and conses, so es comes out ordered like the responses (last round first).
int bppp_nl_verify_challenges(bppp_oracle_fn oracle, void *user, const uint64_t *responses_xy, size_t k, uint64_t *transcript_xy,
                              size_t *ntranscript, size_t transcript_cap, uint64_t *es) {
  if (!oracle || (k && (!responses_xy || !es)) || !transcript_xy || !ntranscript) return BPPP_ERR_ARG;
  if (*ntranscript + 2 * k > transcript_cap) return BPPP_ERR_ARG;
  for (size_t i = k; i-- > 0;) {
    memmove(transcript_xy + 16, transcript_xy, *ntranscript * 64);
    memcpy(transcript_xy, responses_xy + 16 * i, 128);
    *ntranscript += 2;
    oracle(user, transcript_xy, *ntranscript, es + 4 * i);
  }
  return BPPP_OK;
}

}  // extern "C"
