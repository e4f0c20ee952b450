// ip.hip — the inner-product flavour of the norm-linear argument with device-resident vectors.
//
// Device counterpart of src/Bulletproof/InnerProductArgument.hs (the CLI's DEFAULT argument, app/Parse.hs:100;
// used by examples/32bit, 64bit, rec_test):
//   makeNorm r ss gs   (:194-206)  pairs (s0,g0),(s1,g1) -> IPF x' g' y' h', g' = g1 + r g0, h' = g1 - r g0, q = r^4
//   InnerProduct       (:43-127)   makeEs e = (1/e, e); makeScalarsComs via foldLR (:17-26, :70-81); collapse (:86-101);
//                                  getWitness (Norm: :222-223); expandChallenges (:103-124)
//   Linear             (:132-181)  half-length L/R openings (:155-158); collapse with rationalReduce(1/e) (:162-170)
//   NormLinear         (:239-267)  BPCompose (Bulletproof.hs:225-273)
// Same round driver contract as nl.hip: commit the two response points on the device, hash on the host, collapse on the
// device.  Points: [G' (m) | H' (m) | H_lin (l) | g], each slice padded to even length with infinity.
#include <string.h>
#include <vector>
#include "ctx.hpp"
#include "ec.hip.h"
#include "hostmath.hpp"

namespace bppp {
int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);
int fold_points_run(bppp_ctx *, const uint64_t *, int, const uint64_t *, int, const void *, size_t, void *);
int fold_scalars_launch(bppp_ctx *, const uint64_t *, const uint64_t *, const void *, size_t, void *);
int fold_points_multi_run(bppp_ctx *, int, const uint64_t *const[], const int[], const uint64_t *const[], const int[], const void *const[], const size_t[], void *const[]);
int tensor_run(bppp_ctx *, const uint64_t *, size_t, const uint64_t *, const uint64_t *, size_t, void *);

struct FrA { uint32_t v[8]; };
static FrA fra(const bppp_host::U256 &x) {
  FrA r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)x.w[i]; r.v[2 * i + 1] = (uint32_t)(x.w[i] >> 32); }
  return r;
}
BPPP_DI fe fe_ofa(const FrA &a) { fe r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i]; return r; }
BPPP_DI fe fr_powu(fe base, uint32_t e) {
  fe acc = fe_one();
  while (e) { if (e & 1u) acc = fe_mul<1>(acc, base); base = fe_sqr<1>(base); e >>= 1; }
  return acc;
}
BPPP_DI void block_sum2_ip(fe &a, fe &b, uint32_t *lds) {
  const int t = threadIdx.x;
  for (int i = 0; i < 8; i++) { lds[t * 16 + i] = a.v[i]; lds[t * 16 + 8 + i] = b.v[i]; }
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (t < d) {
      fe x, y, u, w;
      for (int i = 0; i < 8; i++) { x.v[i] = lds[t * 16 + i]; y.v[i] = lds[t * 16 + 8 + i]; u.v[i] = lds[(t + d) * 16 + i]; w.v[i] = lds[(t + d) * 16 + 8 + i]; }
      x = fe_add<1>(x, u); y = fe_add<1>(y, w);
      for (int i = 0; i < 8; i++) { lds[t * 16 + i] = x.v[i]; lds[t * 16 + 8 + i] = y.v[i]; }
    }
    __syncthreads();
  }
  if (t == 0) for (int i = 0; i < 8; i++) { a.v[i] = lds[i]; b.v[i] = lds[8 + i]; }
}

// IP makeScalarsComs (:70-81): l = sum_j q^(2j) xL yR, r = sum_j q^(2j) xR yL, and the L / R opening scalars laid over
// the even-padded basis slices: on G: [2j] <- R: q xR, [2j+1] <- L: qinv xL; on H': [2j] <- L: yR, [2j+1] <- R: yL.
__global__ void __launch_bounds__(256) k_ip_round(const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, uint32_t m, FrA q_, FrA qinv_,
                                                  FrA q2_, FrA q2_256_, uint32_t *__restrict__ lg, uint32_t *__restrict__ rg,
                                                  uint32_t *__restrict__ lh, uint32_t *__restrict__ rh, uint32_t *__restrict__ sums) {
  __shared__ uint32_t lds[256 * 16];
  const uint32_t np = (m + 1) / 2;
  fe q = fe_ofa(q_), qinv = fe_ofa(qinv_), step = fe_ofa(q2_256_);
  fe w = fr_powu(fe_ofa(q2_), threadIdx.x);
  fe sl = fe_zero(), sr = fe_zero();
  for (uint32_t j = threadIdx.x; j < np; j += 256) {
    bool has = 2 * j + 1 < m;
    fe xl = fe_load(x + (size_t)(2 * j) * 8), yl = fe_load(y + (size_t)(2 * j) * 8);
    fe xr = has ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
    fe yr = has ? fe_load(y + (size_t)(2 * j + 1) * 8) : fe_zero();
    sl = fe_add<1>(sl, fe_mul<1>(w, fe_mul<1>(xl, yr)));
    sr = fe_add<1>(sr, fe_mul<1>(w, fe_mul<1>(xr, yl)));
    fe_store(lg + (size_t)(2 * j) * 8, fe_zero());
    fe_store(lg + (size_t)(2 * j + 1) * 8, fe_mul<1>(qinv, xl));
    fe_store(rg + (size_t)(2 * j) * 8, fe_mul<1>(q, xr));
    fe_store(rg + (size_t)(2 * j + 1) * 8, fe_zero());
    fe_store(lh + (size_t)(2 * j) * 8, yr);
    fe_store(lh + (size_t)(2 * j + 1) * 8, fe_zero());
    fe_store(rh + (size_t)(2 * j) * 8, fe_zero());
    fe_store(rh + (size_t)(2 * j + 1) * 8, yl);
    w = fe_mul<1>(w, step);
  }
  block_sum2_ip(sl, sr, lds);
  if (threadIdx.x == 0) { fe_store(sums, sl); fe_store(sums + 8, sr); }
}
// Linear (IP flavour) makeScalarsComs (:155-158): l = sum cR xL, r = sum cL xR; L opening: xL on H_R, R opening: xR on H_L
__global__ void __launch_bounds__(256) k_iplin_round(const uint32_t *__restrict__ c, const uint32_t *__restrict__ x, uint32_t n,
                                                     uint32_t *__restrict__ ls, uint32_t *__restrict__ rs, uint32_t *__restrict__ sums) {
  __shared__ uint32_t lds[256 * 16];
  const uint32_t np = (n + 1) / 2;
  fe sl = fe_zero(), sr = fe_zero();
  for (uint32_t j = threadIdx.x; j < np; j += 256) {
    bool has = 2 * j + 1 < n;
    fe cl = fe_load(c + (size_t)(2 * j) * 8), xl = fe_load(x + (size_t)(2 * j) * 8);
    fe cr = has ? fe_load(c + (size_t)(2 * j + 1) * 8) : fe_zero();
    fe xr = has ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
    sl = fe_add<1>(sl, fe_mul<1>(cr, xl));
    sr = fe_add<1>(sr, fe_mul<1>(cl, xr));
    fe_store(ls + (size_t)(2 * j) * 8, fe_zero());
    fe_store(ls + (size_t)(2 * j + 1) * 8, xl);
    fe_store(rs + (size_t)(2 * j) * 8, xr);
    fe_store(rs + (size_t)(2 * j + 1) * 8, fe_zero());
  }
  block_sum2_ip(sl, sr, lds);
  if (threadIdx.x == 0) { fe_store(sums, sl); fe_store(sums + 8, sr); }
}
// makeNorm's basis change (:200-206): p = commit (CP r g0) = r*g0 (256 rows, wave-uniform schedule); g' = g1 + p; h' = g1 - p
struct BasisK { uint32_t r[8]; };
__global__ void __launch_bounds__(64) k_ip_basis(const uint32_t *__restrict__ gs, uint32_t n, BasisK K, uint32_t *__restrict__ gout, uint32_t *__restrict__ hout) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m = (n + 1) / 2;
  if (j >= m) return;
  aff g0 = aff_load(gs + (size_t)(2 * j) * 16);
  aff g1 = (2 * j + 1 < n) ? aff_load(gs + (size_t)(2 * j + 1) * 16) : aff_inf();
  xyzz p = xyzz_inf();
  for (int row = 255; row >= 0; row--) {
    p = xyzz_dbl(p);
    if ((K.r[row >> 5] >> (row & 31)) & 1u) xyzz_madd(p, g0);
  }
  xyzz a = p; xyzz_madd(a, g1);
  // -p: negate Y (magnitude 3 -> 4, accepted by xyzz_madd's subtraction bound? keep within 3 by normalising)
  xyzz b = p;
  if (!xyzz_is_inf(b)) b.Y = fq_normalize(fq_neg<3>(b.Y));
  xyzz_madd(b, g1);
  aff_store(gout + (size_t)j * 16, xyzz_to_aff(a));
  aff_store(hout + (size_t)j * 16, xyzz_to_aff(b));
}
// out[i] = pub[i] - (i < nt ? t[i] : 0)
__global__ void __launch_bounds__(256) k_ip_sub(const uint32_t *__restrict__ pub, const uint32_t *__restrict__ t, uint32_t n, uint32_t nt, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe p = fe_load(pub + (size_t)i * 8);
  if (i < nt) p = fe_sub<1>(p, fe_load(t + (size_t)i * 8));
  fe_store(out + (size_t)i * 8, p);
}
}  // namespace bppp

using namespace bppp;
using namespace bppp_host;

struct bppp_ip {
  bppp_ctx *ctx;
  size_t m, l, cap;                       // IP length (pairs of norm elements), linear length, buffer capacity
  uint32_t *x[2], *y[2], *lx[2], *lc[2];
  uint32_t *P[2];
  uint32_t *sc;                           // [2][cap] scalars of the L / R MSMs
  uint32_t *sums;
  int cur;
  U256 s, nx, ny, q, qinv, ln, psv, scomp;
  U256 sL, sR;
};
static size_t ev2(size_t v) { return v + (v & 1); }
static const Mod &RM() { return FR(); }
#define IP_HIP(ip, call)                                                                                   \
  do {                                                                                                     \
    hipError_t _e = (call);                                                                                \
    if (_e != hipSuccess) return bppp::fail((ip)->ctx, BPPP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

static int ip_transform_basis(bppp_ctx *ctx, const U256 &r, const uint32_t *d_gs, size_t nlen, uint32_t *d_g, uint32_t *d_h) {
  if (!nlen) return BPPP_OK;
  BasisK K;
  for (int i = 0; i < 4; i++) { K.r[2 * i] = (uint32_t)r.w[i]; K.r[2 * i + 1] = (uint32_t)(r.w[i] >> 32); }
  uint32_t m = (uint32_t)((nlen + 1) / 2);
  k_ip_basis<<<dim3((m + 63) / 64), dim3(64), 0, ctx->stream>>>(d_gs, (uint32_t)nlen, K, d_g, d_h);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

extern "C" {

void bppp_ip_destroy(bppp_ip *ip) {
  if (!ip) return;
  bppp_ctx *ctx = ip->ctx;                 // kept alive by this handle's reference even after bppp_ctx_destroy
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 2; k++) { hipFree(ip->x[k]); hipFree(ip->y[k]); hipFree(ip->lx[k]); hipFree(ip->lc[k]); hipFree(ip->P[k]); }
  hipFree(ip->sc); hipFree(ip->sums);
  delete ip;
  ctx_release(ctx);
}

// makeNormLinearBP' 1 r cs nss ngs lss lgs (InnerProductArgument.hs:248) inside makePSV s g
int bppp_ip_create(bppp_ctx *ctx, const uint64_t s[4], const uint64_t g_xy[8], const uint64_t r_[4], const uint64_t *norm_s,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy,
                   size_t llen, bppp_ip **out) {
  if (!ctx || !out || !s || !g_xy || !r_ || ctx_closed(ctx)) return BPPP_ERR_ARG;
  if ((nlen && (!norm_s || !norm_g_xy)) || (llen && (!lin_c || !lin_x || !lin_h_xy))) return fail(ctx, BPPP_ERR_ARG, "ip_create: null vector");
  if (nlen + llen == 0 || nlen >= (1u << 30) || llen >= (1u << 30)) return fail(ctx, BPPP_ERR_ARG, "ip_create: bad lengths");
  hipSetDevice(ctx->device);
  const Mod &M = RM();
  bppp_ip *ip = new bppp_ip();
  memset(ip, 0, sizeof *ip);
  const size_t m = (nlen + 1) / 2;
  ip->ctx = ctx; ctx_retain(ctx); ip->m = m; ip->l = llen; ip->cap = 2 * ev2(m) + ev2(llen) + 1; ip->cur = 0;
  U256 r = U256::load(r_), r2 = mmul(r, r, M);
  ip->q = mmul(r2, r2, M); ip->qinv = minv(ip->q, M);
  ip->s = U256::from_u64(4); ip->nx = U256::one(); ip->ny = U256::one(); ip->ln = U256::one();
  ip->psv = U256::load(s); ip->scomp = U256::one();
  hipStream_t st = ctx->stream;
  bool bad = false;
  for (int k = 0; k < 2; k++) {
    bad |= hipMalloc(&ip->x[k], (ev2(m) + 2) * 32) != hipSuccess || hipMalloc(&ip->y[k], (ev2(m) + 2) * 32) != hipSuccess;
    bad |= hipMalloc(&ip->lx[k], (ev2(llen) + 2) * 32) != hipSuccess || hipMalloc(&ip->lc[k], (ev2(llen) + 2) * 32) != hipSuccess;
    bad |= hipMalloc(&ip->P[k], ip->cap * 64) != hipSuccess;
  }
  bad |= hipMalloc(&ip->sc, 2 * ip->cap * 32) != hipSuccess || hipMalloc(&ip->sums, 256) != hipSuccess;
  uint32_t *tmp = nullptr;
  if (!bad && nlen) bad |= hipMalloc(&tmp, ev2(nlen) * 64) != hipSuccess;
  if (bad) { if (tmp) hipFree(tmp); bppp_ip_destroy(ip); return fail(ctx, BPPP_ERR_HIP, "ip_create: hipMalloc failed"); }
  auto fill = [&]() -> int {                 // any failure below goes through ONE cleanup (tmp freed, handle destroyed)
  IP_HIP(ip, hipMemsetAsync(ip->P[0], 0, ip->cap * 64, st));
  if (nlen) {
    // x' = s0/(2r) + s1/2,  y' = -s0/(2r) + s1/2   (:202-203)
    U256 half = minv(U256::from_u64(2), M), r2i = minv(madd(r, r, M), M);
    std::vector<uint64_t> xs(4 * m), ys(4 * m);
    for (size_t j = 0; j < m; j++) {
      U256 s0 = U256::load(norm_s + 8 * j), s1 = (2 * j + 1 < nlen) ? U256::load(norm_s + 8 * j + 4) : U256::zero();
      U256 a = mmul(r2i, s0, M), b = mmul(half, s1, M);
      madd(a, b, M).store(&xs[4 * j]); msub(b, a, M).store(&ys[4 * j]);
    }
    IP_HIP(ip, hipMemcpyAsync(ip->x[0], xs.data(), m * 32, hipMemcpyHostToDevice, st));
    IP_HIP(ip, hipMemcpyAsync(ip->y[0], ys.data(), m * 32, hipMemcpyHostToDevice, st));
    IP_HIP(ip, hipMemsetAsync(tmp, 0, ev2(nlen) * 64, st));
    IP_HIP(ip, hipMemcpyAsync(tmp, norm_g_xy, nlen * 64, hipMemcpyHostToDevice, st));
    IP_HIP(ip, hipStreamSynchronize(st));
    int rc = ip_transform_basis(ctx, r, tmp, nlen, ip->P[0], ip->P[0] + ev2(m) * 16);
    if (rc) return rc;
  }
  if (llen) {
    IP_HIP(ip, hipMemcpyAsync(ip->lc[0], lin_c, llen * 32, hipMemcpyHostToDevice, st));
    IP_HIP(ip, hipMemcpyAsync(ip->lx[0], lin_x, llen * 32, hipMemcpyHostToDevice, st));
    IP_HIP(ip, hipMemcpyAsync(ip->P[0] + 2 * ev2(m) * 16, lin_h_xy, llen * 64, hipMemcpyHostToDevice, st));
  }
  IP_HIP(ip, hipMemcpyAsync(ip->P[0] + (2 * ev2(m) + ev2(llen)) * 16, g_xy, 64, hipMemcpyHostToDevice, st));
  IP_HIP(ip, hipStreamSynchronize(st));
  return BPPP_OK;
  };
  const int rc_fill = fill();
  if (tmp) { hipStreamSynchronize(st); hipFree(tmp); }
  if (rc_fill) { bppp_ip_destroy(ip); return rc_fill; }
  *out = ip;
  return BPPP_OK;
}

int bppp_ip_lengths(const bppp_ip *ip, size_t *ip_len, size_t *llen) {
  if (!ip || !ip_len || !llen) return BPPP_ERR_ARG;
  *ip_len = ip->m; *llen = ip->l;
  return BPPP_OK;
}

// first half of proveRoundM (Bulletproof.hs:346-350) for the IP flavour
int bppp_ip_round_commit(bppp_ip *ip, uint64_t sL[4], uint64_t L_xy[8], uint64_t sR[4], uint64_t R_xy[8]) {
  if (!ip || !sL || !L_xy || !sR || !R_xy) return BPPP_ERR_ARG;
  bppp_ctx *ctx = ip->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = RM();
  const size_t me = ev2(ip->m), le = ev2(ip->l), T = 2 * me + le + 1;
  const int c = ip->cur;
  hipStream_t st = ctx->stream;
  uint32_t *scL = ip->sc, *scR = ip->sc + T * 8;
  int rc = ensure_pinned(ctx, 256); if (rc) return rc;
  U256 sLn = U256::zero(), sRn = U256::zero(), sLl = U256::zero(), sRl = U256::zero();
  uint64_t *hs = (uint64_t *)ctx->pinned;
  if (ip->m) {
    U256 q2 = mmul(ip->q, ip->q, M), stp = q2;
    for (int i = 0; i < 8; i++) stp = mmul(stp, stp, M);   // (q^2)^256
    k_ip_round<<<dim3(1), dim3(256), 0, st>>>(ip->x[c], ip->y[c], (uint32_t)ip->m, fra(ip->q), fra(ip->qinv), fra(q2), fra(stp), scL, scR, scL + me * 8,
                                              scR + me * 8, ip->sums);
    IP_HIP(ip, hipMemcpyAsync(hs, ip->sums, 64, hipMemcpyDeviceToHost, st));
    IP_HIP(ip, hipStreamSynchronize(st));
    U256 kk = mmul(mmul(ip->s, ip->nx, M), ip->ny, M);
    sLn = mmul(mmul(kk, ip->q, M), U256::load(hs), M);          // s q nx ny l   (:74)
    sRn = mmul(mmul(kk, q2, M), U256::load(hs + 4), M);         // s q^2 nx ny r (:75)
  }
  if (ip->l) {
    k_iplin_round<<<dim3(1), dim3(256), 0, st>>>(ip->lc[c], ip->lx[c], (uint32_t)ip->l, scL + 2 * me * 8, scR + 2 * me * 8, ip->sums);
    IP_HIP(ip, hipMemcpyAsync(hs, ip->sums, 64, hipMemcpyDeviceToHost, st));
    IP_HIP(ip, hipStreamSynchronize(st));
    sLl = U256::load(hs); sRl = U256::load(hs + 4);
  }
  ip->sL = madd(sLn, sLl, M); ip->sR = madd(sRn, sRl, M);
  ip->sL.store(sL); ip->sR.store(sR);
  IP_HIP(ip, hipMemcpyAsync(scL + (2 * me + le) * 8, sL, 32, hipMemcpyHostToDevice, st));
  IP_HIP(ip, hipMemcpyAsync(scR + (2 * me + le) * 8, sR, 32, hipMemcpyHostToDevice, st));
  uint64_t outs[16];
  rc = msm_run(ctx, ip->sc, ip->P[c], T, 2, 1, 0, outs);
  if (rc) return rc;
  memcpy(L_xy, outs, 64); memcpy(R_xy, outs + 8, 64);
  return BPPP_OK;
}

// second half of proveRoundM: makeEs e = (1/e, e) (:68); s += e0 sL + e1 sR; collapse (:86-101, :162-170)
int bppp_ip_round_collapse(bppp_ip *ip, const uint64_t e_[4]) {
  if (!ip || !e_) return BPPP_ERR_ARG;
  bppp_ctx *ctx = ip->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = RM();
  const U256 e = U256::load(e_);
  if (cmp(e, M.m) >= 0 || e.is_zero()) return fail(ctx, BPPP_ERR_ARG, "ip_round_collapse: challenge must be a non-zero canonical scalar");
  const U256 ei = minv(e, M);
  const int c = ip->cur, d = 1 - c;
  const size_t me = ev2(ip->m), le = ev2(ip->l);
  const size_t m2 = (ip->m + 1) / 2, l2 = (ip->l + 1) / 2, me2 = ev2(m2), le2 = ev2(l2);
  hipStream_t st = ctx->stream;
  ip->psv = madd(ip->psv, madd(mmul(ei, ip->sL, M), mmul(e, ip->sR, M), M), M);
  IP_HIP(ip, hipMemsetAsync(ip->P[d], 0, ip->cap * 64, st));
  uint64_t u[4], v[4];
  int rc;
  std::pair<SInt, SInt> ab, cd, abl;
  const uint64_t *bm[3], *am[3]; int bn[3], an[3]; const void *src[3]; void *dst[3]; size_t cnt[3]; int nseg = 0;
  if (ip->m) {
    ab = rational_reduce_scalar(mmul(ip->qinv, ei, M));
    U256 b0 = extract_scalar(ab.second), b0i = minv(b0, M);
    cd = rational_reduce_scalar(e);
    U256 d0 = extract_scalar(cd.second), d0i = minv(d0, M);
    b0i.store(u); mmul(b0i, mmul(e, ip->q, M), M).store(v);
    rc = fold_scalars_launch(ctx, u, v, ip->x[c], ip->m, ip->x[d]); if (rc) return rc;
    d0i.store(u); mmul(d0i, ei, M).store(v);
    rc = fold_scalars_launch(ctx, u, v, ip->y[c], ip->m, ip->y[d]); if (rc) return rc;
    bm[nseg] = ab.second.m; bn[nseg] = ab.second.neg; am[nseg] = ab.first.m; an[nseg] = ab.first.neg;
    src[nseg] = ip->P[c]; dst[nseg] = ip->P[d]; cnt[nseg] = ip->m; nseg++;
    bm[nseg] = cd.second.m; bn[nseg] = cd.second.neg; am[nseg] = cd.first.m; an[nseg] = cd.first.neg;
    src[nseg] = ip->P[c] + me * 16; dst[nseg] = ip->P[d] + me2 * 16; cnt[nseg] = ip->m; nseg++;
    ip->ny = mmul(ip->ny, d0, M);
    ip->nx = mmul(mmul(ip->nx, b0, M), ip->qinv, M);
    ip->q = mmul(ip->q, ip->q, M); ip->qinv = mmul(ip->qinv, ip->qinv, M);
  }
  if (ip->l) {
    abl = rational_reduce_scalar(ei);
    U256 a0 = extract_scalar(abl.first), b0 = extract_scalar(abl.second), b0i = minv(b0, M);
    b0.store(u); a0.store(v);
    rc = fold_scalars_launch(ctx, u, v, ip->lc[c], ip->l, ip->lc[d]); if (rc) return rc;
    b0i.store(u); mmul(e, b0i, M).store(v);
    rc = fold_scalars_launch(ctx, u, v, ip->lx[c], ip->l, ip->lx[d]); if (rc) return rc;
    bm[nseg] = abl.second.m; bn[nseg] = abl.second.neg; am[nseg] = abl.first.m; an[nseg] = abl.first.neg;
    src[nseg] = ip->P[c] + 2 * me * 16; dst[nseg] = ip->P[d] + 2 * me2 * 16; cnt[nseg] = ip->l; nseg++;
    ip->ln = mmul(ip->ln, b0, M);
  }
  rc = fold_points_multi_run(ctx, nseg, bm, bn, am, an, src, cnt, dst); if (rc) return rc;
  IP_HIP(ip, hipMemcpyAsync(ip->P[d] + (2 * me2 + le2) * 16, ip->P[c] + (2 * me + le) * 16, 64, hipMemcpyDeviceToDevice, st));
  IP_HIP(ip, hipStreamSynchronize(st));
  ip->m = ip->m ? m2 : 0; ip->l = ip->l ? l2 : 0; ip->cur = d;
  return BPPP_OK;
}

// getWitness: Norm (nx x - ny y, nx x + ny y) per element (:222-223), Linear nrmlz * x (:160); also the PSV scalar
int bppp_ip_get_witness(bppp_ip *ip, uint64_t *norm_w /*2*ip_len*/, uint64_t *lin_w, uint64_t s[4]) {
  if (!ip || (ip->m && !norm_w) || (ip->l && !lin_w)) return BPPP_ERR_ARG;
  bppp_ctx *ctx = ip->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  const Mod &M = RM();
  const int c = ip->cur;
  std::vector<uint64_t> xs(4 * (ip->m ? ip->m : 1)), ys(4 * (ip->m ? ip->m : 1));
  if (ip->m) {
    IP_HIP(ip, hipMemcpyAsync(xs.data(), ip->x[c], ip->m * 32, hipMemcpyDeviceToHost, ctx->stream));
    IP_HIP(ip, hipMemcpyAsync(ys.data(), ip->y[c], ip->m * 32, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (ip->l) IP_HIP(ip, hipMemcpyAsync(lin_w, ip->lx[c], ip->l * 32, hipMemcpyDeviceToHost, ctx->stream));
  IP_HIP(ip, hipStreamSynchronize(ctx->stream));
  for (size_t j = 0; j < ip->m; j++) {
    U256 a = mmul(ip->nx, U256::load(&xs[4 * j]), M), b = mmul(ip->ny, U256::load(&ys[4 * j]), M);
    mmul(msub(a, b, M), ip->scomp, M).store(norm_w + 8 * j);
    mmul(madd(a, b, M), ip->scomp, M).store(norm_w + 8 * j + 4);
  }
  for (size_t i = 0; i < ip->l; i++) mmul(mmul(U256::load(lin_w + 4 * i), ip->ln, M), ip->scomp, M).store(lin_w + 4 * i);
  if (s) ip->psv.store(s);
  return BPPP_OK;
}

// verifyBPM (Bulletproof.hs:370-378) for the IP flavour: the basis change of makeNorm, expandChallenges
// (:103-124, :172-181) and the single commit.  wit_norm holds fn scalars as decodeProof' receives them
// (RangeProof.hs:81: makeNormLinearBP 1 ... nrmScs, i.e. makeNorm with r = 1).  es / responses last round first.
int bppp_ip_verify(bppp_ctx *ctx, const uint64_t r_[4], const uint64_t sp_[4], const uint64_t g_xy[8], const uint64_t *pub_norm,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *pub_lin_c, const uint64_t *pub_lin_x, const uint64_t *lin_h_xy,
                   size_t llen, const uint64_t *es, size_t k, const uint64_t *wit_norm, size_t fn, const uint64_t *wit_lin, size_t fl,
                   const uint64_t *init_scalars, const uint64_t *init_points_xy, size_t ninit, const uint64_t *responses_xy, uint64_t out_xy[8]) {
  if (!ctx || !r_ || !sp_ || !g_xy || !out_xy) return BPPP_ERR_ARG;
  if ((nlen && (!pub_norm || !norm_g_xy)) || (llen && (!pub_lin_c || !pub_lin_x || !lin_h_xy)) || (k && (!es || !responses_xy)) ||
      (fn && !wit_norm) || (fl && !wit_lin) || (ninit && (!init_scalars || !init_points_xy)) || k > 30 || (fn & 1))
    return fail(ctx, BPPP_ERR_ARG, "ip_verify: bad arguments");
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  // the proof-supplied data is untrusted: scalars must be canonical, points on the curve (or the infinity encoding)
  if (!scalars_canonical(r_, 1) || !scalars_canonical(sp_, 1) || !scalars_canonical(pub_norm, nlen) || !scalars_canonical(pub_lin_c, llen) ||
      !scalars_canonical(pub_lin_x, llen) || !scalars_canonical(es, k) || !scalars_canonical(wit_norm, fn) || !scalars_canonical(wit_lin, fl) ||
      !scalars_canonical(init_scalars, ninit))
    return fail(ctx, BPPP_ERR_ARG, "ip_verify: a scalar is not canonical (>= n)");
  if (!points_on_curve(g_xy, 1) || !points_on_curve(norm_g_xy, nlen) || !points_on_curve(lin_h_xy, llen) || !points_on_curve(init_points_xy, ninit) ||
      !points_on_curve(responses_xy, 2 * k))
    return fail(ctx, BPPP_ERR_POINT, "ip_verify: a point is not on the curve");
  hipSetDevice(ctx->device);
  const Mod &M = RM();
  hipStream_t st = ctx->stream;
  const size_t m = (nlen + 1) / 2, fm = fn / 2;
  const size_t T = 2 * m + llen + 1 + ninit + 2 * k;
  const size_t tm = fm << k, tl = fl << k;
  size_t mx = m > llen ? m : llen;
  size_t words = (T + 2 * tm + tl + mx + 8) * 8 + (T + ev2(nlen) + 4) * 16;
  { int rc0 = ensure_scratch(ctx, words * 4); if (rc0) return rc0; }
  uint32_t *d_sc = (uint32_t *)ctx->ws2, *d_tx = d_sc + T * 8, *d_ty = d_tx + tm * 8, *d_tl = d_ty + tm * 8, *d_pub = d_tl + tl * 8,
           *d_pts = d_pub + (mx + 8) * 8, *d_raw = d_pts + T * 16;
  int rc = BPPP_OK;
  U256 r = U256::load(r_), r2 = mmul(r, r, M), q = mmul(r2, r2, M), half = minv(U256::from_u64(2), M), r2i = minv(madd(r, r, M), M);
  std::vector<uint64_t> qs(4 * (k ? k : 1)), ones(4 * (k ? k : 1), 0), esx(4 * (k ? k : 1)), tl_host(4 * (tl ? tl : 1));
  U256 qp = q;
  for (size_t i = 0; i < k; i++) { qp.store(&qs[4 * i]); qp = mmul(qp, qp, M); ones[4 * i] = 1; minv(U256::load(es + 4 * i), M).store(&esx[4 * i]); }
  do {
    // public IP vectors: pub_norm (nlen plain norm scalars) goes through makeNorm like every makeNormLinearBP call (:248)
    std::vector<uint64_t> px(4 * (m ? m : 1)), py(4 * (m ? m : 1)), vx(4 * (fm ? fm : 1)), vy(4 * (fm ? fm : 1));
    for (size_t j = 0; j < m; j++) {
      U256 s0 = U256::load(pub_norm + 8 * j), s1 = (2 * j + 1 < nlen) ? U256::load(pub_norm + 8 * j + 4) : U256::zero();
      U256 a = mmul(r2i, s0, M), b = mmul(half, s1, M);
      madd(a, b, M).store(&px[4 * j]); msub(b, a, M).store(&py[4 * j]);
    }
    for (size_t j = 0; j < fm; j++) {     // witness through makeNorm 1: x = (s0 + s1)/2, y = (s1 - s0)/2
      U256 s0 = U256::load(wit_norm + 8 * j), s1 = U256::load(wit_norm + 8 * j + 4);
      mmul(half, madd(s0, s1, M), M).store(&vx[4 * j]); mmul(half, msub(s1, s0, M), M).store(&vy[4 * j]);
    }
    // basis change on the device
    if (nlen) {
      if (hipMemsetAsync(d_raw, 0, ev2(nlen) * 64, st) != hipSuccess || hipMemcpyAsync(d_raw, norm_g_xy, nlen * 64, hipMemcpyHostToDevice, st) != hipSuccess) {
        rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: upload"); break; }
      rc = ip_transform_basis(ctx, r, d_raw, nlen, d_pts, d_pts + m * 16); if (rc) break;
    }
    // tsX = tensor' vsX (1/es) (iterate (^2) q); tsY = tensor' vsY es (repeat 1)    (:118-119)
    if (fm) {
      rc = tensor_run(ctx, vx.data(), fm, esx.data(), qs.data(), k, d_tx); if (rc) break;
      rc = tensor_run(ctx, vy.data(), fm, es, ones.data(), k, d_ty); if (rc) break;
    }
    if (m) {
      hipError_t he = hipMemcpyAsync(d_pub, px.data(), m * 32, hipMemcpyHostToDevice, st);
      if (he != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: upload"); break; }
      k_ip_sub<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st>>>(d_pub, d_tx, (uint32_t)m, (uint32_t)(tm < m ? tm : m), d_sc);
      if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: x scalars"); break; }
      he = hipMemcpyAsync(d_pub, py.data(), m * 32, hipMemcpyHostToDevice, st);
      if (he != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: upload"); break; }
      k_ip_sub<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st>>>(d_pub, d_ty, (uint32_t)m, (uint32_t)(tm < m ? tm : m), d_sc + m * 8);
      if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: y scalars"); break; }
    }
    // Linear: challenges inverted (:174)
    if (fl) { rc = tensor_run(ctx, wit_lin, fl, esx.data(), ones.data(), k, d_tl); if (rc) break; }
    if (llen) {
      if (hipMemcpyAsync(d_pub, pub_lin_x, llen * 32, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: upload"); break; }
      k_ip_sub<<<dim3((unsigned)((llen + 255) / 256)), dim3(256), 0, st>>>(d_pub, d_tl, (uint32_t)llen, (uint32_t)(tl < llen ? tl : llen), d_sc + 2 * m * 8);
      if (tl && hipMemcpyAsync(tl_host.data(), d_tl, tl * 32, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: download"); break; }
      if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: lin scalars"); break; }
    }
    // sc = s * weightedDotZip (powers' qF) vsX vsY + dotZip (contract' expEs cs) vs   (:116, :176-178); s = 4 (makeNorm)
    U256 sc = U256::zero(), w = qp;
    for (size_t j = 0; j < fm; j++) { sc = madd(sc, mmul(w, mmul(U256::load(&vx[4 * j]), U256::load(&vy[4 * j]), M), M), M); w = mmul(w, qp, M); }
    sc = mmul(sc, U256::from_u64(4), M);
    for (size_t j = 0; j < llen && j < tl; j++) sc = madd(sc, mmul(U256::load(pub_lin_c + 4 * j), U256::load(&tl_host[4 * j]), M), M);
    std::vector<uint64_t> tail(4 * (1 + ninit + 2 * k));
    msub(U256::load(sp_), sc, M).store(&tail[0]);
    if (ninit) memcpy(&tail[4], init_scalars, ninit * 32);
    for (size_t i = 0; i < k; i++) {      // makeEs e = (1/e, e)
      memcpy(&tail[4 * (1 + ninit + 2 * i)], &esx[4 * i], 32);
      memcpy(&tail[4 * (1 + ninit + 2 * i + 1)], es + 4 * i, 32);
    }
    hipError_t he = hipMemcpyAsync(d_sc + (2 * m + llen) * 8, tail.data(), tail.size() * 8, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && llen) he = hipMemcpyAsync(d_pts + 2 * m * 16, lin_h_xy, llen * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_pts + (2 * m + llen) * 16, g_xy, 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && ninit) he = hipMemcpyAsync(d_pts + (2 * m + llen + 1) * 16, init_points_xy, ninit * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && k) he = hipMemcpyAsync(d_pts + (2 * m + llen + 1 + ninit) * 16, responses_xy, 2 * k * 64, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify: upload of the MSM tail failed"); break; }
    rc = msm_run(ctx, d_sc, d_pts, T, 1, 1, 0, out_xy);
  } while (0);
  hipStreamSynchronize(st);
  return rc;
}

}  // extern "C"
