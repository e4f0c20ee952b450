// rounds.hip — Fr (scalar-field) vector kernels of the norm / linear halving round, NL flavour.
//
// Replaces the scalar halves of
//   Norm.makeScalarsComs   (src/Bulletproof/NormArgument.hs:113-118, foldXR :20-29)
//   Linear.makeScalarsComs (src/Bulletproof/NormArgument.hs:56-59)
//   Norm/Linear.collapse   (src/Bulletproof/NormArgument.hs:123-129, :64-71) — scalar parts
//   tensor'                (src/Bulletproof.hs:94-95) used by expandChallenges (NormArgument.hs:73-81, :131-145)
// and the fold helpers they are built from (src/Utils.hs:104-111, :209-216).
// Vectors are traversed as ADJACENT pairs (x0,x1),(x2,x3),.. with a zero default for an odd tail
// (src/Bulletproof.hs:77-90).  All values are canonical integers mod n.
#include "ctx.hpp"
#include "fe.hip.h"
#include "modinv.hip.h"
#include "hostmath.hpp"

namespace bppp {

struct Fr4 { uint32_t v[8]; };
static Fr4 to_fr4(const uint64_t x[4]) {
  Fr4 r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)x[i]; r.v[2 * i + 1] = (uint32_t)(x[i] >> 32); }
  return r;
}
BPPP_DI fe fe_from(const Fr4 &a) { fe r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i]; return r; }

BPPP_DI fe fr_pow_u32(fe base, uint32_t e) {
  fe acc = fe_one();
  while (e) { if (e & 1u) acc = fe_mul<1>(acc, base); base = fe_sqr<1>(base); e >>= 1; }
  return acc;
}

// block-wide sum of two Fr values per thread (256 threads), result valid in thread 0
BPPP_DI void block_sum2(fe &a, fe &b, uint32_t *lds) {
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

// sx = sum_j q4^j xL_j xR_j ; sr = sum_j q4^j xR_j^2   (one workgroup)
__global__ void __launch_bounds__(256) k_norm_round_sums(const uint32_t *__restrict__ x, uint32_t n, Fr4 q4_, Fr4 q4_256_, uint32_t *__restrict__ out) {
  __shared__ uint32_t lds[256 * 16];
  const uint32_t np = (n + 1) / 2;
  fe q4 = fe_from(q4_), step = fe_from(q4_256_);
  fe w = fr_pow_u32(q4, threadIdx.x);
  fe sx = fe_zero(), sr = fe_zero();
  for (uint32_t j = threadIdx.x; j < np; j += 256) {
    fe xl = fe_load(x + (size_t)(2 * j) * 8);
    fe xr = (2 * j + 1 < n) ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
    fe wxr = fe_mul<1>(w, xr);
    sx = fe_add<1>(sx, fe_mul<1>(wxr, xl));
    sr = fe_add<1>(sr, fe_mul<1>(wxr, xr));
    w = fe_mul<1>(w, step);
  }
  block_sum2(sx, sr, lds);
  if (threadIdx.x == 0) { fe_store(out, sx); fe_store(out + 8, sr); }
}
// sx = sum cL xR + cR xL ; sr = sum cR xR
__global__ void __launch_bounds__(256) k_lin_round_sums(const uint32_t *__restrict__ c, const uint32_t *__restrict__ x, uint32_t n, uint32_t *__restrict__ out) {
  __shared__ uint32_t lds[256 * 16];
  const uint32_t np = (n + 1) / 2;
  fe sx = fe_zero(), sr = fe_zero();
  for (uint32_t j = threadIdx.x; j < np; j += 256) {
    bool has = 2 * j + 1 < n;
    fe cl = fe_load(c + (size_t)(2 * j) * 8), xl = fe_load(x + (size_t)(2 * j) * 8);
    fe cr = has ? fe_load(c + (size_t)(2 * j + 1) * 8) : fe_zero();
    fe xr = has ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
    sx = fe_add<1>(sx, fe_add<1>(fe_mul<1>(cl, xr), fe_mul<1>(cr, xl)));
    sr = fe_add<1>(sr, fe_mul<1>(cr, xr));
  }
  block_sum2(sx, sr, lds);
  if (threadIdx.x == 0) { fe_store(out, sx); fe_store(out + 8, sr); }
}
// norm: xw[2j] = q xR, xw[2j+1] = qinv xL, rw[j] = xR.   lin (scale = 0): xw[2j] = xR, xw[2j+1] = xL
__global__ void __launch_bounds__(256) k_round_openings(const uint32_t *__restrict__ x, uint32_t n, int scale, Fr4 q_, Fr4 qinv_,
                                                        uint32_t *__restrict__ xw, uint32_t *__restrict__ rw) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (n + 1) / 2) return;
  fe xl = fe_load(x + (size_t)(2 * j) * 8);
  fe xr = (2 * j + 1 < n) ? fe_load(x + (size_t)(2 * j + 1) * 8) : fe_zero();
  fe a = xr, b = xl;
  if (scale) { a = fe_mul<1>(fe_from(q_), xr); b = fe_mul<1>(fe_from(qinv_), xl); }
  fe_store(xw + (size_t)(2 * j) * 8, a);
  fe_store(xw + (size_t)(2 * j + 1) * 8, b);
  fe_store(rw + (size_t)j * 8, xr);
}
// out[j] = u x[2j] + v x[2j+1]
__global__ void __launch_bounds__(256) k_fold_scalars(const uint32_t *__restrict__ x, uint32_t n, Fr4 u_, Fr4 v_, uint32_t *__restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (n + 1) / 2) return;
  fe xl = fe_load(x + (size_t)(2 * j) * 8);
  fe r = fe_mul<1>(fe_from(u_), xl);
  if (2 * j + 1 < n) r = fe_add<1>(r, fe_mul<1>(fe_from(v_), fe_load(x + (size_t)(2 * j + 1) * 8)));
  fe_store(out + (size_t)j * 8, r);
}
// tensor': out[i*2^k + t] = bs[i] * prod_r (bit r of t ? e_r : q_r); fac = [q_0..q_{k-1}, e_0..e_{k-1}]
// with r = 0 the FIRST round (the list version processes it first, so it is the least significant bit)
__global__ void __launch_bounds__(256) k_tensor(const uint32_t *__restrict__ bs, uint32_t nb, const uint32_t *__restrict__ fac, int k,
                                                uint32_t *__restrict__ out) {
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = (uint64_t)nb << k;
  if (idx >= total) return;
  uint32_t t = (uint32_t)(idx & ((1ull << k) - 1)), i = (uint32_t)(idx >> k);
  fe acc = fe_load(bs + (size_t)i * 8);
  for (int r = 0; r < k; r++) {
    const uint32_t *f = fac + (size_t)(((t >> r) & 1u) ? k + r : r) * 8;
    acc = fe_mul<1>(acc, fe_load(f));
  }
  fe_store(out + idx * 8, acc);
}

// batchInverse (src/Data/Field/BatchInverse.hs:14-24): Montgomery's trick, 0 -> 0.  On a SIMD machine one Fermat inversion
// per LANE costs the same wall time as one per wavefront, so the trick pays only along a lane: each lane inverts a run
// of BI_RUN consecutive values with 3 multiplications per value and ONE inversion.
static constexpr int BI_RUN = 8;
template <int MOD> __global__ void __launch_bounds__(64) k_batch_inverse(const uint32_t *__restrict__ x, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t i0 = t * BI_RUN;
  if (i0 >= n) return;
  fe v[BI_RUN], pre[BI_RUN];
  fe acc = fe_one();
#pragma unroll
  for (int k = 0; k < BI_RUN; k++) {
    v[k] = (i0 + k < n) ? fe_load(x + (size_t)(i0 + k) * 8) : fe_zero();
    pre[k] = acc;
    if (!fe_is_zero(v[k])) acc = fe_mul<MOD>(acc, v[k]);          // rec0 skips zeros (BatchInverse.hs:18)
  }
  fe y = fe_modinv<MOD>(acc);      // division steps (modinv.hip.h): every lane inverts here, and their stream is input-independent
#pragma unroll
  for (int k = BI_RUN - 1; k >= 0; k--) {
    fe r = fe_zero();
    if (!fe_is_zero(v[k])) { r = fe_mul<MOD>(y, pre[k]); y = fe_mul<MOD>(y, v[k]); }   // rec1 (BatchInverse.hs:21-24)
    if (i0 + k < n) fe_store(out + (size_t)(i0 + k) * 8, r);
  }
}

// ------------------------------------------------------------------------------------------------ host side
using bppp_host::U256;

static int sums_finish(bppp_ctx *ctx, uint32_t *d_out, uint64_t sx[4], uint64_t sr[4]) {
  int rc = ensure_pinned(ctx, 64); if (rc) return rc;
  BPPP_HIP(ctx, hipGetLastError());
  BPPP_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_out, 64, hipMemcpyDeviceToHost, ctx->stream));
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(sx, ctx->pinned, 32); memcpy(sr, (char *)ctx->pinned + 32, 32);
  return BPPP_OK;
}

int norm_round_sums_run(bppp_ctx *ctx, const void *d_x, size_t n, const uint64_t q4[4], uint64_t sx[4], uint64_t sr[4]) {
  if (!sx || !sr || !q4) return fail(ctx, BPPP_ERR_ARG, "norm_round_sums: null pointer");
  if (n == 0) { memset(sx, 0, 32); memset(sr, 0, 32); return BPPP_OK; }
  if (!d_x || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "norm_round_sums: bad input");
  int rc = ensure_workspace(ctx, 256); if (rc) return rc;
  U256 q = U256::load(q4), s = q;
  for (int i = 0; i < 8; i++) s = bppp_host::mmul(s, s, bppp_host::FR());   // q4^256
  uint64_t sw[4]; s.store(sw);
  k_norm_round_sums<<<dim3(1), dim3(256), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, to_fr4(q4), to_fr4(sw), (uint32_t *)ctx->ws);
  return sums_finish(ctx, (uint32_t *)ctx->ws, sx, sr);
}
int lin_round_sums_run(bppp_ctx *ctx, const void *d_c, const void *d_x, size_t n, uint64_t sx[4], uint64_t sr[4]) {
  if (!sx || !sr) return fail(ctx, BPPP_ERR_ARG, "lin_round_sums: null pointer");
  if (n == 0) { memset(sx, 0, 32); memset(sr, 0, 32); return BPPP_OK; }
  if (!d_x || !d_c || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "lin_round_sums: bad input");
  int rc = ensure_workspace(ctx, 256); if (rc) return rc;
  k_lin_round_sums<<<dim3(1), dim3(256), 0, ctx->stream>>>((const uint32_t *)d_c, (const uint32_t *)d_x, (uint32_t)n, (uint32_t *)ctx->ws);
  return sums_finish(ctx, (uint32_t *)ctx->ws, sx, sr);
}
int round_openings_run(bppp_ctx *ctx, const void *d_x, size_t n, int scale, const uint64_t q[4], const uint64_t qinv[4], void *d_xw, void *d_rw) {
  if (n == 0) return BPPP_OK;
  if (!d_x || !d_xw || !d_rw || (scale && (!q || !qinv)) || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "round_openings: bad input");
  uint64_t z[4] = {0, 0, 0, 0};
  uint32_t np = (uint32_t)((n + 1) / 2);
  k_round_openings<<<dim3((np + 255) / 256), dim3(256), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, scale, to_fr4(scale ? q : z),
                                                                         to_fr4(scale ? qinv : z), (uint32_t *)d_xw, (uint32_t *)d_rw);
  BPPP_HIP(ctx, hipGetLastError());
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
int fold_scalars_launch(bppp_ctx *ctx, const uint64_t u[4], const uint64_t v[4], const void *d_x, size_t n, void *d_out) {
  if (n == 0) return BPPP_OK;
  if (!d_x || !d_out || !u || !v || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "fold_scalars: bad input");
  uint32_t np = (uint32_t)((n + 1) / 2);
  k_fold_scalars<<<dim3((np + 255) / 256), dim3(256), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, to_fr4(u), to_fr4(v), (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}
int fold_scalars_run(bppp_ctx *ctx, const uint64_t u[4], const uint64_t v[4], const void *d_x, size_t n, void *d_out) {
  int rc = fold_scalars_launch(ctx, u, v, d_x, n, d_out);
  if (rc) return rc;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
int batch_inverse_run(bppp_ctx *ctx, const void *d_x, size_t n, int modulus, void *d_out) {
  if (n == 0) return BPPP_OK;
  if (!d_x || !d_out || n >= (1ull << 31) || (modulus != 0 && modulus != 1)) return fail(ctx, BPPP_ERR_ARG, "batch_inverse: bad input");
  uint32_t threads = (uint32_t)((n + BI_RUN - 1) / BI_RUN);
  if (modulus) k_batch_inverse<1><<<dim3((threads + 63) / 64), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, (uint32_t *)d_out);
  else k_batch_inverse<0><<<dim3((threads + 63) / 64), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_x, (uint32_t)n, (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
int tensor_run(bppp_ctx *ctx, const uint64_t *bs, size_t nb, const uint64_t *es, const uint64_t *qs, size_t k, void *d_out) {
  if (nb == 0) return BPPP_OK;
  if (!bs || !d_out || (k && (!es || !qs)) || k > 30 || (nb << k) >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "tensor: bad input");
  size_t need = (nb + 2 * k) * 32;
  int rc = ensure_workspace(ctx, need); if (rc) return rc;
  rc = ensure_pinned(ctx, need); if (rc) return rc;
  uint64_t *h = (uint64_t *)ctx->pinned;
  memcpy(h, bs, nb * 32);
  // factor table: q_r for bit r = 0 (first round first), then e_r; es arrives LAST ROUND FIRST
  for (size_t r = 0; r < k; r++) {
    memcpy(h + 4 * (nb + r), qs + 4 * r, 32);
    memcpy(h + 4 * (nb + k + r), es + 4 * (k - 1 - r), 32);
  }
  BPPP_HIP(ctx, hipMemcpyAsync(ctx->ws, h, need, hipMemcpyHostToDevice, ctx->stream));
  uint64_t total = (uint64_t)nb << k;
  k_tensor<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>((const uint32_t *)ctx->ws, (uint32_t)nb,
                                                                                (const uint32_t *)ctx->ws + nb * 8, (int)k, (uint32_t *)d_out);
  BPPP_HIP(ctx, hipGetLastError());
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}

}  // namespace bppp
