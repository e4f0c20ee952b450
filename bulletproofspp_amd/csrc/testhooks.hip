// testhooks.hip — device field / group arithmetic exposed to the parity tests (include/bppp_test.h).
#include "../../include/bppp_test.h"
#include <vector>
#include "ctx.hpp"
#include "ec.hip.h"
#include "modinv.hip.h"
#include "fr26.hip.h"

namespace bppp {
template <int MOD> BPPP_DI fe apply_op(int op, const fe &a, const fe &b) {
  switch (op) {
    case BPPP_FE_ADD: return fe_add<MOD>(a, b);
    case BPPP_FE_SUB: return fe_sub<MOD>(a, b);
    case BPPP_FE_MUL: return fe_mul<MOD>(a, b);
    case BPPP_FE_SQR: return fe_sqr<MOD>(a);
    case BPPP_FE_INV: return fe_inv<MOD>(a);
    case 7: return fe_inv_vartime<MOD>(a);       // binary extended Euclid (fe.hip.h)
    case 8: return fe_modinv<MOD>(a);            // safegcd division steps (modinv.hip.h)
    default: return fe_neg<MOD>(a);
  }
}
// Fq through the production representation (10 x 26-bit limbs, csrc/fq26.hip.h)
BPPP_DI fe apply_op_fq(int op, const fe &a, const fe &b) {
  fq x = fq_from_fe(a), y = fq_from_fe(b);
  switch (op) {
    case BPPP_FE_ADD: return fq_to_fe(fq_add(x, y));
    case BPPP_FE_SUB: return fq_to_fe(fq_sub<1>(x, y));
    case BPPP_FE_MUL: return fq_to_fe(fq_mul(x, y));
    case BPPP_FE_SQR: return fq_to_fe(fq_sqr(x));
    case BPPP_FE_INV: return fq_to_fe(fq_inv(x));            // production: safegcd
    case 9: return fq_to_fe(fq_inv_fermat(x));               // the addition chain, kept as a cross-check
    default: return fq_to_fe(fq_neg<1>(x));
  }
}
// worst-case magnitudes: (8a) * (8b) with both operands built by repeated lazy additions
BPPP_DI fe apply_mag8_mul(const fe &a, const fe &b) {
  fq x = fq_from_fe(a), y = fq_from_fe(b);
  fq x8 = fq_mul_int(x, 8), y8 = fq_neg<7>(fq_mul_int(y, 7));   // magnitudes 8 and 8
  return fq_to_fe(fq_mul(x8, y8));                               // = -56 a b
}
// Fr through the 10 x 26-bit lazy limbs the verifier's and provers' scalar kernels use (csrc/fr26.hip.h)
BPPP_DI fe apply_op_fr26(int op, const fe &a, const fe &b) {
  fr x = fr_from_fe(a), y = fr_from_fe(b);
  switch (op) {
    case BPPP_FE_ADD: return fr_to_fe(fr_add(x, y));
    case BPPP_FE_SUB: return fr_to_fe(fr_sub<1>(x, y));
    case BPPP_FE_MUL: return fr_to_fe(fr_mul(x, y));
    case BPPP_FE_SQR: return fr_to_fe(fr_sqr(x));
    case 6: {                                                 // (8a) * (-7b): both operands at the magnitude-8 bound of fr_mul
      fr x8 = fr_mul_int(x, 8), y8 = fr_neg<7>(fr_mul_int(y, 7));
      return fr_to_fe(fr_mul(x8, y8));
    }
    case 10: {                                                // (8a)^2 through fr_sqr's doubled operand (a2 = a << 1 < 2^31)
      return fr_to_fe(fr_sqr(fr_mul_int(x, 8)));
    }
    case 11: {                                                // magnitude 16 into fr_normalize: 8a - 7b with sub<7>, result magnitude 16
      return fr_to_fe(fr_sub<7>(fr_mul_int(x, 8), fr_mul_int(y, 7)));
    }
    case 12: {                                                // the reduced linear operations, chained: ((a + b) - 2b) + (-a) + 2a = a - b + ... = 2a - b
      fr t = fr_subr(fr_addr(x, y), fr_dblr(y));             // a - b
      return fr_to_fe(fr_addr(fr_addr(t, fr_negr(x)), fr_dblr(x)));   // a - b - a + 2a = 2a - b
    }
    case 13: return fr_to_fe(fr_weak(fr_mul_int(x, 16)));    // one weak pass from magnitude 16, then normalise: 16a
    case 14: return fe{{(uint32_t)fr_is_zero(fr_sub<7>(fr_mul_int(x, 8), fr_mul_int(y, 7))), 0, 0, 0, 0, 0, 0, 0}};   // 8a == 7b ? (magnitude 16)
    default: return fr_to_fe(fr_neg<1>(x));
  }
}
__global__ void k_test_fe(int op, int mod, const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_load(a + (size_t)i * 8), y = fe_load(b + (size_t)i * 8);
  fe r;
  if (mod == 1) r = apply_op<1>(op, x, y);            // Fr, 8 x 32
  else if (mod == 2) r = apply_op<0>(op, x, y);       // Fq, legacy 8 x 32 code path (cross-check)
  else if (mod == 3) r = apply_op_fr26(op, x, y);     // Fr, production 10 x 26 (fr26.hip.h)
  else if (op == 6) r = apply_mag8_mul(x, y);
  else r = apply_op_fq(op, x, y);                     // Fq, production 10 x 26
  fe_store(out + (size_t)i * 8, r);
}
__global__ void k_test_point(int op, const uint32_t *p, const uint32_t *q, uint32_t n, uint32_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aff P = aff_load(p + (size_t)i * 16), Q = aff_load(q + (size_t)i * 16);
  xyzz acc = xyzz_from_aff(P);
  if (op == 0) xyzz_madd(acc, Q);
  else if (op == 1) { xyzz t = xyzz_dbl_aff(Q); xyzz_madd(t, aff_cneg(Q, true)); /* t = Q in non-trivial XYZZ form */ xyzz_add(acc, t); }
  else acc = xyzz_dbl(xyzz_dbl_aff(P));  // 4P via both doubling forms
  aff_store(out + (size_t)i * 16, xyzz_to_aff(acc));
}
}  // namespace bppp

using namespace bppp;

static int run2(bppp_ctx *ctx, const uint64_t *a, const uint64_t *b, size_t n, size_t words, uint64_t *out, void **da, void **db, void **dout) {
  size_t bytes = n * words * 8;
  BPPP_HIP(ctx, hipMalloc(da, bytes)); BPPP_HIP(ctx, hipMalloc(db, bytes)); BPPP_HIP(ctx, hipMalloc(dout, bytes));
  BPPP_HIP(ctx, hipMemcpyAsync(*da, a, bytes, hipMemcpyHostToDevice, ctx->stream));
  BPPP_HIP(ctx, hipMemcpyAsync(*db, b, bytes, hipMemcpyHostToDevice, ctx->stream));
  (void)out;
  return BPPP_OK;
}

extern "C" int bppp_test_fe_op(bppp_ctx *ctx, int op, int modulus, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
  if (!ctx || !a || !b || !out) return BPPP_ERR_ARG;
  if (n == 0) return BPPP_OK;
  hipSetDevice(ctx->device);
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  int rc = run2(ctx, a, b, n, 4, out, &da, &db, &dout);
  if (!rc) {
    k_test_fe<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>(op, modulus, (const uint32_t *)da, (const uint32_t *)db, (uint32_t)n, (uint32_t *)dout);
    if (hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
      rc = bppp::fail(ctx, BPPP_ERR_HIP, "test_fe_op: kernel or copy failed");
  }
  hipFree(da); hipFree(db); hipFree(dout);
  return rc;
}
extern "C" int bppp_test_point_op(bppp_ctx *ctx, int op, const uint64_t *p, const uint64_t *q, size_t n, uint64_t *out) {
  if (!ctx || !p || !q || !out) return BPPP_ERR_ARG;
  if (n == 0) return BPPP_OK;
  hipSetDevice(ctx->device);
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  int rc = run2(ctx, p, q, n, 8, out, &da, &db, &dout);
  if (!rc) {
    k_test_point<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>(op, (const uint32_t *)da, (const uint32_t *)db, (uint32_t)n, (uint32_t *)dout);
    if (hipMemcpyAsync(out, dout, n * 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
      rc = bppp::fail(ctx, BPPP_ERR_HIP, "test_point_op: kernel or copy failed");
  }
  hipFree(da); hipFree(db); hipFree(dout);
  return rc;
}

// ---- measured VALU ceiling for the field layer: independent fq_mul chains, 8 waves per SIMD, nothing but multiplies.
// bench.py reports the accumulate kernel's modular-multiplication rate as a fraction of this.
namespace bppp {
__global__ void __launch_bounds__(256) k_mulmod_rate(const uint32_t *__restrict__ seed, int iters, uint32_t *__restrict__ sink) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  fq a[4], b;
  for (int k = 0; k < 4; k++) a[k] = fq_from_fe(fe_load(seed + (size_t)((t + 17 * k) & 1023) * 8));
  b = fq_from_fe(fe_load(seed + (size_t)((t * 7 + 3) & 1023) * 8));
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = fq_mul(a[k], b);
  }
  fq r = fq_add(fq_add(a[0], a[1]), fq_add(a[2], a[3]));
  uint32_t x = 0;
  for (int i = 0; i < 10; i++) x ^= r.n[i];
  if (x == 0x12345u) sink[t & 63] = x;       // keeps the chains live; practically never taken
}
}  // namespace bppp
extern "C" int bppp_test_mulmod_rate(bppp_ctx *ctx, int iters, double *mulmods_per_sec) {
  if (!ctx || !mulmods_per_sec || iters < 1) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  uint32_t *seed = nullptr, *sink = nullptr;
  if (hipMalloc(&seed, 1024 * 32) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { hipFree(seed); return bppp::fail(ctx, BPPP_ERR_HIP, "mulmod_rate: hipMalloc"); }
  std::vector<uint32_t> h(1024 * 8);
  uint64_t z = 0x9E3779B97F4A7C15ull;
  for (auto &w : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; w = (uint32_t)(z >> 16); }
  for (int i = 0; i < 1024; i++) h[8 * i + 7] &= 0x7FFFFFFFu;      // < p
  hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(sink, 0, 256);
  const int blocks = 256 * 4 * 8 / 4 * 2;     // 8 waves per SIMD on 256 CUs, two rounds of them
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bppp::k_mulmod_rate<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(seed, 8, sink);          // warm-up
  hipEventRecord(e0, ctx->stream);
  bppp::k_mulmod_rate<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(seed, iters, sink);
  hipEventRecord(e1, ctx->stream);
  int rc = BPPP_OK;
  if (hipEventSynchronize(e1) != hipSuccess) rc = bppp::fail(ctx, BPPP_ERR_HIP, "mulmod_rate: kernel failed");
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  *mulmods_per_sec = ms > 0 ? (double)blocks * 256.0 * 4.0 * iters / (ms * 1e-3) : 0.0;
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(seed); hipFree(sink);
  return rc;
}
