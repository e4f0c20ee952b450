// basis.hip — a registered basis with fixed-base precomputation (SURVEY.md 8(b) "Ownership": `bppp_basis_upload`).
//
// The Pedersen bases of a setup are fixed — G, H, g come from the setup once (src/RangeProof/TypedReciprocal.hs:348-359,
// src/RangeProof/Binary.hs:147-148) and every later commit / verifyBPM call multiplies scalars onto the SAME points
// (commitRPW, src/RangeProof/Internal.hs:45-50).  A handle keeps them in HBM together with the table
//     T[w][i] = 2^(c w) * P_i          w < W = 256 / c + 1
// so that the signed digit of scalar i in window w addresses T[w][i] and ALL windows fall into one set of 2^(c-1) buckets: an MSM
// over a registered basis has one bucket reduction instead of W and no window combine (csrc/msm.hip, table_stride != 0).  The
// group element is the one `innerProduct` defines (src/Commitment.hs:325-335); only the route differs.
#include <string.h>
#include "comb.hpp"
#include "ctx.hpp"
#include "ec.hip.h"

namespace bppp {
int msm_run_ex(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *, size_t, uint32_t *);

// one lane per point: the chain P, 2^c P, 2^(2c) P, ... (c doublings and one normalisation per step; canonical affine rows)
__global__ void __launch_bounds__(64) k_basis_table(const uint32_t *__restrict__ pts, uint32_t n, int c, int W, uint32_t *__restrict__ table) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aff P = aff_load(pts + (size_t)i * 16);
  aff_store(table + (size_t)i * 16, P);
  for (int w = 1; w < W; w++) {
    xyzz a = xyzz_dbl_aff(P);
    for (int k = 1; k < c; k++) a = xyzz_dbl(a);
    P = xyzz_to_aff(a);
    aff_store(table + ((size_t)w * n + i) * 16, P);
  }
}

// the flat route prices one bucket reduction per instance (not per window): wider windows pay off sooner
static int choose_window_flat(size_t n, size_t batch) {
  double best = 1e300; int bc = 8;
  for (int c = 4; c <= 16; c++) {
    const int full = 254 / c, r = 255 - c * full;
    const double weff = full + 1 + (r == c ? 0.5 : 0.0);
    const bool groups = c <= 9 && (double)batch >= 4096.0;
    const double cost = weff * (double)n + (groups ? 4.5 : 10.0) * (double)(1u << (c - 1));
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}
}  // namespace bppp

using namespace bppp;

struct bppp_basis {
  bppp_ctx *ctx;
  size_t n;
  int c, W;
  uint32_t *table;       // [W][n] affine
  CombTable *comb;       // optional: every multiple of every window (bppp_basis_enable_comb), for many instances over a short basis
  uint32_t *d_out; size_t out_cap;
};

extern "C" {

void bppp_basis_destroy(bppp_basis *h) {
  if (!h) return;
  bppp_ctx *ctx = h->ctx;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (h->table) hipFree(h->table);
  if (h->d_out) hipFree(h->d_out);
  if (h->comb) comb_destroy(h->comb);
  delete h;
  ctx_release(ctx);
}

// d_points_xy: n affine points already in HBM (copied into the handle's table; the caller may free them afterwards).
// window_bits = 0: chosen for MSMs of all n terms, `batch_hint` instances at a time.
int bppp_basis_create_device(bppp_ctx *ctx, const void *d_points_xy, size_t n, int window_bits, size_t batch_hint, bppp_basis **out) {
  if (!ctx || !out || ctx_closed(ctx)) return BPPP_ERR_ARG;
  *out = nullptr;
  if (!d_points_xy || !n || n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "basis_create: bad arguments");
  const int c = window_bits ? window_bits : choose_window_flat(n, batch_hint ? batch_hint : 1);
  if (c < 2 || c > 16) return fail(ctx, BPPP_ERR_ARG, "basis_create: window_bits must be in [2,16]");
  const int W = 256 / c + 1;
  if ((uint64_t)W * n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "basis_create: windows * n must be < 2^31");
  hipSetDevice(ctx->device);
  bppp_basis *h = new bppp_basis();
  h->ctx = ctx; ctx_retain(ctx); h->n = n; h->c = c; h->W = W; h->table = nullptr; h->comb = nullptr; h->d_out = nullptr; h->out_cap = 0;
  auto fill = [&]() -> int {
    BPPP_HIP(ctx, hipMalloc(&h->table, (size_t)W * n * 64));
    k_basis_table<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>((const uint32_t *)d_points_xy, (uint32_t)n, c, W, h->table);
    BPPP_HIP(ctx, hipGetLastError());
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BPPP_OK;
  };
  if (int rc = fill()) { bppp_basis_destroy(h); return rc; }
  *out = h;
  return BPPP_OK;
}

int bppp_basis_create(bppp_ctx *ctx, const uint64_t *points_xy, size_t n, int window_bits, size_t batch_hint, bppp_basis **out) {
  if (!ctx || !out || ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!points_xy || !n) return fail(ctx, BPPP_ERR_ARG, "basis_create: bad arguments");
  hipSetDevice(ctx->device);
  void *d = nullptr;
  BPPP_HIP(ctx, hipMalloc(&d, n * 64));
  int rc = BPPP_OK;
  if (hipMemcpy(d, points_xy, n * 64, hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, BPPP_ERR_HIP, "basis_create: upload failed");
  if (!rc) rc = bppp_basis_create_device(ctx, d, n, window_bits, batch_hint, out);
  hipFree(d);
  return rc;
}

int bppp_basis_info(const bppp_basis *h, size_t *n, int *window_bits, size_t *table_bytes) {
  if (!h) return BPPP_ERR_ARG;
  if (n) *n = h->n;
  if (window_bits) *window_bits = h->c;
  if (table_bytes) *table_bytes = (size_t)h->W * h->n * 64;
  return BPPP_OK;
}

// `batch` MSMs over the first n_terms points of the registered basis: d_scalars is [batch][n_terms] canonical scalars in HBM,
// out_xy [batch][8] on the host.  Same results as bppp_msm_batch_device(shared_points = 1) over those points.
int bppp_msm_basis(bppp_basis *h, const void *d_scalars, size_t n_terms, size_t batch, uint64_t *out_xy) {
  if (!h) return BPPP_ERR_ARG;
  bppp_ctx *ctx = h->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (n_terms > h->n) return fail(ctx, BPPP_ERR_ARG, "msm_basis: more terms than registered points");
  hipSetDevice(ctx->device);
  if (h->comb && batch >= 64 && n_terms) {      // one wavefront per instance: pays once there are enough instances to fill the chip's SIMDs
    if (!d_scalars || !out_xy) return fail(ctx, BPPP_ERR_ARG, "msm_basis: null argument");
    if (batch > h->out_cap) {
      BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (h->d_out) BPPP_HIP(ctx, hipFree(h->d_out));
      h->d_out = nullptr; h->out_cap = 0;
      BPPP_HIP(ctx, hipMalloc(&h->d_out, batch * 64 + comb_scratch_bytes(1023)));      // results, then room for the partial sums of any small launch
      h->out_cap = batch;
    }
    int rc = comb_msm(h->comb, (const uint32_t *)d_scalars, batch, h->d_out, ctx->stream, false, n_terms, h->d_out + h->out_cap * 16, comb_scratch_bytes(1023));
    if (rc) return rc;
    BPPP_HIP(ctx, hipMemcpyAsync(out_xy, h->d_out, batch * 64, hipMemcpyDeviceToHost, ctx->stream));
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BPPP_OK;
  }
  return msm_run_ex(ctx, d_scalars, h->table, n_terms, batch, 1, h->c, out_xy, h->n, nullptr);
}

// Adds the comb table tab[w][i][d-1] = d 2^(c w) P_i (csrc/comb.hip) to a registered basis: bppp_msm_basis of >= 64 instances then
// costs one mixed addition per non-zero digit and nothing else.  window_bits = 0: the widest window (<= 18) whose table fits
// budget_bytes.  The table stays until the handle is destroyed; *table_bytes (may be NULL) reports its size.
int bppp_basis_enable_comb(bppp_basis *h, int window_bits, size_t budget_bytes, int *window_bits_out, size_t *table_bytes) {
  if (!h) return BPPP_ERR_ARG;
  bppp_ctx *ctx = h->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!h->comb) {
    int rc = comb_create(ctx, h->table, h->n, window_bits, budget_bytes, &h->comb);      // row 0 of the table is the basis itself
    if (rc) { h->comb = nullptr; return rc; }
  }
  if (window_bits_out) *window_bits_out = h->comb->c;
  if (table_bytes) *table_bytes = h->comb->bytes;
  return BPPP_OK;
}

}  // extern "C"
