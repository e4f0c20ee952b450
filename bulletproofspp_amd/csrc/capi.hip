// capi.hip — extern "C" boundary of libbppp_hip.so (see include/bppp.h for the contract and the
// reference interface each symbol replaces).
#include <stdlib.h>
#include <string.h>
#include "ctx.hpp"
#include "hostmath.hpp"
#include "hosteis.hpp"

namespace bppp {
int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);
int fold_points_run(bppp_ctx *, const uint64_t *, int, const uint64_t *, int, const void *, size_t, void *);
int norm_round_sums_run(bppp_ctx *, const void *, size_t, const uint64_t *, uint64_t *, uint64_t *);
int lin_round_sums_run(bppp_ctx *, const void *, const void *, size_t, uint64_t *, uint64_t *);
int round_openings_run(bppp_ctx *, const void *, size_t, int, const uint64_t *, const uint64_t *, void *, void *);
int fold_scalars_run(bppp_ctx *, const uint64_t *, const uint64_t *, const void *, size_t, void *);
int tensor_run(bppp_ctx *, const uint64_t *, size_t, const uint64_t *, const uint64_t *, size_t, void *);
int lift_x_run(bppp_ctx *, const void *, size_t, void *);
int batch_inverse_run(bppp_ctx *, const void *, size_t, int, void *);
int fold_points_eis_run(bppp_ctx *, const uint64_t *, const int *, const uint64_t *, const int *, const void *, size_t, void *);

void ctx_retain(bppp_ctx *ctx) { ctx->refs.fetch_add(1); }
void ctx_release(bppp_ctx *ctx) {
  if (ctx->refs.fetch_sub(1) != 1) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);
  if (ctx->ws) hipFree(ctx->ws);
  if (ctx->ws2) hipFree(ctx->ws2);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->ev_ready) for (int i = 0; i <= BPPP_NUM_STAGES; i++) hipEventDestroy(ctx->ev[i]);
  if (ctx->aux_fork) hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_join) hipEventDestroy(ctx->aux_join);
  if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}
}  // namespace bppp
void MsmTune::from_env() {
  auto geti = [](const char *n) { const char *e = getenv(n); return e ? atoi(e) : 0; };
  if (const char *e = getenv("BPPP_GCOST")) gcost = atof(e);
  cmin = geti("BPPP_CMIN"); lw = geti("BPPP_LW"); rg = geti("BPPP_RG"); marg_s = geti("BPPP_MARG_S"); lacc = geti("BPPP_LACC");
  window_batched = geti("BPPP_WINDOW_BATCHED"); comb_wpe = geti("BPPP_COMB_WPE"); comb_rows_waves = geti("BPPP_COMB_ROWS_WAVES"); reduce_old = getenv("BPPP_REDUCE_OLD") != nullptr;
  small_c = geti("BPPP_MSM_SMALL_C"); small_len = geti("BPPP_MSM_SMALL_LEN"); small_max = geti("BPPP_MSM_SMALL_MAX"); hist_ch = geti("BPPP_HIST_CH"); no_small = getenv("BPPP_MSM_NO_SMALL") != nullptr; comb_no_wsplit = getenv("BPPP_COMB_NO_WSPLIT") != nullptr; comb_no_packed = getenv("BPPP_COMB_NO_PACKED") != nullptr; if (getenv("BPPP_COMB_ROWS_MIN_MB")) comb_rows_min_bytes = (size_t)strtoull(getenv("BPPP_COMB_ROWS_MIN_MB"), nullptr, 10) << 20; no_balance = getenv("BPPP_MSM_NO_BALANCE") != nullptr;
}
namespace bppp {
int ctx_aux(bppp_ctx *ctx) {
  if (ctx->aux_stream) return BPPP_OK;
  BPPP_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  BPPP_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
  BPPP_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
  return BPPP_OK;
}
int ensure_workspace(bppp_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return BPPP_OK;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws) BPPP_HIP(ctx, hipFree(ctx->ws));
  ctx->ws = nullptr; ctx->ws_bytes = 0;
  size_t want = bytes + bytes / 8 + (1 << 20);
  BPPP_HIP(ctx, hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return BPPP_OK;
}
int ensure_scratch(bppp_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->ws2_bytes) return BPPP_OK;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws2) BPPP_HIP(ctx, hipFree(ctx->ws2));
  ctx->ws2 = nullptr; ctx->ws2_bytes = 0;
  size_t want = bytes + bytes / 8 + (1 << 20);
  BPPP_HIP(ctx, hipMalloc(&ctx->ws2, want));
  ctx->ws2_bytes = want;
  return BPPP_OK;
}
int ensure_pinned(bppp_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->pinned_bytes) return BPPP_OK;
  if (ctx->pinned) BPPP_HIP(ctx, hipHostFree(ctx->pinned));
  ctx->pinned = nullptr; ctx->pinned_bytes = 0;
  size_t want = bytes < 65536 ? 65536 : bytes * 2;
  BPPP_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
  ctx->pinned_bytes = want;
  return BPPP_OK;
}
void prof_mark(bppp_ctx *ctx, int idx) {
  if (!ctx->profile) return;
  if (!ctx->ev_ready) {
    for (int i = 0; i <= BPPP_NUM_STAGES; i++) hipEventCreate(&ctx->ev[i]);
    ctx->ev_ready = true;
  }
  hipEventRecord(ctx->ev[idx], ctx->stream);
}
void prof_collect(bppp_ctx *ctx, int nmarks) {
  if (!ctx->profile || !ctx->ev_ready) return;
  hipEventSynchronize(ctx->ev[nmarks - 1]);
  for (int i = 0; i + 1 < nmarks; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]) == hipSuccess) ctx->stage_ms[i] += ms;
  }
  ctx->calls++;
}
}  // namespace bppp

using namespace bppp;

extern "C" {

const char *bppp_version(void) { return "bppp-hip 0.1 (gfx950)"; }

int bppp_ctx_create(int device, bppp_ctx **out) {
  if (!out) return BPPP_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return BPPP_ERR_NODEVICE;
  bppp_ctx *ctx = new bppp_ctx();
  ctx->device = device;
  ctx->tune.from_env();
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) ctx->tune.num_cus = cus; }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return BPPP_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return BPPP_OK;
}
void bppp_ctx_destroy(bppp_ctx *ctx) {
  if (!ctx) return;
  if (ctx->closed.exchange(true)) return;      // a second destroy of the same handle is ignored while children keep it alive
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  // a caller-bound stream may not outlive this call: fall back to the context's own for whatever the children still do
  ctx->stream = ctx->own_stream;
  ctx_release(ctx);
}
int bppp_ctx_set_stream(bppp_ctx *ctx, void *hip_stream) {
  if (!ctx) return BPPP_ERR_ARG;
  hipStreamSynchronize(ctx->stream);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return BPPP_OK;
}
const char *bppp_last_error(const bppp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

#define CTX_ENTER(ctx)                         \
  if (!(ctx)) return BPPP_ERR_ARG;             \
  if ((ctx)->closed.load()) return BPPP_ERR_ARG; \
  if (hipSetDevice((ctx)->device) != hipSuccess) return bppp::fail(ctx, BPPP_ERR_HIP, "hipSetDevice failed")

int bppp_msm_batch_device(bppp_ctx *ctx, const void *d_scalars, const void *d_points_xy, size_t n, size_t batch, int shared_points,
                          int window_bits, uint64_t *out_xy) {
  CTX_ENTER(ctx);
  return msm_run(ctx, d_scalars, d_points_xy, n, batch, shared_points, window_bits, out_xy);
}
int bppp_msm_device(bppp_ctx *ctx, const void *d_scalars, const void *d_points_xy, size_t n, int window_bits, uint64_t out_xy[8]) {
  CTX_ENTER(ctx);
  return msm_run(ctx, d_scalars, d_points_xy, n, 1, 1, window_bits, out_xy);
}
namespace {
struct PointUpload { bppp_ctx *ctx; void *dst; const void *src; size_t bytes; };
int upload_points_now(void *arg) {
  PointUpload *u = (PointUpload *)arg;
  if (hipMemcpyAsync(u->dst, u->src, u->bytes, hipMemcpyHostToDevice, u->ctx->stream) != hipSuccess) return fail(u->ctx, BPPP_ERR_HIP, "msm: point upload failed");
  return BPPP_OK;
}
}  // namespace
int bppp_msm(bppp_ctx *ctx, const uint64_t *scalars, const uint64_t *points_xy, size_t n, uint64_t out_xy[8]) {
  CTX_ENTER(ctx);
  if (!out_xy) return fail(ctx, BPPP_ERR_ARG, "msm: null output");
  if (n == 0) { memset(out_xy, 0, 64); return BPPP_OK; }
  if (!scalars || !points_xy) return fail(ctx, BPPP_ERR_ARG, "msm: null input");
  // inputs staged in the context's second grow-only buffer (no hipMalloc per call); the scalars go first, the points are
  // copied by the hook msm_run calls just before its accumulate kernel: the host-side copy of 64 B/pair then runs while the
  // GPU recodes and sorts the digits
  { int rc0 = ensure_scratch(ctx, n * 96 + 256); if (rc0) return rc0; }
  char *ds = (char *)ctx->ws2, *dp = ds + ((n * 32 + 255) & ~(size_t)255);
  if (hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(ctx, BPPP_ERR_HIP, "msm: scalar upload failed");
  PointUpload up{ctx, dp, points_xy, n * 64};
  ctx->pre_acc = upload_points_now; ctx->pre_acc_arg = &up;
  int rc = msm_run(ctx, ds, dp, n, 1, 1, 0, out_xy);
  ctx->pre_acc = nullptr;
  hipStreamSynchronize(ctx->stream);
  return rc;
}

// The tail of a sharded MSM: the N partial points that the ranks all-gathered (one 64-B affine point each; RCCL has no mod-p
// reduction) are added here, on the host, with the complete group law — the same place and for the same reason as the MSM's
// final Horner combine: a handful of dependent additions is latency, not throughput.
int bppp_sum_points(bppp_ctx *ctx, const uint64_t *points_xy, size_t n, uint64_t out_xy[8]) {
  using namespace bppp_host;
  if (!ctx) return BPPP_ERR_ARG;
  if (!out_xy || (n && !points_xy)) return fail(ctx, BPPP_ERR_ARG, "sum_points: null pointer");
  if (n > 65536) return fail(ctx, BPPP_ERR_ARG, "sum_points: meant for a few partial points; use bppp_msm for more");
  HJac acc = hj_inf();
  for (size_t i = 0; i < n; i++) {
    HAff a{U256::load(points_xy + 8 * i), U256::load(points_xy + 8 * i + 4)};
    if (cmp(a.x, FQ().m) >= 0 || cmp(a.y, FQ().m) >= 0) return fail(ctx, BPPP_ERR_ARG, "sum_points: coordinate not canonical");
    acc = hj_add(acc, hj_from_aff(a));
  }
  HAff r = hj_to_aff(acc);
  r.x.store(out_xy); r.y.store(out_xy + 4);
  return BPPP_OK;
}

int bppp_rational_reduce(const uint64_t x[4], uint64_t a_mag[3], int *a_neg, uint64_t b_mag[3], int *b_neg) {
  if (!x || !a_mag || !a_neg || !b_mag || !b_neg) return BPPP_ERR_ARG;
  bppp_host::U256 v = bppp_host::U256::load(x);
  if (bppp_host::cmp(v, bppp_host::FR().m) >= 0) return BPPP_ERR_ARG;
  auto ab = bppp_host::rational_reduce_scalar(v);
  memcpy(a_mag, ab.first.m, 24); *a_neg = ab.first.neg;
  memcpy(b_mag, ab.second.m, 24); *b_neg = ab.second.neg;
  return BPPP_OK;
}

// rationalReduceScalar of the FastPrime / Eisenstein configuration (Commitment.hs:242-255 over :293-306; Eis.hs:72-82)
int bppp_rational_reduce_eis(const uint64_t x[4], uint64_t a_mag[4], int a_neg[2], uint64_t b_mag[4], int b_neg[2]) {
  if (!x || !a_mag || !a_neg || !b_mag || !b_neg) return BPPP_ERR_ARG;
  bppp_host::U256 v = bppp_host::U256::load(x);
  if (bppp_host::cmp(v, bppp_host::FR().m) >= 0) return BPPP_ERR_ARG;
  auto ab = bppp_eis::rational_reduce_eis(v);
  const bppp_eis::Big *comp[4] = {&ab.first.a, &ab.first.b, &ab.second.a, &ab.second.b};
  for (int k = 0; k < 4; k++) {
    for (int i = 2; i < bppp_eis::Big::L; i++) if (comp[k]->m[i]) return BPPP_ERR_ARG;      // cannot happen: components are ~65 bits
    uint64_t *dst = (k < 2 ? a_mag : b_mag) + 2 * (k & 1);
    dst[0] = comp[k]->m[0]; dst[1] = comp[k]->m[1];
    (k < 2 ? a_neg : b_neg)[k & 1] = comp[k]->neg ? 1 : 0;
  }
  return BPPP_OK;
}
int bppp_fold_points_eis_device(bppp_ctx *ctx, const uint64_t b_mag[4], const int b_neg[2], const uint64_t a_mag[4], const int a_neg[2], const void *d_points_xy,
                                size_t n, void *d_out_xy) {
  CTX_ENTER(ctx);
  int rc = fold_points_eis_run(ctx, b_mag, b_neg, a_mag, a_neg, d_points_xy, n, d_out_xy);
  if (rc) return rc;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}

int bppp_fold_points_device(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg, const void *d_points_xy,
                            size_t n, void *d_out_xy) {
  CTX_ENTER(ctx);
  int rc = fold_points_run(ctx, b_mag, b_neg, a_mag, a_neg, d_points_xy, n, d_out_xy);
  if (rc) return rc;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
int bppp_fold_points(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg, const uint64_t *points_xy, size_t n,
                     uint64_t *out_xy) {
  CTX_ENTER(ctx);
  if (n == 0) return BPPP_OK;
  if (!points_xy || !out_xy) return fail(ctx, BPPP_ERR_ARG, "fold_points: null pointer");
  size_t np = (n + 1) / 2;
  void *dp = nullptr, *dout = nullptr;
  BPPP_HIP(ctx, hipMalloc(&dp, n * 64));
  if (hipMalloc(&dout, np * 64) != hipSuccess) { hipFree(dp); return fail(ctx, BPPP_ERR_HIP, "hipMalloc failed"); }
  int rc = BPPP_OK;
  if (hipMemcpyAsync(dp, points_xy, n * 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, BPPP_ERR_HIP, "upload failed");
  if (!rc) rc = fold_points_run(ctx, b_mag, b_neg, a_mag, a_neg, dp, n, dout);
  if (!rc && hipMemcpyAsync(out_xy, dout, np * 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, BPPP_ERR_HIP, "download failed");
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail(ctx, BPPP_ERR_HIP, "fold_points: kernel failed");
  hipFree(dp); hipFree(dout);
  return rc;
}

int bppp_norm_round_sums_device(bppp_ctx *ctx, const void *d_x, size_t n, const uint64_t q4[4], uint64_t sx[4], uint64_t sr[4]) {
  CTX_ENTER(ctx);
  return norm_round_sums_run(ctx, d_x, n, q4, sx, sr);
}
int bppp_lin_round_sums_device(bppp_ctx *ctx, const void *d_c, const void *d_x, size_t n, uint64_t sx[4], uint64_t sr[4]) {
  CTX_ENTER(ctx);
  return lin_round_sums_run(ctx, d_c, d_x, n, sx, sr);
}
int bppp_norm_round_openings_device(bppp_ctx *ctx, const void *d_x, size_t n, const uint64_t q[4], const uint64_t qinv[4], void *d_xw, void *d_rw) {
  CTX_ENTER(ctx);
  return round_openings_run(ctx, d_x, n, 1, q, qinv, d_xw, d_rw);
}
int bppp_lin_round_openings_device(bppp_ctx *ctx, const void *d_x, size_t n, void *d_xw, void *d_rw) {
  CTX_ENTER(ctx);
  return round_openings_run(ctx, d_x, n, 0, nullptr, nullptr, d_xw, d_rw);
}
int bppp_fold_scalars_device(bppp_ctx *ctx, const uint64_t u[4], const uint64_t v[4], const void *d_x, size_t n, void *d_out) {
  CTX_ENTER(ctx);
  return fold_scalars_run(ctx, u, v, d_x, n, d_out);
}
int bppp_tensor_device(bppp_ctx *ctx, const uint64_t *bs, size_t nb, const uint64_t *es, const uint64_t *qs, size_t k, void *d_out) {
  CTX_ENTER(ctx);
  return tensor_run(ctx, bs, nb, es, qs, k, d_out);
}

int bppp_batch_inverse_device(bppp_ctx *ctx, const void *d_x, size_t n, int modulus, void *d_out) {
  CTX_ENTER(ctx);
  return batch_inverse_run(ctx, d_x, n, modulus, d_out);
}
int bppp_lift_x_device(bppp_ctx *ctx, const void *d_x, size_t n, void *d_points_xy) {
  CTX_ENTER(ctx);
  return lift_x_run(ctx, d_x, n, d_points_xy);
}

int bppp_device_alloc(bppp_ctx *ctx, size_t bytes, void **d_ptr) {
  CTX_ENTER(ctx);
  if (!d_ptr) return fail(ctx, BPPP_ERR_ARG, "device_alloc: null pointer");
  BPPP_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
  return BPPP_OK;
}
int bppp_device_free(bppp_ctx *ctx, void *d_ptr) {
  CTX_ENTER(ctx);
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  BPPP_HIP(ctx, hipFree(d_ptr));
  return BPPP_OK;
}
int bppp_host_alloc(bppp_ctx *ctx, size_t bytes, void **ptr) {
  CTX_ENTER(ctx);
  if (!ptr) return fail(ctx, BPPP_ERR_ARG, "host_alloc: null pointer");
  BPPP_HIP(ctx, hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
  return BPPP_OK;
}
int bppp_host_free(bppp_ctx *ctx, void *ptr) {
  CTX_ENTER(ctx);
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));      // a copy out of the buffer may still be in flight,
  if (ctx->aux_stream) BPPP_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));   // on either stream (the verifier's sliced uploads use the second)
  BPPP_HIP(ctx, hipHostFree(ptr));
  return BPPP_OK;
}
int bppp_upload(bppp_ctx *ctx, void *d_dst, const void *src, size_t bytes) {
  CTX_ENTER(ctx);
  if (!bytes) return BPPP_OK;
  BPPP_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}
int bppp_download(bppp_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
  CTX_ENTER(ctx);
  if (!bytes) return BPPP_OK;
  BPPP_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BPPP_OK;
}

int bppp_profile_enable(bppp_ctx *ctx, int on) {
  if (!ctx) return BPPP_ERR_ARG;
  ctx->profile = on != 0;
  return BPPP_OK;
}
int bppp_profile_read(bppp_ctx *ctx, double ms[BPPP_NUM_STAGES], uint64_t *calls, int reset) {
  if (!ctx || !ms) return BPPP_ERR_ARG;
  for (int i = 0; i < BPPP_NUM_STAGES; i++) ms[i] = ctx->stage_ms[i];
  if (calls) *calls = ctx->calls;
  if (reset) { for (int i = 0; i < BPPP_NUM_STAGES; i++) ctx->stage_ms[i] = 0; ctx->calls = 0; }
  return BPPP_OK;
}

}  // extern "C"
