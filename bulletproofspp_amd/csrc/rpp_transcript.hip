// rpp_transcript.hip — the Fiat-Shamir transcript and the prover's randomness for B proofs in lockstep, on the device.
//
// Every challenge of a proof is shaOracle over the WHOLE transcript so far (app/Main.hs:75-80, src/ZKP.hs:96-101: new commitments go in
// front, `show`n as decimal coordinates); the prover's random scalars are hashToScalar prefix . show counter (app/Main.hs:83-87;
// ZKPT.random, src/ZKP.hs:88-92).  Shared by the three lockstep provers: typed-reciprocal (csrc/rpprove_dev.hip), binary
// (csrc/brpprove_dev.hip) and the inner-product argument (csrc/ipb.hip).
//   k_rpp_draws                       every random scalar of every proof: one SHA-256 per lane
//   k_rpp_text_prepend / k_rpp_hash   the transcript text of every proof (newest commitment first) grows at its FRONT; a challenge is
//                                     SHA-256 (header <> text from the current start)
//   RppTranscript                     the host object that queues them: headers of all oracle calls uploaded once, so a proof is ONE stream of
//                                     kernels; for a handful of proofs the hashing moves to the host cores (a host core hashes a 64by64
//                                     transcript in ~50 us, one GPU lane needs ~600 us)
#include <string.h>
#include <thread>
#include "fe.hip.h"
#include "rphash.hip.h"
#include "rpp_transcript.hpp"

namespace bppp {

// ------------------------------------------------------------------------------------------------ randomness
// random n = hash (prefix <> show n) (hashToScalar, app/Main.hs:83-84; ZKPT.random, src/ZKP.hs:88-92), n = 0 .. nd-1
__global__ void __launch_bounds__(64) k_rpp_draws(const uint8_t *__restrict__ prefix, uint32_t plen, uint32_t batch, uint32_t nd, uint32_t *__restrict__ rnd) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * nd) return;
  const uint32_t b = (uint32_t)(g / nd), c = (uint32_t)(g % nd);
  // decimal digits of the counter (at most 10), most significant first
  uint32_t dig[10], nd10 = 0, v = c;
  do { dig[nd10++] = v % 10u; v /= 10u; } while (v);
  const uint8_t *pre = prefix + (size_t)b * plen;
  const uint32_t mlen = plen + nd10, nblk = (mlen + 9 + 63) / 64;
  uint32_t st[8], w[16];
  sha256_init(st);
  for (uint32_t blk = 0; blk < nblk; blk++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      uint32_t word = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t p = blk * 64 + 4 * i + j;
        uint32_t byte = 0;
        if (p < plen) byte = pre[p];
        else if (p < mlen) {
          const uint32_t k = nd10 - 1 - (p - plen);
          uint32_t dv = 0;
#pragma unroll
          for (int q = 0; q < 10; q++) if ((uint32_t)q == k) dv = dig[q];
          byte = '0' + dv;
        } else if (p == mlen) byte = 0x80;
        word = (word << 8) | byte;
      }
      w[i] = word;
    }
    if (blk == nblk - 1) { w[14] = 0; w[15] = mlen * 8; }
    sha256_compress(st, w);
  }
  fe r; sha256_digest_to_limbs(st, r.v);
  fe t; const uint32_t br = raw_sub(t, r, fr_modulus());
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : t.v[i];
  fe_store(rnd + g * 8, r);
}

// ------------------------------------------------------------------------------------------------ transcript
// The text of a proof sits RIGHT-ALIGNED in its buffer [b * stride, (b + 1) * stride - 16): new commitments are written in front
// of the current start (ZKPT.oracle prepends, src/ZKP.hs:98).  One workgroup per proof; pts is [batch][m] affine.
__global__ void __launch_bounds__(256) k_rpp_text_prepend(const uint32_t *__restrict__ pts, uint32_t m, uint8_t *__restrict__ text, uint32_t stride,
                                                          uint32_t *__restrict__ tstart) {
  extern __shared__ uint32_t lens[];            // [m + 1]
  __shared__ uint32_t wsum[4];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  for (uint32_t t = tid; t < m; t += 256) {
    const uint32_t *p = pts + ((size_t)b * m + t) * 16;
    lens[t] = dec_convert(fe_load(p)).len + dec_convert(fe_load(p + 8)).len;
  }
  __syncthreads();
  const uint32_t per = (m + 255) / 256, lo = min(m, tid * per), hi = min(m, lo + per);
  uint32_t s = 0;
  for (uint32_t t = lo; t < hi; t++) s += lens[t];
  uint32_t inc = s;
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((int)(tid & 63) >= d) inc += o; }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - s;
  for (uint32_t w = 0; w < (tid >> 6); w++) run += wsum[w];
  __syncthreads();
  for (uint32_t t = lo; t < hi; t++) { const uint32_t l = lens[t]; lens[t] = run; run += l; }
  if (tid == 255) lens[m] = run;
  __syncthreads();
  const uint32_t total = lens[m], start = tstart[b] - total;
  uint8_t *tx = text + (size_t)b * stride + start;
  for (uint32_t t = tid; t < m; t += 256) {
    const uint32_t *p = pts + ((size_t)b * m + t) * 16;
    const Dec dx = dec_convert(fe_load(p)), dy = dec_convert(fe_load(p + 8));
    uint8_t *end = tx + lens[t] + dx.len + dy.len;
    end = dec_write_backward(dy, end);
    dec_write_backward(dx, end);
  }
  __syncthreads();
  if (tid == 0) tstart[b] = start;
}

// the same for calls that add a HANDFUL of points (m <= 8: every call after the first adds one or two): eight lanes per proof, 32 proofs per workgroup —
// a 256-lane workgroup per proof leaves 250 lanes idle and costs 0.54 ms per call at 8192 single-value proofs
__global__ void __launch_bounds__(256) k_rpp_text_prepend_small(const uint32_t *__restrict__ pts, uint32_t m, uint32_t batch, uint8_t *__restrict__ text, uint32_t stride,
                                                                uint32_t *__restrict__ tstart) {
  const uint32_t b = blockIdx.x * 32 + threadIdx.x / 8, t = threadIdx.x & 7u;
  const bool act = b < batch && t < m;
  Dec dx, dy;
  dx.len = dy.len = 0;
  if (act) {
    const uint32_t *p = pts + ((size_t)b * m + t) * 16;
    dx = dec_convert(fe_load(p)); dy = dec_convert(fe_load(p + 8));
  }
  const uint32_t len = act ? dx.len + dy.len : 0u;
  uint32_t inc = len;                                         // inclusive scan over the 8 lanes of a proof
  for (int d = 1; d < 8; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 8); if ((int)t >= d) inc += o; }
  const uint32_t total = __shfl(inc, 7, 8);
  const uint32_t start = (b < batch ? tstart[b] : 0u) - total;     // the eight lanes of a proof sit in one wavefront: all have read tstart[b] before lane 0 stores
  if (act) {
    uint8_t *end = text + (size_t)b * stride + start + inc;   // this point's text ends at its inclusive offset
    end = dec_write_backward(dy, end);
    dec_write_backward(dx, end);
  }
  if (t == 0 && b < batch) tstart[b] = start;
}

// challenge n (1 <= n <= count) of every proof: SHA-256 (hdr_n <> text from the current start); out slot of challenge n: ch_slot[n-1]
// into ch[b][7] (slot < 7) or es[b] (slot = 7)
struct RppHdrs { uint32_t hdr_be[3][RP_HDR_MAX / 4]; uint32_t hlen[3]; uint32_t slot[3]; };
// 64 hashes per workgroup of two wavefronts (producer / consumer, rphash.hip.h): g = n * batch + b
__global__ void __launch_bounds__(128) k_rpp_hash(const RppHdrs *__restrict__ H, uint32_t count, uint32_t batch, const uint8_t *__restrict__ text, uint32_t stride,
                                                  const uint32_t *__restrict__ tstart, uint32_t tend, uint32_t *__restrict__ ch, uint32_t *__restrict__ es) {
  __shared__ uint32_t lds[RP_HASH_PC_LDS_WORDS];
  const uint64_t g = (uint64_t)blockIdx.x * 64 + (threadIdx.x & 63u);
  const bool active = g < (uint64_t)batch * count;
  const uint32_t n = active ? (uint32_t)(g / batch) : 0u, b = active ? (uint32_t)(g % batch) : 0u;
  const uint32_t s = tstart[b];
  const fe v = rp_hash_to_fr_pc(active, H->hdr_be[n], H->hlen[n], text + (size_t)b * stride + s, tend - s, lds);
  if (!active || threadIdx.x < 64) return;
  const uint32_t slot = H->slot[n];
  if (slot < 7) fe_store(ch + ((size_t)b * 7 + slot) * 8, v);
  else fe_store(es + (size_t)b * 8, v);
}

}  // namespace bppp

using namespace bppp;

namespace bppp {

size_t RppTranscript::hdr_bytes(size_t ncalls) { return ncalls * sizeof(RppHdrs); }

int rpp_draws(bppp_ctx *ctx, const uint8_t *d_prefix, size_t prefix_len, size_t batch, size_t nd, uint32_t *d_rnd) {
  const uint64_t n = (uint64_t)batch * nd;
  if (!n) return BPPP_OK;
  k_rpp_draws<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream>>>(d_prefix, (uint32_t)prefix_len, (uint32_t)batch, (uint32_t)nd, d_rnd);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}

int RppTranscript::begin(bppp_rp *rp_, size_t batch, const std::vector<RppCall> &layer_calls, size_t rounds, bool host_oracle, uint8_t *d_text, uint32_t *d_tstart,
                         void *d_hdrs, uint32_t *d_ch, uint32_t *d_es) {
  rp = rp_; B = batch; calls = layer_calls; host = host_oracle;
  text = d_text; tstart = d_tstart; hdrs = d_hdrs; ch = d_ch; es = d_es;
  stride = rp->D.text_stride; tend = stride - 16;
  for (size_t r = 0; r < rounds; r++) calls.push_back(RppCall{2, 1, 7});
  bppp_ctx *ctx = rp->ctx;
  hipStream_t st = ctx->stream;
  std::vector<uint32_t> ts(B, tend);
  BPPP_HIP(ctx, hipMemcpyAsync(tstart, ts.data(), B * 4, hipMemcpyHostToDevice, st));
  std::vector<RppHdrs> hh(calls.size());
  size_t np_total = 0;
  for (size_t c = 0; c < calls.size(); c++) {
    memset(&hh[c], 0, sizeof(RppHdrs));
    if (calls[c].count < 1 || calls[c].count > 3) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: an oracle call has 1 .. 3 outputs");
    np_total += calls[c].points;
    for (uint32_t n = 1; n <= calls[c].count; n++) {
      const std::string hs = rp->tag + std::to_string(n) + std::to_string(np_total);
      if (hs.size() > (size_t)RP_HDR_MAX) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: oracle tag too long");
      rp_pack_header(hs, hh[c].hdr_be[n - 1]); hh[c].hlen[n - 1] = (uint32_t)hs.size();
      hh[c].slot[n - 1] = calls[c].first_slot < 7 ? calls[c].first_slot + (n - 1) : 7u;
    }
  }
  BPPP_HIP(ctx, hipMemcpyAsync(hdrs, hh.data(), hh.size() * sizeof(RppHdrs), hipMemcpyHostToDevice, st));
  BPPP_HIP(ctx, hipStreamSynchronize(st));          // ts, hh go out of scope
  groups.assign(host ? B : 0, std::vector<std::string>());
  np.assign(host ? B : 0, 0);
  return BPPP_OK;
}

int RppTranscript::call(const uint32_t *pts_dev, size_t call_index) {
  bppp_ctx *ctx = rp->ctx;
  hipStream_t st = ctx->stream;
  if (call_index >= calls.size()) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: oracle call out of plan");
  const size_t m = calls[call_index].points; const int count = (int)calls[call_index].count; const uint32_t first_slot = calls[call_index].first_slot;
  if (!host) {
    if (m <= 8 && B >= 256) k_rpp_text_prepend_small<<<dim3((unsigned)((B + 31) / 32)), dim3(256), 0, st>>>(pts_dev, (uint32_t)m, (uint32_t)B, text, stride, tstart);
    else k_rpp_text_prepend<<<dim3((unsigned)B), dim3(256), (m + 1) * 4, st>>>(pts_dev, (uint32_t)m, text, stride, tstart);
    const uint64_t n = (uint64_t)B * count;
    k_rpp_hash<<<dim3((unsigned)((n + 63) / 64)), dim3(128), 0, st>>>((const RppHdrs *)hdrs + call_index, (uint32_t)count, (uint32_t)B, text, stride, tstart, tend, ch, es);
    BPPP_HIP(ctx, hipGetLastError());
    return BPPP_OK;
  }
  std::vector<uint64_t> hp(B * m * 8), ho(B * 3 * 4);
  BPPP_HIP(ctx, hipMemcpyAsync(hp.data(), pts_dev, B * m * 64, hipMemcpyDeviceToHost, st));
  BPPP_HIP(ctx, hipStreamSynchronize(st));
  {
    auto work = [&](size_t lo, size_t hi) { for (size_t b = lo; b < hi; b++) rpp_host_oracle(rp->tag, groups[b], np[b], &hp[b * m * 8], m, count, &ho[b * 12]); };
    const size_t nt = std::min<size_t>(B / 16, 16);          // a proof's call is ~10 us of hashing and text: threads only pay from a few dozen proofs
    if (nt <= 1) work(0, B);
    else {
      std::vector<std::thread> th;
      for (size_t t = 0; t < nt; t++) th.emplace_back(work, B * t / nt, B * (t + 1) / nt);
      for (auto &x : th) x.join();
    }
  }
  if (first_slot >= 7) BPPP_HIP(ctx, hipMemcpy2DAsync(es, 32, ho.data(), 96, 32, B, hipMemcpyHostToDevice, st));
  else BPPP_HIP(ctx, hipMemcpy2DAsync(ch + first_slot * 8, 7 * 32, ho.data(), 96, (size_t)count * 32, B, hipMemcpyHostToDevice, st));
  BPPP_HIP(ctx, hipStreamSynchronize(st));          // hp, ho go out of scope
  return BPPP_OK;
}

}  // namespace bppp
