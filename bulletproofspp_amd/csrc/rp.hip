// rp.hip — the range-proof layer end to end: encoded proofs in, accept / reject out, for a whole batch, on the device.
//
// Replaces, for B proofs of ONE typed-reciprocal setup, the reference's unit of verification work:
//   decodeProof' / decodeCommitments          src/RangeProof.hs:68-85, src/Encoding.hs:92-128   (x-only points, packed signs)
//   verifyM of RangeProof                     src/RangeProof.hs:103-105
//     verifyTRRPM                             src/RangeProof/TypedReciprocal.hs:447-467
//       its three oracle calls                :459-462, through ZKPT.oracle (src/ZKP.hs:96-101) and shaOracle (app/Main.hs:64-80)
//     verifyBPM                               src/Bulletproof.hs:370-378 (one more oracle call per round, :374)
// with the random-linear-combination batch check the reference only sketches (TODOs at TypedReciprocal.hs:469-472,
// RangeProof.hs:103-106; semantics SURVEY.md 8(c)).
//
// Pipeline (all kernels on the context's stream, nothing but the final 64-byte point and the status words return to the host):
//   k_rp_decode_points   one lane per encoded point: Binary (Prime p) x (4 big-endian words, least significant first), toP,
//                        pointX (square root by the (p+1)/4 addition chain), fromXWithSign (Encoding.hs:97-103)
//   k_rp_decode_scalars  the final witness scalars of the proof file, toP
//   k_rp_text            one workgroup per proof: `show x <> show y` of every commitment in transcript order (newest first),
//                        compacted into one text per proof + the offset of every commitment
//   k_rp_hash            one lane per (proof, oracle output): SHA-256 of  tag <> show n <> show (length ps) <> text suffix,
//                        digest -> field by Binary (Prime p); 7 + rounds challenges per proof and the batch weight rho
//   k_trrp_public        (csrc/trrp.hip) verifyTRRPM's scalar work from the challenges
//   bppp_nl_verify_batch_device (csrc/nlbatch.hip) challenge expansion, shared-basis merge, ONE combined MSM
//
// The same file holds the setup handle (`bppp_rp`: ranges, layout, basis resident in HBM) and, further down, the batch prover.
#include <string.h>
#include <string>
#include <chrono>
#include <thread>
#include <vector>
#include "ctx.hpp"
#include "ec.hip.h"
#include "hostmath.hpp"
#include "rpsetup.hpp"
#include "sha256.hip.h"
#include "rp_internal.hpp"
#include "brp.hpp"
#include "rphash.hip.h"
#include "fr26.hip.h"

namespace bppp {

// ------------------------------------------------------------------------------------------------ decode
// Binary (Prime p) get (Encoding.hs:76-80): limb i = big-endian 64-bit word at bytes 8i..8i+7, least-significant limb first;
// toP reduces (one conditional subtraction: the value is < 2^256 < 2m)
template <int MOD> BPPP_DI fe load_field_be(const uint8_t *p) {
  fe v;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint8_t *q = p + 8 * i;
    v.v[2 * i + 1] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    v.v[2 * i] = ((uint32_t)q[4] << 24) | ((uint32_t)q[5] << 16) | ((uint32_t)q[6] << 8) | q[7];
  }
  fe t;
  uint32_t br = raw_sub(t, v, modulus<MOD>());
#pragma unroll
  for (int i = 0; i < 8; i++) v.v[i] = br ? v.v[i] : t.v[i];
  return v;
}

// Point t of a proof IN TRANSCRIPT ORDER (newest first, the order shaOracle's final call sees, src/ZKP.hs:98):
//   t < 2k            the argument's responses, last round first  = bpComs of the proof file (RangeProof.hs:60-66)
//   2k <= t < 2k + 4  blCom, rCom, dmCom, mCom  (Binary: blCom, dCom) = rpComs of the proof file
//   else              the input commitments                         = the commitments file
// Output: responses to resp[b][t], the rest to init[b][...] in the order blCom : rCom : dmCom : mCom : nComs.
__global__ void __launch_bounds__(64) k_rp_decode_points(RpDims D, uint32_t batch, const uint8_t *__restrict__ coms, const uint8_t *__restrict__ proofs,
                                                         uint32_t *__restrict__ init_pts, uint32_t *__restrict__ resp_pts, uint32_t *__restrict__ bad,
                                                         uint32_t *__restrict__ any_bad) {   // bad[b] per proof of this launch, *any_bad for the call
  const uint32_t npts = rp_npts(D);
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * npts) return;
  const uint32_t b = (uint32_t)(g / npts), t = (uint32_t)(g % npts);
  const uint8_t *signs, *xs;
  uint32_t idx;
  const uint32_t nproof_pts = D.nrp + 2 * D.k;
  if (t < 2 * D.k + D.nrp) {
    const uint8_t *pf = proofs + (size_t)b * D.proof_bytes + (size_t)(D.fn + D.fl) * 32;
    signs = pf; xs = pf + (nproof_pts + 7) / 8;
    idx = t < 2 * D.k ? D.nrp + t : t - 2 * D.k;
  } else {
    const uint8_t *cf = coms + (size_t)b * D.coms_bytes;
    signs = cf; xs = cf + (D.nr + 7) / 8;
    idx = t - 2 * D.k - D.nrp;
  }
  const bool want_big = (signs[idx >> 3] >> (idx & 7)) & 1;
  const fe xe = load_field_be<0>(xs + (size_t)idx * 32);
  const fq x = fq_from_fe(xe);
  fq seven = fq_zero(); seven.n[0] = 7;
  const fq rhs = fq_add(fq_mul(fq_sqr(x), x), seven);         // magnitude 2
  fq y = fq_sqrt_candidate(rhs);
  const bool ok = fq_normalizes_to_zero(fq_sub<2>(fq_sqr(y), rhs));
  y = fq_normalize(y);
  // fromXWithSign (Encoding.hs:97-103): keep the root whose (y > p - y) equals the sign bit
  const fe ye = fq_to_fe(y), yn = fe_neg<0>(ye);
  fe d;
  const bool y_big = raw_sub(d, yn, ye) != 0;                  // -y < y
  aff r; r.x = x; r.y = (y_big != want_big) ? fq_from_fe(yn) : y;
  if (!ok) { r = aff_inf(); atomicOr(bad + b, 1u); atomicOr(any_bad, 1u); }
  uint32_t *out = t < 2 * D.k ? resp_pts + ((size_t)b * 2 * D.k + t) * 16
                              : init_pts + ((size_t)b * (D.nrp + D.nr) + (t - 2 * D.k)) * 16;
  aff_store(out, r);
}

// final witness scalars: norm part then linear part (encodeProof', RangeProof.hs:60-66)
__global__ void __launch_bounds__(64) k_rp_decode_scalars(RpDims D, uint32_t batch, const uint8_t *__restrict__ proofs, uint32_t *__restrict__ wit_norm,
                                                          uint32_t *__restrict__ wit_lin) {
  const uint32_t ns = D.fn + D.fl;
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)batch * ns) return;
  const uint32_t b = (uint32_t)(g / ns), i = (uint32_t)(g % ns);
  const fe v = load_field_be<1>(proofs + (size_t)b * D.proof_bytes + (size_t)i * 32);
  if (i < D.fn) fe_store(wit_norm + ((size_t)b * D.fn + i) * 8, v);
  else fe_store(wit_lin + ((size_t)b * D.fl + (i - D.fn)) * 8, v);
}

// ------------------------------------------------------------------------------------------------ transcript text (helpers: rphash.hip.h)
BPPP_DI const uint32_t *rp_point_ptr(const RpDims &D, const uint32_t *init_pts, const uint32_t *resp_pts, uint32_t b, uint32_t t) {
  return t < 2 * D.k ? resp_pts + ((size_t)b * 2 * D.k + t) * 16 : init_pts + ((size_t)b * (D.nrp + D.nr) + (t - 2 * D.k)) * 16;
}

// One workgroup per proof.  Pass 1: the text length of every point (x digits + y digits); exclusive scan; pass 2: the digits
// again, written at the point's offset.  text_off[b][t] (t <= npts) are byte offsets into text[b]; a proof whose points did not all
// decode gets "0"s for the missing ones — its hashes are never used (bad[b] rejects it).
__global__ void __launch_bounds__(256) k_rp_text(RpDims D, const uint32_t *__restrict__ init_pts, const uint32_t *__restrict__ resp_pts,
                                                 uint8_t *__restrict__ text, uint32_t *__restrict__ text_off) {
  extern __shared__ uint32_t lens[];            // [npts + 1] lengths, then offsets
  __shared__ uint32_t wsum[4];
  const uint32_t npts = rp_npts(D), b = blockIdx.x, tid = threadIdx.x;
  for (uint32_t t = tid; t < npts; t += 256) {
    const uint32_t *p = rp_point_ptr(D, init_pts, resp_pts, b, t);
    lens[t] = dec_convert(fe_load(p)).len + dec_convert(fe_load(p + 8)).len;
  }
  __syncthreads();
  // exclusive scan of lens[0..npts): every thread owns a contiguous chunk
  const uint32_t per = (npts + 255) / 256, lo = min(npts, tid * per), hi = min(npts, lo + per);
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
  if (tid == 255) lens[npts] = run;             // total (thread 255's chunk ends the list, possibly empty)
  __syncthreads();
  uint8_t *tx = text + (size_t)b * D.text_stride;
  for (uint32_t t = tid; t <= npts; t += 256) text_off[(size_t)b * (npts + 1) + t] = lens[t];
  for (uint32_t t = tid; t < npts; t += 256) {
    const uint32_t *p = rp_point_ptr(D, init_pts, resp_pts, b, t);
    const Dec dx = dec_convert(fe_load(p)), dy = dec_convert(fe_load(p + 8));
    uint8_t *end = tx + lens[t] + dx.len + dy.len;
    end = dec_write_backward(dy, end);
    dec_write_backward(dx, end);
  }
}

// The same text through LDS (round 3): one lane per COORDINATE converts it once (the nine 9-digit chunks wait in LDS across the offset
// scan), the digits are written as bytes into an LDS image of the proof's text, and the image goes out in 16-byte coalesced stores —
// the first version converted every coordinate twice on 84 of its 256 lanes and wrote 13 KB per proof as single-byte global stores
// (0.245 ms per 4096 proofs).  Dynamic LDS: off[npts + 1] | len[2 npts] | chunks[2 npts][9] | image[text_stride]; setups whose image does
// not fit (thousands of commitments) keep k_rp_text.
__host__ __device__ inline size_t rp_text_lds_bytes(const RpDims &D) {
  const size_t npts = rp_npts(D);
  return (((npts + 1) + 2 * npts + 18 * npts) * 4 + 15) / 16 * 16 + D.text_stride;
}
__global__ void __launch_bounds__(256) k_rp_text_lds(RpDims D, const uint32_t *__restrict__ init_pts, const uint32_t *__restrict__ resp_pts,
                                                     uint8_t *__restrict__ text, uint32_t *__restrict__ text_off) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  __shared__ uint32_t wsum[4];
  const uint32_t npts = rp_npts(D), nc = 2 * npts, b = blockIdx.x, tid = threadIdx.x;
  uint32_t *off = sm, *clen = off + (npts + 1), *chunks = clen + nc;
  uint8_t *image = (uint8_t *)sm + (((npts + 1) + nc + 9 * nc) * 4 + 15) / 16 * 16;
  for (uint32_t c = tid; c < nc; c += 256) {
    const uint32_t *p = rp_point_ptr(D, init_pts, resp_pts, b, c >> 1) + (c & 1u) * 8;
    const Dec d = dec_convert(fe_load(p));
    clen[c] = d.len | (d.top << 16);
#pragma unroll
    for (int k = 0; k < 9; k++) chunks[c * 9 + k] = d.ch[k];
  }
  __syncthreads();
  // exclusive scan of the point lengths: every thread owns a contiguous chunk of points
  const uint32_t per = (npts + 255) / 256, lo = min(npts, tid * per), hi = min(npts, lo + per);
  uint32_t s = 0;
  for (uint32_t t = lo; t < hi; t++) s += (clen[2 * t] & 0xFFFFu) + (clen[2 * t + 1] & 0xFFFFu);
  uint32_t inc = s;
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((int)(tid & 63) >= d) inc += o; }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - s;
  for (uint32_t w = 0; w < (tid >> 6); w++) run += wsum[w];
  for (uint32_t t = lo; t < hi; t++) { off[t] = run; run += (clen[2 * t] & 0xFFFFu) + (clen[2 * t + 1] & 0xFFFFu); }
  if (tid == 255) off[npts] = run;
  __syncthreads();
  for (uint32_t t = tid; t <= npts; t += 256) text_off[(size_t)b * (npts + 1) + t] = off[t];
  for (uint32_t c = tid; c < nc; c += 256) {
    Dec d;
#pragma unroll
    for (int k = 0; k < 9; k++) d.ch[k] = chunks[c * 9 + k];
    d.len = clen[c] & 0xFFFFu; d.top = clen[c] >> 16;
    const uint32_t start = off[c >> 1] + ((c & 1u) ? (clen[c - 1] & 0xFFFFu) : 0u);
    dec_write_backward(d, image + start + d.len);
  }
  __syncthreads();
  const uint32_t total = off[npts], n16 = (total + 15) / 16;
  uint4 *dst = (uint4 *)(text + (size_t)b * D.text_stride);
  const uint4 *src = (const uint4 *)image;
  for (uint32_t i = tid; i < n16; i += 256) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ hashing
// Hash h of a proof (h < 7 + k):  0,1,2 -> e, x, r0   first oracle call  [dmCom, mCom] ++ nComs      (TypedReciprocal.hs:459)
//                                 3,4,5 -> q, x', r1  second call, rCom prepended                    (:460)
//                                 6     -> t          third call, blCom prepended                    (:462)
//                                 7 + j -> e of round j + 1 (first round first), (X, R) prepended    (Bulletproof.hs:374)
// The hashed message is  header_h <> text[b][off[start_h] ..]  with header_h = tag <> show n <> show (length ps), identical for
// every proof (precomputed on the host, HashPlan).
//
// The batch weight rho_b (k_rp_rho, one lane per proof) is bound to the verifier's seed, to the proof's GLOBAL position in the job and
// to the proof itself:  rho_b = decode (SHA-256 (seed[32] <> le64 (index_offset + b) <> t <> e_last <> final witness scalars)), every
// scalar as its 32 little-endian bytes; t (the third oracle call, TypedReciprocal.hs:462) and e_last (the last round's challenge,
// Bulletproof.hs:374) are hashes of every commitment and response of the proof, the final witness scalars are the only proof bytes no
// challenge covers.  No weight is fixed to 1: with the batch sharded proof-per-GPU the ranks pass their offsets, so no two proofs of a
// job share a weight and error terms cannot be made to cancel between ranks (or, with a known seed, inside one batch).
BPPP_DI uint32_t rp_rho_word(const RpDims &D, uint32_t i, const uint8_t *seed, uint64_t idx, const uint32_t *t, const uint32_t *e_last, const uint32_t *wn,
                             const uint32_t *wl) {
  if (i < 8) return ((uint32_t)seed[4 * i] << 24) | ((uint32_t)seed[4 * i + 1] << 16) | ((uint32_t)seed[4 * i + 2] << 8) | seed[4 * i + 3];
  if (i == 8) return __builtin_bswap32((uint32_t)idx);
  if (i == 9) return __builtin_bswap32((uint32_t)(idx >> 32));
  uint32_t j = i - 10;
  if (j < 8) return __builtin_bswap32(t[j]);
  j -= 8;
  if (D.k) { if (j < 8) return __builtin_bswap32(e_last[j]); j -= 8; }
  if (j < 8 * D.fn) return __builtin_bswap32(wn[j]);
  return __builtin_bswap32(wl[j - 8 * D.fn]);
}
__global__ void __launch_bounds__(64) k_rp_rho(RpDims D, uint32_t batch, uint64_t index_offset, const uint8_t *__restrict__ seed, const uint32_t *__restrict__ ch,
                                               const uint32_t *__restrict__ es, const uint32_t *__restrict__ wit_norm, const uint32_t *__restrict__ wit_lin,
                                               uint32_t *__restrict__ rho) {
  const uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  const uint32_t *t = ch + ((size_t)b * 7 + 6) * 8, *e_last = es + (size_t)b * D.k * 8;     // es is LAST round first
  const uint32_t *wn = wit_norm + (size_t)b * D.fn * 8, *wl = wit_lin + (size_t)b * D.fl * 8;
  const uint32_t nwords = 10 + 8 * (1 + (D.k ? 1 : 0) + D.fn + D.fl);
  const uint32_t nblocks = (nwords * 4 + 9 + 63) / 64;
  uint32_t st[8], w[16];
  sha256_init(st);
  for (uint32_t blk = 0; blk < nblocks; blk++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t wi = blk * 16 + i;
      uint32_t v = 0;
      if (wi < nwords) v = rp_rho_word(D, wi, seed, index_offset + b, t, e_last, wn, wl);
      else if (wi == nwords) v = 0x80000000u;
      if (blk == nblocks - 1 && i == 15) v = nwords * 32;
      w[i] = v;
    }
    sha256_compress(st, w);
  }
  fe v; sha256_digest_to_limbs(st, v.v);
  fe r; const uint32_t br = raw_sub(r, v, fr_modulus());
  for (int i = 0; i < 8; i++) v.v[i] = br ? v.v[i] : r.v[i];
  if (fe_is_zero(v)) v = fe_one();
  fe_store(rho + (size_t)b * 8, v);
}
// 64 hashes per workgroup of two wavefronts (producer / consumer, rphash.hip.h); hashes of one KIND sit in one workgroup (equal
// lengths, the same header): g = h * batch + b
__global__ void __launch_bounds__(128) k_rp_hash(RpDims D, uint32_t batch, uint32_t nhash, const HashPlan *__restrict__ plan, const uint8_t *__restrict__ text,
                                                 const uint32_t *__restrict__ text_off, uint32_t *__restrict__ ch, uint32_t *__restrict__ es) {
  __shared__ uint32_t lds[RP_HASH_PC_LDS_WORDS];
  const uint64_t g = (uint64_t)blockIdx.x * 64 + (threadIdx.x & 63u);
  const bool active = g < (uint64_t)batch * nhash;
  const uint32_t h = active ? (uint32_t)(g / batch) : 0u, b = active ? (uint32_t)(g % batch) : 0u;
  const HashPlan *pl = plan + h;
  const uint32_t npts = rp_npts(D);
  const uint32_t *off = text_off + (size_t)b * (npts + 1);
  const uint32_t t0 = off[pl->start_pt], t1 = off[npts];
  const fe v = rp_hash_to_fr_pc(active, pl->hdr_be, pl->hlen, text + (size_t)b * D.text_stride + t0, t1 - t0, lds);   // the suffix of the proof's text this call hashes
  if (!active || threadIdx.x < 64) return;                  // the consumer wavefront holds the digests
  const uint32_t slot = pl->out_slot;                      // < 7: a challenge of the range-proof layer (ch rows are 7 wide for both kinds), else 7 + round slot
  if (slot < 7) fe_store(ch + ((size_t)b * 7 + slot) * 8, v);
  else fe_store(es + ((size_t)b * D.k + (slot - 7)) * 8, v);
}


// ------------------------------------------------------------------------------------------------ RangeProof.Binary: public scalars
// The scalar work of verifyBRPM (src/RangeProof/Binary.hs:206-222) for a batch, one wavefront per proof, from the proof's challenges
// (q, x, r, t) = ch slots 0, 1, 2, 6:
//   makePublicConsts (:73-98): bss_i = x^(2(j+1)) b_i over the positions of the ranges that are not assumed; p_i = bss_i q0^-(i+1) - 1/2;
//                              sc = -2 (net' + sum_j min_j x^(2(j+1))) + sum_i q0^(i+1) p_i^2,  net' = -x netPublic when conserving
//   the argument's public opening (:215-219): scalar t^2 sc, norm vector t p_i (zero beyond the live positions), linear weights
//   [0, r t] (setupBRP's psv, :151-152), and the initCom scalars of TranscriptBRP (:107-110) in commitment order blCom : dCom : nComs =
//   1, t, 2 t^2 inputCoeffs (:127-129).  q0 = q^2 (NL) or -q^2 (IP) as qPowers' has it.
__global__ void __launch_bounds__(256) k_brp_public(BrpDims D, uint32_t batch, const uint32_t *__restrict__ pos_range, const uint32_t *__restrict__ pos_coeff,
                                                   const uint32_t *__restrict__ range_min, const uint32_t *__restrict__ range_flags, const uint32_t *__restrict__ net_public,
                                                   const uint32_t *__restrict__ ch, uint32_t *__restrict__ out_q, uint32_t *__restrict__ out_sp,
                                                   uint32_t *__restrict__ out_norm, uint32_t *__restrict__ out_cs, uint32_t *__restrict__ out_init) {
  extern __shared__ uint32_t lds[];               // [nr] x^(2(j+1)), then [blockDim.x] partial sums
  uint32_t *x2s = lds, *part = lds + (size_t)D.nr * 8;
  const uint32_t b = blockIdx.x, l = threadIdx.x, NT = blockDim.x;      // 64 .. 256 lanes per proof (a power of two)
  const uint32_t *c = ch + (size_t)b * 56;
  const fe q = fe_load(c), x = fe_load(c + 8), r = fe_load(c + 16), t = fe_load(c + 48);
  fe q0 = fe_sqr<1>(q);
  if (D.flavour) q0 = fe_neg<1>(q0);
  const fe q0i = fe_modinv<1>(q0);               // every lane the same division steps: no divergence, one pass
  const fe xx = fe_sqr<1>(x);
  auto powu = [](fe base, uint32_t e) { fe acc = fe_one(); while (e) { if (e & 1u) acc = fe_mul<1>(acc, base); base = fe_sqr<1>(base); e >>= 1; } return acc; };
  {
    fe xj = powu(xx, l + 1);
    const fe step = powu(xx, NT);
    for (uint32_t j = l; j < D.nr; j += NT) { for (int k = 0; k < 8; k++) x2s[j * 8 + k] = xj.v[k]; xj = fe_mul<1>(xj, step); }
  }
  __syncthreads();
  auto x2 = [&](uint32_t j) { fe v; for (int k = 0; k < 8; k++) v.v[k] = x2s[j * 8 + k]; return v; };
  fe half = fe_zero();                            // (n + 1) / 2
  { const fe n = fr_modulus(); uint32_t carry = 1; fe tt;
    for (int i = 0; i < 8; i++) { const uint64_t s_ = (uint64_t)n.v[i] + carry; tt.v[i] = (uint32_t)s_; carry = (uint32_t)(s_ >> 32); }
    for (int i = 0; i < 8; i++) half.v[i] = (tt.v[i] >> 1) | (i < 7 ? tt.v[i + 1] << 31 : carry << 31); }
  fe acc = fe_zero();
  {
    // the per-position part in 10 x 26-bit limbs (fr26.hip.h: 413 instructions per multiplication against 785): nlen / NT positions per lane,
    // seven multiplications each — all of this kernel's time at the 64 x 64-bit shape (4096 positions)
    auto powr = [](fr base, uint32_t e) { fr a = fr_one(); while (e) { if (e & 1u) a = fr_mul(a, base); base = fr_sqr(base); e >>= 1; } return a; };
    const fr q0r = fr_from_fe(q0), q0ir = fr_from_fe(q0i), tr_ = fr_from_fe(t), halfr = fr_from_fe(half);
    fr qp = powr(q0r, l + 1), qi = powr(q0ir, l + 1), accr = fr_zero();
    const fr qs = powr(q0r, NT), qis = powr(q0ir, NT);
    for (uint32_t i = l; i < D.nlen; i += NT) {
      fr p = fr_zero();
      if (i < D.nlive) {
        p = fr_sub<1>(fr_mul(fr_mul(fr_from_fe(x2(pos_range[i])), fr_load(pos_coeff + (size_t)i * 8)), qi), halfr);     // magnitude 3
        accr = fr_addr(accr, fr_mul(qp, fr_sqr(p)));
        p = fr_mul(tr_, p);
      }
      fr_store(out_norm + ((size_t)b * D.nlen + i) * 8, p);
      qp = fr_mul(qp, qs); qi = fr_mul(qi, qis);
    }
    acc = fr_to_fe(accr);
  }
  // z = -2 (net' + sum_j min_j x^(2(j+1)))  (assumed ranges contribute no minimum, :91)
  fe z = fe_zero();
  for (uint32_t j = l; j < D.nr; j += NT)
    if (!(range_flags[j] & 2u)) z = fe_add<1>(z, fe_mul<1>(fe_load(range_min + (size_t)j * 8), x2(j)));
  if (l == 0 && D.conserve) z = fe_sub<1>(z, fe_mul<1>(x, fe_load(net_public)));
  acc = fe_sub<1>(acc, fe_dbl<1>(z));
  for (int k = 0; k < 8; k++) part[l * 8 + k] = acc.v[k];
  __syncthreads();
  for (int d = (int)NT >> 1; d >= 1; d >>= 1) {
    if ((int)l < d) {
      fe a, o;
      for (int k = 0; k < 8; k++) { a.v[k] = part[l * 8 + k]; o.v[k] = part[(l + d) * 8 + k]; }
      a = fe_add<1>(a, o);
      for (int k = 0; k < 8; k++) part[l * 8 + k] = a.v[k];
    }
    __syncthreads();
  }
  const fe t2 = fe_sqr<1>(t);
  if (l == 0) {
    fe sc; for (int k = 0; k < 8; k++) sc.v[k] = part[k];
    fe_store(out_sp + (size_t)b * 8, fe_mul<1>(t2, sc));
    fe_store(out_q + (size_t)b * 8, q);
    fe_store(out_cs + (size_t)b * 16, fe_zero());
    fe_store(out_cs + (size_t)b * 16 + 8, fe_mul<1>(r, t));
    fe_store(out_init + (size_t)b * (2 + D.nr) * 8, fe_one());
    fe_store(out_init + ((size_t)b * (2 + D.nr) + 1) * 8, t);
  }
  const fe two_t2 = fe_dbl<1>(t2);
  for (uint32_t j = l; j < D.nr; j += NT) {
    const uint32_t fl = range_flags[j];            // bit 0: output, bit 1: assumed
    fe ic = (fl & 2u) ? fe_zero() : x2(j);
    if (D.conserve) ic = (fl & 1u) ? fe_sub<1>(ic, x) : fe_add<1>(ic, x);
    fe_store(out_init + ((size_t)b * (2 + D.nr) + 2 + j) * 8, fe_mul<1>(two_t2, ic));
  }
}

}  // namespace bppp

using namespace bppp;
using bppp_host::U256;

namespace {

std::string dec_str(uint64_t v) { return std::to_string(v); }

int rp_build_tables(bppp_rp *rp) {
  const bppp_rps::Setup &st = rp->st;
  std::vector<uint32_t> kind, rng, slot, psym, cs_slot, cs_sym, pub_out, pub_sym, assumed;
  std::vector<uint64_t> coeff, mins, syms, pub_amt;
  std::vector<U256> sym_vals;
  auto sym = [&](const U256 &v) -> uint32_t {
    for (size_t i = 0; i < sym_vals.size(); i++) if (sym_vals[i] == v) return (uint32_t)i;
    sym_vals.push_back(v);
    for (int i = 0; i < 4; i++) syms.push_back(v.w[i]);
    return (uint32_t)(sym_vals.size() - 1);
  };
  auto push = [](std::vector<uint64_t> &dst, const U256 &v) { for (int i = 0; i < 4; i++) dst.push_back(v.w[i]); };
  for (const bppp_rps::Pos &p : st.pos) {
    const uint32_t k = p.kind & 0xFFu;
    kind.push_back(p.kind); rng.push_back(p.range);
    slot.push_back(k == bppp_rps::POS_TYPING ? 0u : (uint32_t)st.slot_of(p.radix));
    psym.push_back((k == bppp_rps::POS_INLINE && p.sym_small) ? sym(U256::from_u64(p.sym_small)) : bppp_rps::POS_NO_SYM);
    push(coeff, p.coeff);
  }
  for (uint32_t b : st.m_bases)
    for (uint32_t s = 1; s < b; s++) { cs_slot.push_back((uint32_t)st.slot_of(b)); cs_sym.push_back(sym(U256::from_u64(s))); }
  for (const bppp_rps::RangeData &rd : st.rds) { push(mins, bppp_rps::s_mod_n(rd.lo)); assumed.push_back(rd.assumed ? 1u : 0u); }
  for (const bppp_rps::PublicVT &pv : st.pubs) { pub_out.push_back(pv.is_output ? 1u : 0u); pub_sym.push_back(sym(pv.type)); push(pub_amt, pv.amount); }
  auto p32 = [](std::vector<uint32_t> &v) { if (v.empty()) v.push_back(0); return v.data(); };
  auto p64 = [](std::vector<uint64_t> &v) { if (v.empty()) v.assign(4, 0); return v.data(); };
  const size_t nsyms = sym_vals.size();
  return bppp_trrp_create(rp->ctx, st.flavour, st.has_types ? 1 : 0, st.nlen, st.llen, st.rds.size(), p32(kind), p32(rng), p32(slot), p32(psym), p64(coeff), p64(mins),
                          p32(assumed), nsyms, p64(syms), p32(cs_slot), p32(cs_sym), st.pubs.size(), p32(pub_out), p64(pub_amt), p32(pub_sym), &rp->tabs);
}

// header_h = tag <> show n <> show (length ps) for the 7 + k oracle outputs of a verification (see k_rp_hash)
int rp_build_plan(bppp_rp *rp) {
  const uint32_t k = rp->D.k, nr = rp->D.nr;
  std::vector<HashPlan> plan;
  auto add = [&](uint32_t n, uint32_t count, uint32_t start, uint32_t slot) -> bool {
    HashPlan p; memset(&p, 0, sizeof p);
    std::string h = rp->tag + dec_str(n) + dec_str(count);
    if (h.size() > RP_HDR_MAX) return false;
    rp_pack_header(h, p.hdr_be);
    p.hlen = (uint32_t)h.size(); p.start_pt = start; p.out_slot = slot;
    plan.push_back(p);
    return true;
  };
  bool ok = true;
  if (rp->st.kind == 0) {
    for (uint32_t n = 1; n <= 3; n++) ok &= add(n, 2 + nr, 2 * k + 2, n - 1);          // e, x, r0
    for (uint32_t n = 1; n <= 3; n++) ok &= add(n, 3 + nr, 2 * k + 1, 3 + n - 1);      // q, x', r1
    ok &= add(1, 4 + nr, 2 * k, 6);                                                    // t
  } else {                                     // RangeProof.Binary: oracle' (dCom : nComs) (Binary.hs:179, :209), then oracle [blCom] (:189, :213)
    for (uint32_t n = 1; n <= 3; n++) ok &= add(n, 1 + nr, 2 * k + 1, n - 1);          // q, x, r  -> ch slots 0, 1, 2
    ok &= add(1, 2 + nr, 2 * k, 6);                                                    // t        -> ch slot 6 (where k_rp_rho reads it)
  }
  for (uint32_t j = 1; j <= k; j++) ok &= add(1, rp->D.nrp + nr + 2 * j, 2 * (k - j), 7 + (k - j));   // round j: es is LAST round first
  if (!ok) return fail(rp->ctx, BPPP_ERR_ARG, "rp_create: oracle tag too long");
  rp->nhash = (uint32_t)plan.size();
  BPPP_HIP(rp->ctx, hipMalloc(&rp->d_plan, plan.size() * sizeof(HashPlan)));
  BPPP_HIP(rp->ctx, hipMemcpy(rp->d_plan, plan.data(), plan.size() * sizeof(HashPlan), hipMemcpyHostToDevice));
  return BPPP_OK;
}

}  // namespace

extern "C" void bppp_basis_destroy(bppp_basis *basis);

static void brp_tabs_destroy(bppp_brp_tabs *t) {
  if (!t) return;
  for (uint32_t *p : {t->pos_range, t->pos_coeff, t->range_min, t->range_flags, t->net_public}) if (p) hipFree(p);
  delete t;
}
static int brp_build_tables(bppp_rp *rp) {
  const bppp_rps::Setup &st = rp->st;
  bppp_ctx *ctx = rp->ctx;
  bppp_brp_tabs *t = new bppp_brp_tabs();
  rp->btabs = t;
  t->D = BrpDims{(uint32_t)st.nlen, (uint32_t)st.nlive, (uint32_t)st.rds.size(), st.conserve ? 1u : 0u, st.flavour ? 1u : 0u};
  std::vector<uint32_t> pr, fl;
  std::vector<uint64_t> pc, mins;
  for (const bppp_rps::Pos &p : st.pos) { pr.push_back(p.range); for (int i = 0; i < 4; i++) pc.push_back(p.coeff.w[i]); }
  for (const bppp_rps::RangeData &rd : st.rds) {
    fl.push_back((rd.output ? 1u : 0u) | (rd.assumed ? 2u : 0u));
    const U256 m = bppp_rps::s_mod_n(rd.lo);
    for (int i = 0; i < 4; i++) mins.push_back(m.w[i]);
  }
  if (pr.empty()) { pr.push_back(0); pc.assign(4, 0); }
  auto up = [&](uint32_t **dst, const void *src, size_t bytes) -> int {
    BPPP_HIP(ctx, hipMalloc(dst, bytes));
    BPPP_HIP(ctx, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return BPPP_OK;
  };
  int rc;
  if ((rc = up(&t->pos_range, pr.data(), pr.size() * 4)) || (rc = up(&t->pos_coeff, pc.data(), pc.size() * 8)) || (rc = up(&t->range_min, mins.data(), mins.size() * 8)) ||
      (rc = up(&t->range_flags, fl.data(), fl.size() * 4)) || (rc = up(&t->net_public, st.net_public.w, 32)))
    return rc;
  return BPPP_OK;
}
namespace bppp {
int brp_public_device(bppp_rp *rp, size_t batch, const uint32_t *ch, uint32_t *q, uint32_t *sp, uint32_t *pub_norm, uint32_t *pub_lin_c, uint32_t *init_sc) {
  bppp_ctx *ctx = rp->ctx;
  const bppp_brp_tabs *t = rp->btabs;
  // lanes per proof: a lane pays ~60 multiplications for its first powers of q0 and q0^-1 whatever its share of the positions, and 1024 proofs at one
  // wavefront each leave the SIMDs at one wavefront (VALU-busy 0.44 at the 64 x 64-bit shape, profiles/r04_pmc_binary_verify_per_kernel.csv): long
  // norm vectors take 2 or 4 wavefronts per proof while the batch does not fill the chip by itself
  unsigned nt = 64;
  // (measured, 64 x 64-bit shape: 1024 proofs 2.43 / 2.15 / 2.25 ms per verify call at 64 / 128 / 256 lanes, 4096 proofs 4.90 / 5.23 / 5.38 ms)
  while (nt < 256 && (size_t)t->D.nlen >= (size_t)nt * 16 && batch * (nt / 64) < 2048) nt <<= 1;
  const size_t lds = ((size_t)t->D.nr + nt) * 32;
  if (lds > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_brp_public, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_brp_public<<<dim3((unsigned)batch), dim3(nt), lds, ctx->stream>>>(t->D, (uint32_t)batch, t->pos_range, t->pos_coeff, t->range_min, t->range_flags, t->net_public, ch, q, sp,
                                                                     pub_norm, pub_lin_c, init_sc);
  BPPP_HIP(ctx, hipGetLastError());
  return BPPP_OK;
}
}  // namespace bppp

void RpOptions::from_env() {
  auto num = [](const char *name, size_t &dst) { if (const char *e = getenv(name)) dst = (size_t)atol(e); };
  num("BPPP_RP_COMB_MIN", comb_min); num("BPPP_RP_SPLIT_MIN", split_min); num("BPPP_RP_SPLIT_MIN_BINARY", split_min_binary); num("BPPP_RP_HASH_FORK_MAX", hash_fork_max);
  if (const char *e = getenv("BPPP_RP_COMB_GB")) comb_budget = (size_t)std::max(1, atoi(e)) << 30;
  if (const char *e = getenv("BPPP_RP_COMB_BITS")) comb_bits = atoi(e);
  if (const char *e = getenv("BPPP_RP_HOST_ORACLE_MAX")) host_oracle_verify = host_oracle_prove = (size_t)atol(e);
  no_comb = getenv("BPPP_RP_NO_COMB") != nullptr; no_split = getenv("BPPP_RP_NO_SPLIT") != nullptr;
  fold_points = getenv("BPPP_NLB_FOLD_POINTS") != nullptr; host_algebra = getenv("BPPP_RP_HOST_ALGEBRA") != nullptr;
  timing = getenv("BPPP_RP_TIMING") != nullptr;
}

extern "C" {

void bppp_rp_destroy(bppp_rp *rp) {
  if (!rp) return;
  bppp_ctx *ctx = rp->ctx;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (rp->tabs) bppp_trrp_destroy(rp->tabs);
  brp_tabs_destroy(rp->btabs);
  if (rp->d_basis) hipFree(rp->d_basis);
  if (rp->d_plan) hipFree(rp->d_plan);
  if (rp->work) hipFree(rp->work);
  if (rp->stage) hipFree(rp->stage);
  if (rp->hflag) hipHostFree(rp->hflag);
  if (rp->hstage) hipHostFree(rp->hstage);
  for (auto &e : rp->slice_ev) if (e) hipEventDestroy(e);
  delete rp->pool;
  if (rp->d_fixed) hipFree(rp->d_fixed);
  if (rp->commit_basis) bppp_basis_destroy(rp->commit_basis);
  if (rp->pwork) hipFree(rp->pwork);
  if (rp->awork) hipFree(rp->awork);
  if (rp->hpin) hipHostFree(rp->hpin);
  if (rp->twin) bppp_rp_destroy(rp->twin);
  if (rp->d_comb_out) hipFree(rp->d_comb_out);
  if (rp->comb && rp->comb_owned) bppp::comb_destroy(rp->comb);
  if (rp->twin_ctx) bppp_ctx_destroy(rp->twin_ctx);
  delete rp;
  ctx_release(ctx);
}

int bppp_rp_create(bppp_ctx *ctx, int flavour, int has_types, const bppp_rp_range *ranges, size_t nranges, const bppp_rp_public *pubs, size_t npub,
                   const uint64_t *points_xy, size_t npoints, const char *oracle_tag, bppp_rp **out) {
  if (!ctx || !out || ctx_closed(ctx)) return BPPP_ERR_ARG;
  *out = nullptr;
  if (!ranges || !nranges || (npub && !pubs) || !points_xy) return fail(ctx, BPPP_ERR_ARG, "rp_create: null argument");
  if (flavour != 0 && flavour != 1) return fail(ctx, BPPP_ERR_ARG, "rp_create: flavour must be 0 (norm-linear argument) or 1 (inner-product argument)");
  if (nranges >= (1u << 20) || npub >= (1u << 20)) return fail(ctx, BPPP_ERR_ARG, "rp_create: too many ranges");
  std::vector<bppp_rps::RangeData> rds(nranges);
  std::string err;
  for (size_t i = 0; i < nranges; i++) {
    const bppp_rp_range &r = ranges[i];
    if (!bppp_rps::make_range_data(r.base, U256::load(r.min), U256::load(r.max), (r.flags & BPPP_RP_SHARED) != 0, (r.flags & BPPP_RP_OUTPUT) != 0,
                                  (r.flags & BPPP_RP_ASSUMED) != 0, rds[i], err))
      return fail(ctx, BPPP_ERR_ARG, "rp_create: range " + std::to_string(i) + ": " + err);
  }
  std::vector<bppp_rps::PublicVT> pv(npub);
  for (size_t i = 0; i < npub; i++) {
    if (!bppp_host::scalars_canonical(pubs[i].type, 1) || !bppp_host::scalars_canonical(pubs[i].amount, 1))
      return fail(ctx, BPPP_ERR_ARG, "rp_create: public type / amount not canonical");
    pv[i] = bppp_rps::PublicVT{pubs[i].is_output != 0, U256::load(pubs[i].type), U256::load(pubs[i].amount)};
  }
  bppp_rp *rp = new bppp_rp();
  rp->ctx = ctx; ctx_retain(ctx);
  rp->opt.from_env();
  auto fill = [&]() -> int {
    if (!bppp_rps::make_setup(has_types != 0, rds, pv, rp->st, err, flavour)) return fail(ctx, BPPP_ERR_ARG, "rp_create: " + err);
    const bppp_rps::Setup &st = rp->st;
    // points = h : g : hs (linLen) ++ gs (nrmLen)   (TypedReciprocal.hs:334, :348-349); h is not used by the proof
    if (npoints < 2 + st.llen + st.nlen) return fail(ctx, BPPP_ERR_ARG, "rp_create: not enough basis points (need 2 + linLen + nrmLen)");
    if (!bppp_host::points_on_curve(points_xy, 2 + st.llen + st.nlen)) return fail(ctx, BPPP_ERR_POINT, "rp_create: a basis point is not on the curve");
    rp->h_g.assign(points_xy + 8, points_xy + 16);
    rp->h_H.assign(points_xy + 16, points_xy + 16 + 8 * st.llen);
    rp->h_G.assign(points_xy + 16 + 8 * st.llen, points_xy + 16 + 8 * (st.llen + st.nlen));
    rp->tag = oracle_tag ? oracle_tag : "";
    rp->c_ranges.assign(ranges, ranges + nranges); if (npub) rp->c_pubs.assign(pubs, pubs + npub);
    rp->c_points.assign(points_xy, points_xy + 8 * (2 + st.llen + st.nlen)); rp->c_has_types = has_types;
    hipSetDevice(ctx->device);
    BPPP_HIP(ctx, hipMalloc(&rp->d_basis, (1 + st.nlen + st.llen) * 64));
    BPPP_HIP(ctx, hipHostMalloc((void **)&rp->hflag, 64, hipHostMallocDefault));
    BPPP_HIP(ctx, hipMemcpy(rp->d_basis, rp->h_g.data(), 64, hipMemcpyHostToDevice));
    BPPP_HIP(ctx, hipMemcpy(rp->d_basis + 16, rp->h_H.data(), st.llen * 64, hipMemcpyHostToDevice));
    BPPP_HIP(ctx, hipMemcpy(rp->d_basis + 16 * (1 + st.llen), rp->h_G.data(), st.nlen * 64, hipMemcpyHostToDevice));
    RpDims &D = rp->D;
    D.nr = (uint32_t)st.rds.size(); D.k = (uint32_t)st.rounds; D.fn = (uint32_t)st.fn; D.fl = (uint32_t)st.fl; D.nrp = 4; D.nch = 7;
    D.coms_bytes = (D.nr + 7) / 8 + 32 * D.nr;
    const uint32_t npp = 4 + 2 * D.k;
    D.proof_bytes = 32 * (D.fn + D.fl) + (npp + 7) / 8 + 32 * npp;
    D.text_stride = ((rp_npts(D) * 2 * 78 + 15) & ~15u) + 16;      // 78 = decimal digits of 2^256, +16: aligned over-read of the last word
    int rc = rp_build_tables(rp); if (rc) return rc;
    return rp_build_plan(rp);
  };
  if (int rc = fill()) { bppp_rp_destroy(rp); return rc; }
  *out = rp;
  return BPPP_OK;
}

// host-only: the shape a setup would have (no context, no GPU) — setup's arithmetic of TypedReciprocal.hs:332-359 alone
int bppp_rp_shape_of(int flavour, int has_types, const bppp_rp_range *ranges, size_t nranges, bppp_rp_shape *out) {
  if (!ranges || !nranges || !out || (flavour != 0 && flavour != 1) || nranges >= (1u << 20)) return BPPP_ERR_ARG;
  std::vector<bppp_rps::RangeData> rds(nranges);
  std::string err;
  for (size_t i = 0; i < nranges; i++) {
    const bppp_rp_range &r = ranges[i];
    if (!bppp_rps::make_range_data(r.base, U256::load(r.min), U256::load(r.max), (r.flags & BPPP_RP_SHARED) != 0, (r.flags & BPPP_RP_OUTPUT) != 0,
                                   (r.flags & BPPP_RP_ASSUMED) != 0, rds[i], err))
      return BPPP_ERR_ARG;
  }
  bppp_rps::Setup st;
  if (!bppp_rps::make_setup(has_types != 0, rds, std::vector<bppp_rps::PublicVT>(), st, err, flavour)) return BPPP_ERR_ARG;
  const size_t nr = nranges, k = st.rounds, npp = 4 + 2 * k;
  out->nranges = nr; out->norm_len = st.nlen; out->lin_len = st.llen; out->rounds = k; out->final_norm = st.fn; out->final_lin = st.fl;
  out->coms_bytes = (nr + 7) / 8 + 32 * nr;
  out->proof_bytes = 32 * (st.fn + st.fl) + (npp + 7) / 8 + 32 * npp;
  out->challenges_per_proof = 7 + k;
  return BPPP_OK;
}

// host-only: the digits and multiplicity bookkeeping are integer work; this is `digits` (TypedReciprocal.hs:125-127) for one value of
// one range: out_digits receives *ndigits entries (capacity `cap`); BPPP_ERR_ARG if the value is outside [min, max)
int bppp_rp_digits(const bppp_rp_range *range, const uint64_t amount[4], uint32_t *out_digits, size_t cap, size_t *ndigits, int *has_bit) {
  if (!range || !amount || !out_digits || !ndigits) return BPPP_ERR_ARG;
  bppp_rps::RangeData rd;
  std::string err;
  if (!bppp_rps::make_range_data(range->base, U256::load(range->min), U256::load(range->max), false, false, false, rd, err)) return BPPP_ERR_ARG;
  const U256 v = U256::load(amount);
  if (bppp_rps::s_lt(v, rd.lo) || !bppp_rps::s_lt(v, rd.hi)) return BPPP_ERR_ARG;
  const std::vector<uint32_t> ds = bppp_rps::digits(rd, bppp_rps::u_sub(v, rd.lo));
  if (ds.size() > cap) return BPPP_ERR_ARG;
  for (size_t i = 0; i < ds.size(); i++) out_digits[i] = ds[i];
  *ndigits = ds.size();
  if (has_bit) *has_bit = rd.has_bit ? 1 : 0;
  return BPPP_OK;
}

// host-only: the CLI's hash (app/Main.hs:64-65) as the prover and the oracle use it: decode (SHA-256 (data)) through Binary (Prime p)
// into the scalar field; hashToScalar p s = bppp_hash_to_scalar (p <> s) (app/Main.hs:83-84)
int bppp_hash_to_scalar(const uint8_t *data, size_t len, uint64_t out[4]) {
  if ((len && !data) || !out) return BPPP_ERR_ARG;
  Sha256 h;
  h.update(data, len);
  uint32_t d[8], v[8];
  h.finish(d);
  sha256_digest_to_limbs(d, v);
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = ((uint64_t)v[2 * i + 1] << 32) | v[2 * i];
  bppp_rps::u_mod_n(r).store(out);
  return BPPP_OK;
}

int bppp_rp_set_option(bppp_rp *rp, int option, uint64_t value) {
  if (!rp) return BPPP_ERR_ARG;
  RpOptions &o = rp->opt;
  switch (option) {
    case BPPP_RP_OPT_COMB_MIN: o.comb_min = (size_t)value; break;
    case BPPP_RP_OPT_COMB_BUDGET: o.comb_budget = (size_t)value; o.no_comb = value == 0; if (value) rp->comb_failed = false; break;
    case BPPP_RP_OPT_COMB_BITS: if (value && (value < 4 || value > 18)) return fail(rp->ctx, BPPP_ERR_ARG, "rp_set_option: comb window must be 0 or in [4,18]");
                                o.comb_bits = (int)value; rp->comb_failed = false; break;
    case BPPP_RP_OPT_SPLIT_MIN: o.split_min = (size_t)value; o.no_split = value == 0; break;
    case BPPP_RP_OPT_HOST_ORACLE_MAX: if (value == UINT64_MAX) { o.host_oracle_verify = RpOptions().host_oracle_verify; o.host_oracle_prove = RpOptions().host_oracle_prove; }
                                      else o.host_oracle_verify = o.host_oracle_prove = (size_t)value;
                                      break;
    case BPPP_RP_OPT_FOLD_POINTS: o.fold_points = value != 0; break;
    case BPPP_RP_OPT_HOST_ALGEBRA: o.host_algebra = value != 0; break;
    case BPPP_RP_OPT_TIMING: o.timing = value != 0; break;
    default: return fail(rp->ctx, BPPP_ERR_ARG, "rp_set_option: unknown option");
  }
  if (rp->twin) rp->twin->opt = o;
  return BPPP_OK;
}

int bppp_rp_info(const bppp_rp *rp, bppp_rp_shape *out) {
  if (!rp || !out) return BPPP_ERR_ARG;
  const bppp_rps::Setup &st = rp->st;
  out->nranges = st.rds.size(); out->norm_len = st.nlen; out->lin_len = st.llen; out->rounds = st.rounds;
  out->final_norm = st.fn; out->final_lin = st.fl; out->coms_bytes = rp->D.coms_bytes; out->proof_bytes = rp->D.proof_bytes;
  out->challenges_per_proof = rp->D.nch + st.rounds;
  return BPPP_OK;
}

// RangeProof.Binary behind the same handle: setupBRP (src/RangeProof/Binary.hs:143-156).  points = [h, g, h0, h1] ++ gs (:147-148)
int bppp_rp_create_binary(bppp_ctx *ctx, int flavour, int conserve, const bppp_rp_range *ranges, size_t nranges, const uint64_t net_public[4],
                          const uint64_t *points_xy, size_t npoints, const char *oracle_tag, bppp_rp **out) {
  if (!ctx || !out || ctx_closed(ctx)) return BPPP_ERR_ARG;
  *out = nullptr;
  if (!ranges || !nranges || !net_public || !points_xy) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: null argument");
  if (flavour != 0 && flavour != 1) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: flavour must be 0 (norm-linear argument) or 1 (inner-product argument)");
  if (nranges > 1024) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: at most 1024 ranges");
  std::vector<bppp_rps::RangeData> rds(nranges);
  std::string err;
  for (size_t i = 0; i < nranges; i++) {
    const bppp_rp_range &r = ranges[i];
    if (r.base != 2 || (r.flags & BPPP_RP_SHARED)) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: range " + std::to_string(i) + ": base must be 2 and digits are not shared (app/Parse.hs:141-146)");
    if (!bppp_rps::make_range_data_binary(U256::load(r.min), U256::load(r.max), (r.flags & BPPP_RP_OUTPUT) != 0, (r.flags & BPPP_RP_ASSUMED) != 0, rds[i], err))
      return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: range " + std::to_string(i) + ": " + err);
  }
  bppp_rp *rp = new bppp_rp();
  rp->ctx = ctx; ctx_retain(ctx);
  rp->opt.from_env();
  auto fill = [&]() -> int {
    if (!bppp_rps::make_setup_binary(conserve != 0, rds, bppp_rps::s_mod_n(U256::load(net_public)), flavour, rp->st, err)) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: " + err);
    const bppp_rps::Setup &st = rp->st;
    if (npoints < 4 + st.nlen) return fail(ctx, BPPP_ERR_ARG, "rp_create_binary: not enough basis points (need 4 + nrmLen)");
    if (!bppp_host::points_on_curve(points_xy, 4 + st.nlen)) return fail(ctx, BPPP_ERR_POINT, "rp_create_binary: a basis point is not on the curve");
    rp->h_g.assign(points_xy + 8, points_xy + 16);
    rp->h_H.assign(points_xy + 16, points_xy + 32);
    rp->h_G.assign(points_xy + 32, points_xy + 32 + 8 * st.nlen);
    rp->tag = oracle_tag ? oracle_tag : "";
    rp->c_ranges.assign(ranges, ranges + nranges); rp->c_points.assign(points_xy, points_xy + 8 * (4 + st.nlen));
    rp->c_conserve = conserve; memcpy(rp->c_net_public, net_public, 32);
    hipSetDevice(ctx->device);
    BPPP_HIP(ctx, hipMalloc(&rp->d_basis, (3 + st.nlen) * 64));
    BPPP_HIP(ctx, hipHostMalloc((void **)&rp->hflag, 64, hipHostMallocDefault));
    BPPP_HIP(ctx, hipMemcpy(rp->d_basis, points_xy + 8, (3 + st.nlen) * 64, hipMemcpyHostToDevice));      // [g | h0 h1 | G]: commitRPW's term order
    RpDims &D = rp->D;
    D.nr = (uint32_t)st.rds.size(); D.k = (uint32_t)st.rounds; D.fn = (uint32_t)st.fn; D.fl = (uint32_t)st.fl; D.nrp = 2; D.nch = 4;
    D.coms_bytes = (D.nr + 7) / 8 + 32 * D.nr;
    const uint32_t npp = 2 + 2 * D.k;
    D.proof_bytes = 32 * (D.fn + D.fl) + (npp + 7) / 8 + 32 * npp;
    D.text_stride = ((rp_npts(D) * 2 * 78 + 15) & ~15u) + 16;
    int rc = brp_build_tables(rp); if (rc) return rc;
    return rp_build_plan(rp);
  };
  if (int rc = fill()) { bppp_rp_destroy(rp); return rc; }
  *out = rp;
  return BPPP_OK;
}

// grow-only device workspace of the verifier; returns the carved pointers through `cv`
static int rp_ensure_work(bppp_rp *rp, size_t bytes) {
  if (bytes <= rp->work_bytes) return BPPP_OK;
  bppp_ctx *ctx = rp->ctx;
  BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (rp->work) BPPP_HIP(ctx, hipFree(rp->work));
  rp->work = nullptr; rp->work_bytes = 0;
  BPPP_HIP(ctx, hipMalloc(&rp->work, bytes + bytes / 8));
  rp->work_bytes = bytes + bytes / 8;
  return BPPP_OK;
}

int bppp_nl_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit, const void *d_g_xy,
                                const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_q, const void *d_sp, const void *d_pub_norm,
                                const void *d_pub_lin_c, const void *d_pub_lin_x, const void *d_es, const void *d_wit_norm, const void *d_wit_lin,
                                const void *d_init_scalars, const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]);
int bppp_ip_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit, const void *d_g_xy,
                                const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_r, const void *d_sp, const void *d_pub_norm,
                                const void *d_pub_lin_c, const void *d_pub_lin_x, const void *d_es, const void *d_wit_norm, const void *d_wit_lin,
                                const void *d_init_scalars, const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]);

}  // extern "C"
// a second handle of the same setup on its own context (stream, workspaces): the two halves of a large batch run side by side
int rp_ensure_twin(bppp_rp *rp) {
  if (rp->twin) return BPPP_OK;
  bppp_ctx *ctx = rp->ctx;
  int rc = bppp_ctx_create(ctx->device, &rp->twin_ctx);
  if (rc) { rp->twin_ctx = nullptr; return fail(ctx, rc, "rp: creating the second context failed"); }
  if (rp->st.kind == 1)
    rc = bppp_rp_create_binary(rp->twin_ctx, rp->st.flavour, rp->c_conserve, rp->c_ranges.data(), rp->c_ranges.size(), rp->c_net_public, rp->c_points.data(),
                               rp->c_points.size() / 8, rp->tag.c_str(), &rp->twin);
  else
    rc = bppp_rp_create(rp->twin_ctx, rp->st.flavour, rp->c_has_types, rp->c_ranges.data(), rp->c_ranges.size(), rp->c_pubs.empty() ? nullptr : rp->c_pubs.data(), rp->c_pubs.size(),
                        rp->c_points.data(), rp->c_points.size() / 8, rp->tag.c_str(), &rp->twin);
  if (rc) {
    const std::string m = bppp_last_error(rp->twin_ctx);
    bppp_ctx_destroy(rp->twin_ctx); rp->twin_ctx = nullptr; rp->twin = nullptr;
    return fail(ctx, rc, "rp: second handle: " + m);
  }
  rp->twin->is_twin = true;
  rp->twin->opt = rp->opt;
  return BPPP_OK;
}

// ---- the verifier's oracle on the HOST, for a handful of proofs: one GPU lane walks the ~160 SHA-256 blocks of a 64by64 transcript
// in ~0.8 ms whatever the batch size; a host core (SHA extensions) hashes the same 11 KB in ~10 us.  The text itself is the device's
// (k_rp_text_lds, downloaded: an undecodable point reads as "00"); same 7 + k challenges as k_rp_hash.
namespace {
U256 host_digest_to_fr(const uint32_t h[8]) {
  uint32_t v[8];
  sha256_digest_to_limbs(h, v);
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = ((uint64_t)v[2 * i + 1] << 32) | v[2 * i];
  return bppp_rps::u_mod_n(r);
}
// text / off: the proof's transcript text and the offsets of its points as k_rp_text_lds wrote them (downloaded); ch_out [7][4], es_out [k][4]
// (the weight rho_b is k_rp_rho's on both routes).  part 0: the seven challenges of verifyTRRPM; part 1: the k round challenges of verifyBPM
void host_verifier_oracle(const bppp_rp *rp, const uint8_t *text, const uint32_t *off, uint64_t *ch_out, uint64_t *es_out, int part) {
  const uint32_t k = rp->D.k, nr = rp->D.nr, nrp = rp->D.nrp, npts = 2 * k + nrp + nr;
  auto one = [&](uint32_t n, uint32_t count, uint32_t start, uint64_t *out) {
    Sha256 h;
    const std::string hdr = rp->tag + std::to_string(n) + std::to_string(count);
    h.update(hdr.data(), hdr.size());
    h.update(text + off[start], off[npts] - off[start]);
    uint32_t d[8];
    h.finish(d);
    host_digest_to_fr(d).store(out);
  };
  if (part == 0 && rp->st.kind == 0) {
    for (uint32_t n = 1; n <= 3; n++) one(n, 2 + nr, 2 * k + 2, ch_out + 4 * (n - 1));          // e, x, r0        (the order of rp_build_plan)
    for (uint32_t n = 1; n <= 3; n++) one(n, 3 + nr, 2 * k + 1, ch_out + 4 * (3 + n - 1));      // q, x', r1
    one(1, 4 + nr, 2 * k, ch_out + 4 * 6);                                                      // t
  } else if (part == 0) {
    for (uint32_t n = 1; n <= 3; n++) one(n, 1 + nr, 2 * k + 1, ch_out + 4 * (n - 1));          // Binary: q, x, r
    one(1, 2 + nr, 2 * k, ch_out + 4 * 6);                                                      // t
  } else {
    for (uint32_t j = 1; j <= k; j++) one(1, nrp + nr + 2 * j, 2 * (k - j), es_out + 4 * (k - j));   // round j: es is LAST round first
  }
}
}  // namespace
extern "C" {

int bppp_rp_verify_batch_device(bppp_rp *rp, size_t batch, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32], int *accept,
                                uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy) {
  return bppp_rp_verify_shard_device(rp, batch, 0, d_coms_files, d_proof_files, seed, accept, proof_status, challenges_out, combined_xy);
}

static int rp_verify_shard_run(bppp_rp *rp, size_t batch, uint64_t index_offset, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32],
                               int *accept, uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy);
int bppp_rp_verify_shard_device(bppp_rp *rp, size_t batch, uint64_t index_offset, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32],
                                int *accept, uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy) {
  if (!rp || !accept) return BPPP_ERR_ARG;
  const int rc = rp_verify_shard_run(rp, batch, index_offset, d_coms_files, d_proof_files, seed, accept, proof_status, challenges_out, combined_xy);
  // A failed call may leave work queued on either stream that still reads the caller's buffers (the sliced uploads from host files run on
  // the context's second stream): nothing of this call is in flight once it has returned, whatever the outcome.
  if (rc && rp->ctx && !ctx_closed(rp->ctx)) {
    hipStreamSynchronize(rp->ctx->stream);
    if (rp->ctx->aux_stream) hipStreamSynchronize(rp->ctx->aux_stream);
    (void)hipGetLastError();
  }
  return rc;
}
static int rp_verify_shard_run(bppp_rp *rp, size_t batch, uint64_t index_offset, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32],
                               int *accept, uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy) {
  bppp_ctx *ctx = rp->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  *accept = 0;
  if (combined_xy) memset(combined_xy, 0, 64);
  if (!batch) { *accept = 1; return BPPP_OK; }
  if (!d_coms_files || !d_proof_files || !seed || batch >= (1u << 22)) return fail(ctx, BPPP_ERR_ARG, "rp_verify_batch: bad arguments");
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  const bppp_rps::Setup &S = rp->st;
  const RpDims D = rp->D;
  const size_t B = batch, nlen = S.nlen, llen = S.llen, k = S.rounds, ninit = D.nrp + D.nr, npts = rp_npts(D);
  size_t need = 0;
  uint32_t *init_pts = nullptr, *resp_pts = nullptr, *wit_norm = nullptr, *wit_lin = nullptr, *text_off = nullptr, *ch = nullptr, *es = nullptr, *rho = nullptr,
           *q = nullptr, *sp = nullptr, *pub_norm = nullptr, *pub_lin_c = nullptr, *pub_lin_x = nullptr, *init_sc = nullptr, *bad = nullptr;
  uint8_t *text = nullptr, *d_seed = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    Carver cv(pass ? rp->work : nullptr, rp->work_bytes);
    init_pts = cv.take<uint32_t>(B * ninit * 16); resp_pts = cv.take<uint32_t>(B * 2 * k * 16 + 16);
    wit_norm = cv.take<uint32_t>(B * D.fn * 8 + 8); wit_lin = cv.take<uint32_t>(B * D.fl * 8 + 8);
    text = cv.take<uint8_t>(B * (size_t)D.text_stride + 64); text_off = cv.take<uint32_t>(B * (npts + 1));
    ch = cv.take<uint32_t>(B * 7 * 8); es = cv.take<uint32_t>(B * k * 8 + 8); rho = cv.take<uint32_t>(B * 8);
    q = cv.take<uint32_t>(B * 8); sp = cv.take<uint32_t>(B * 8); pub_norm = cv.take<uint32_t>(B * nlen * 8); pub_lin_c = cv.take<uint32_t>(B * llen * 8);
    pub_lin_x = cv.take<uint32_t>(B * llen * 8); init_sc = cv.take<uint32_t>(B * ninit * 8); bad = cv.take<uint32_t>(B + 1);
    d_seed = cv.take<uint8_t>(32);
    if (!pass) { need = cv.off; int rc = rp_ensure_work(rp, need); if (rc) return rc; }
  }
  BPPP_HIP(ctx, hipMemsetAsync(bad, 0, (B + 1) * 4, st));
  BPPP_HIP(ctx, hipMemsetAsync(pub_lin_x, 0, B * llen * 32, st));       // the public linear vector of these proofs is zero (TypedReciprocal.hs:466)
  BPPP_HIP(ctx, hipMemcpyAsync(d_seed, seed, 32, hipMemcpyHostToDevice, st));
  // decodeProof.  From host buffers (bppp_rp_verify_batch left them in rp->host_*; the device pointers are then its staging area): the
  // files go up in four slices, each followed by its decode launches, so the square roots of slice i run under the upload of slice
  // i + 1 (a pageable copy keeps the HOST busy staging, not the stream)
  {
    const uint8_t *hc = rp->host_coms, *hp = rp->host_proofs;
    rp->host_coms = rp->host_proofs = nullptr;
    // how many: a slice's decode launch is latency-bound (one square-root chain, ~0.18 ms for 1024 proofs against 0.53 ms for all 4096),
    // so slices cost decode time; pageable files arrive at the host's staging rate (0.7 ms per 4096 proofs) and four slices hide most of
    // it, page-locked ones (bppp_host_alloc) arrive in 0.25 ms and two are enough
    size_t nslices = 1;
    if (hc && B >= 1024) {
      hipPointerAttribute_t at;
      const bool pinned = hipPointerGetAttributes(&at, hc) == hipSuccess && at.type == hipMemoryTypeHost;
      if (!pinned) (void)hipGetLastError();                 // an ordinary pointer is "invalid value" to the runtime: not an error of this call
      nslices = pinned ? 2 : 4;
    }
    for (size_t sl = 0; sl < nslices; sl++) {
      const size_t b0 = B * sl / nslices, b1 = B * (sl + 1) / nslices, nb = b1 - b0;
      const uint8_t *dc = (const uint8_t *)d_coms_files + b0 * (size_t)D.coms_bytes, *dp = (const uint8_t *)d_proof_files + b0 * (size_t)D.proof_bytes;
      if (hc) {
        // the copies go on the context's second stream (a copy on `st` would queue behind the previous slice's kernels), the kernels wait
        // for their slice's event
        hipStream_t up = st;
        if (nslices > 1) {
          int rca = ctx_aux(ctx); if (rca) return rca;
          up = ctx->aux_stream;
          if (!rp->slice_ev[sl]) BPPP_HIP(ctx, hipEventCreateWithFlags(&rp->slice_ev[sl], hipEventDisableTiming));
          if (sl == 0) { BPPP_HIP(ctx, hipEventRecord(ctx->aux_fork, st)); BPPP_HIP(ctx, hipStreamWaitEvent(up, ctx->aux_fork, 0)); }   // the staging area is free again
        }
        BPPP_HIP(ctx, hipMemcpyAsync((void *)dc, hc + b0 * (size_t)D.coms_bytes, nb * (size_t)D.coms_bytes, hipMemcpyHostToDevice, up));
        BPPP_HIP(ctx, hipMemcpyAsync((void *)dp, hp + b0 * (size_t)D.proof_bytes, nb * (size_t)D.proof_bytes, hipMemcpyHostToDevice, up));
        if (nslices > 1) { BPPP_HIP(ctx, hipEventRecord(rp->slice_ev[sl], up)); BPPP_HIP(ctx, hipStreamWaitEvent(st, rp->slice_ev[sl], 0)); }
      }
      const uint64_t np = (uint64_t)nb * npts, ns = (uint64_t)nb * (D.fn + D.fl);
      k_rp_decode_points<<<dim3((unsigned)((np + 63) / 64)), dim3(64), 0, st>>>(D, (uint32_t)nb, dc, dp, init_pts + b0 * ninit * 16, resp_pts + b0 * 2 * k * 16, bad + b0, bad + B);
      if (ns) k_rp_decode_scalars<<<dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, st>>>(D, (uint32_t)nb, dp, wit_norm + b0 * D.fn * 8, wit_lin + b0 * D.fl * 8);
    }
  }
  BPPP_HIP(ctx, hipMemcpyAsync(rp->hflag, bad + B, 4, hipMemcpyDeviceToHost, st));     // pinned; read after the MSM has drained the stream
  const size_t host_oracle_max = rp->opt.host_oracle_verify;
  // async copies below target host vectors: whatever path leaves this function, the stream is drained before they are destroyed
  struct StreamDrain { hipStream_t s; ~StreamDrain() { hipStreamSynchronize(s); } };
  if (B <= host_oracle_max) {
    // the transcript text comes from the device (k_rp_text_lds: ~10 us against ~0.45 us per coordinate on a host core, 168 of them for 64by64)
    const size_t tbytes = (B * (size_t)D.text_stride + 63) & ~(size_t)63, obytes = (B * (npts + 1) * 4 + 63) & ~(size_t)63, n_hch = B * 28, n_hes = B * k * 4 + 4;
    const size_t hbytes = tbytes + obytes + (n_hch + n_hes) * 8;
    if (hbytes > rp->hstage_bytes) {
      if (rp->hstage) { BPPP_HIP(ctx, hipStreamSynchronize(st)); hipHostFree(rp->hstage); rp->hstage = nullptr; rp->hstage_bytes = 0; }
      BPPP_HIP(ctx, hipHostMalloc((void **)&rp->hstage, hbytes, hipHostMallocDefault));
      rp->hstage_bytes = hbytes;
    }
    uint8_t *htext = (uint8_t *)rp->hstage;
    uint32_t *hoff = (uint32_t *)(htext + tbytes);
    uint64_t *hch = (uint64_t *)(htext + tbytes + obytes), *hes = hch + n_hch;
    const bool timing = rp->opt.timing;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = timing ? now() : 0;
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[rp_verify] %-24s %7.1f us\n", what, t - t_last); t_last = t; } };
    if (rp_text_lds_bytes(D) <= 64 * 1024) k_rp_text_lds<<<dim3((unsigned)B), dim3(256), rp_text_lds_bytes(D), st>>>(D, init_pts, resp_pts, text, text_off);
    else k_rp_text<<<dim3((unsigned)B), dim3(256), (npts + 1) * 4, st>>>(D, init_pts, resp_pts, text, text_off);
    BPPP_HIP(ctx, hipMemcpyAsync(hoff, text_off, B * (npts + 1) * 4, hipMemcpyDeviceToHost, st));
    BPPP_HIP(ctx, hipMemcpyAsync(htext, text, B * (size_t)D.text_stride, hipMemcpyDeviceToHost, st));
    BPPP_HIP(ctx, hipStreamSynchronize(st));
    lap("decode + download");
    // the seven challenges of verifyTRRPM first: k_trrp_public needs only those and runs while the host hashes the argument's rounds
    // one proof per item: the calling thread alone for one proof, else with the handle's pool (a std::thread per proof costs more than its 36 us of hashing)
    if (B > 1 && !rp->pool) rp->pool = new bppp::HostPool((unsigned)std::min<size_t>(host_oracle_max, 16) - 1);
    auto all = [&](int part) {
      const std::function<void(size_t)> f = [&](size_t b) { host_verifier_oracle(rp, htext + b * (size_t)D.text_stride, hoff + b * (npts + 1), &hch[b * 28], &hes[b * k * 4], part); };
      if (B == 1) f(0); else rp->pool->run(B, f);
    };
    all(0);
    lap("host oracle, part 0");
    BPPP_HIP(ctx, hipMemcpyAsync(ch, hch, B * 7 * 32, hipMemcpyHostToDevice, st));
    int rc0 = S.kind ? brp_public_device(rp, B, ch, q, sp, pub_norm, pub_lin_c, init_sc) : bppp_trrp_public_device(rp->tabs, B, ch, q, sp, pub_norm, pub_lin_c, init_sc);
    if (rc0) return rc0;
    lap("launch of the scalars");
    all(1);
    lap("host oracle, part 1");
    if (k) BPPP_HIP(ctx, hipMemcpyAsync(es, hes, B * k * 32, hipMemcpyHostToDevice, st));       // (pinned staging: the next call's downloads are ordered behind this copy on the stream)
  } else {
    // transcript text, then the hashing in two halves: the seven challenges of verifyTRRPM on the call's stream, followed there by
    // k_trrp_public (all it needs); the k round challenges of verifyBPM on the context's second stream, beside it — only up to
    // RpOptions::hash_fork_max proofs (default 64, rp_internal.hpp): above that the device oracle's one launch over every hash wins.
    // (Round-2 measurement with the 8 x 32-limb scalar kernels had the fork ahead up to 1024 proofs, 256 proofs 1.65 -> 1.38 ms; with the
    // 10 x 26 kernels of round 3 the crossover fell to ~64.  At 4096 proofs the three kernels already fill the VALU and side by side
    // each only stretches: hash 0.65 -> 0.55 + 0.87, scalars 0.84 -> 1.04 ms: measured, not kept.)
    if (rp_text_lds_bytes(D) <= 64 * 1024) k_rp_text_lds<<<dim3((unsigned)B), dim3(256), rp_text_lds_bytes(D), st>>>(D, init_pts, resp_pts, text, text_off);
    else k_rp_text<<<dim3((unsigned)B), dim3(256), (npts + 1) * 4, st>>>(D, init_pts, resp_pts, text, text_off);
    const uint32_t nch = D.nch, nes = rp->nhash - D.nch;
    hipStream_t aux = st;
    const bool fork = nes && B <= rp->opt.hash_fork_max;
    if (fork) {
      int rca = ctx_aux(ctx); if (rca) return rca;
      aux = ctx->aux_stream;
      BPPP_HIP(ctx, hipEventRecord(ctx->aux_fork, st));
      BPPP_HIP(ctx, hipStreamWaitEvent(aux, ctx->aux_fork, 0));
    }
    const uint32_t nfirst = fork ? nch : rp->nhash;            // one launch over every hash unless the halves run side by side
    k_rp_hash<<<dim3((unsigned)(((uint64_t)B * nfirst + 63) / 64)), dim3(128), 0, st>>>(D, (uint32_t)B, nfirst, rp->d_plan, text, text_off, ch, es);
    if (fork) {
      k_rp_hash<<<dim3((unsigned)(((uint64_t)B * nes + 63) / 64)), dim3(128), 0, aux>>>(D, (uint32_t)B, nes, rp->d_plan + nch, text, text_off, ch, es);
      BPPP_HIP(ctx, hipEventRecord(ctx->aux_join, aux));
    }
    BPPP_HIP(ctx, hipGetLastError());
    int rc0 = S.kind ? brp_public_device(rp, B, ch, q, sp, pub_norm, pub_lin_c, init_sc) : bppp_trrp_public_device(rp->tabs, B, ch, q, sp, pub_norm, pub_lin_c, init_sc);
    if (fork) BPPP_HIP(ctx, hipStreamWaitEvent(st, ctx->aux_join, 0));      // (joined even when the launch above failed: the second stream must not outlive the call's buffers)
    if (rc0) { hipStreamSynchronize(st); return rc0; }
  }
  k_rp_rho<<<dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st>>>(D, (uint32_t)B, index_offset, d_seed, ch, es, wit_norm, wit_lin, rho);
  BPPP_HIP(ctx, hipGetLastError());
  int rc = BPPP_OK;
  uint64_t out_xy[8];
  // verifyBPM of the setup's argument flavour (q is makeNorm's r for the inner-product one)
  // (every input below was made on the device by this call: canonical scalars, points on the curve or infinity — no validation pass)
  auto verify_bp = S.flavour ? bppp::ip_verify_batch_run : bppp::nl_verify_batch_run;
  rc = verify_bp(ctx, B, nlen, llen, k, D.fn, D.fl, ninit, rp->d_g(), rp->d_G(), rp->d_H(), rho, q, sp, pub_norm,
                 pub_lin_c, pub_lin_x, es, wit_norm, wit_lin, init_sc, init_pts, resp_pts, out_xy, false);
  if (rc) return rc;
  // decode failures (an x with no point on the curve): Nothing in the reference (decodeCommitments, Encoding.hs:119-128).  The batch-wide
  // flag reached pinned memory long before the MSM drained the stream; the per-proof words are fetched only when somebody needs them
  const bool any_bad = rp->hflag[0] != 0;
  std::vector<uint32_t> hbad(proof_status ? B : 0);
  StreamDrain drain{st};
  if (proof_status) BPPP_HIP(ctx, hipMemcpyAsync(hbad.data(), bad, B * 4, hipMemcpyDeviceToHost, st));
  if (challenges_out) {                        // [batch][nch + k]: the range-proof layer's challenges (Binary: q, x, r from slots 0-2, t from slot 6), then the rounds'
    const size_t nch = D.nch, row = (nch + k) * 32;
    if (S.kind == 0) BPPP_HIP(ctx, hipMemcpy2DAsync(challenges_out, row, ch, 7 * 32, 7 * 32, B, hipMemcpyDeviceToHost, st));
    else {
      BPPP_HIP(ctx, hipMemcpy2DAsync(challenges_out, row, ch, 7 * 32, 3 * 32, B, hipMemcpyDeviceToHost, st));
      BPPP_HIP(ctx, hipMemcpy2DAsync(challenges_out + 12, row, ch + 6 * 8, 7 * 32, 32, B, hipMemcpyDeviceToHost, st));
    }
    if (k) BPPP_HIP(ctx, hipMemcpy2DAsync(challenges_out + 4 * nch, row, es, k * 32, k * 32, B, hipMemcpyDeviceToHost, st));
  }
  if (proof_status || challenges_out) BPPP_HIP(ctx, hipStreamSynchronize(st));
  auto is_inf = [](const uint64_t *p) { uint64_t o = 0; for (int i = 0; i < 8; i++) o |= p[i]; return o == 0; };
  const bool whole = is_inf(out_xy);
  if (combined_xy) memcpy(combined_xy, out_xy, 64);
  *accept = (whole && !any_bad) ? 1 : 0;
  if (!proof_status) return BPPP_OK;
  for (size_t b = 0; b < B; b++) proof_status[b] = hbad[b] ? BPPP_RP_MALFORMED : BPPP_RP_VALID;
  if (whole) return BPPP_OK;
  // The combination is not the identity: find the culprits by bisection.  Every per-proof array is [batch][...], so a sub-batch
  // [lo, hi) is the same call on offset pointers (any non-zero weights do); a malformed proof decodes to infinity points and is
  // simply another failing member.  O(f log B) combined MSMs for f bad proofs.
  struct Range { size_t lo, hi; bool known_bad; };
  std::vector<Range> todo;
  todo.push_back(Range{0, B, true});
  while (!todo.empty()) {
    const Range r = todo.back(); todo.pop_back();
    bool ok = false;
    if (!r.known_bad) {
      const size_t o = r.lo, n = r.hi - r.lo;
      rc = verify_bp(ctx, n, nlen, llen, k, D.fn, D.fl, ninit, rp->d_g(), rp->d_G(), rp->d_H(), rho + o * 8, q + o * 8,
                                       sp + o * 8, pub_norm + o * nlen * 8, pub_lin_c + o * llen * 8, pub_lin_x + o * llen * 8, es + o * k * 8, wit_norm + o * D.fn * 8,
                                       wit_lin + o * D.fl * 8, init_sc + o * ninit * 8, init_pts + o * ninit * 16, resp_pts + o * 2 * k * 16, out_xy, false);
      if (rc) return rc;
      ok = is_inf(out_xy);
    }
    if (ok) continue;
    if (r.hi - r.lo == 1) { if (proof_status[r.lo] == BPPP_RP_VALID) proof_status[r.lo] = BPPP_RP_INVALID; continue; }
    const size_t mid = r.lo + (r.hi - r.lo) / 2;
    todo.push_back(Range{r.lo, mid, false}); todo.push_back(Range{mid, r.hi, false});
  }
  return BPPP_OK;
}

int bppp_rp_verify_batch(bppp_rp *rp, size_t batch, const uint8_t *coms_files, const uint8_t *proof_files, const uint8_t seed[32], int *accept,
                         uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy) {
  if (!rp || !accept) return BPPP_ERR_ARG;
  bppp_ctx *ctx = rp->ctx;
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  if (!batch) { *accept = 1; return BPPP_OK; }
  if (!coms_files || !proof_files) return fail(ctx, BPPP_ERR_ARG, "rp_verify_batch: null input");
  hipSetDevice(ctx->device);
  const size_t cb = batch * (size_t)rp->D.coms_bytes, pb = batch * (size_t)rp->D.proof_bytes;
  const size_t cbp = (cb + 255) & ~(size_t)255;
  if (cbp + pb + 256 > rp->stage_bytes) {      // a private grow-only buffer: the context's scratch serves bppp_nl_verify_batch_device
    BPPP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rp->stage) BPPP_HIP(ctx, hipFree(rp->stage));
    rp->stage = nullptr; rp->stage_bytes = 0;
    BPPP_HIP(ctx, hipMalloc(&rp->stage, cbp + pb + 256));
    rp->stage_bytes = cbp + pb + 256;
  }
  void *stage = rp->stage;
  rp->host_coms = coms_files; rp->host_proofs = proof_files;          // uploaded in slices by the decode stage of the call below
  int rc = bppp_rp_verify_batch_device(rp, batch, stage, (char *)stage + cbp, seed, accept, proof_status, challenges_out, combined_xy);
  rp->host_coms = rp->host_proofs = nullptr;
  hipStreamSynchronize(ctx->stream);
  if (ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);     // the sliced uploads read the caller's files from there
  return rc;
}

}  // extern "C"
