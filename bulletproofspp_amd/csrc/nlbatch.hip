// nlbatch.hip — batch verification of B norm-linear arguments of one shape over one shared basis.
//
// The reference has no batch verifier (only the TODO at src/RangeProof/TypedReciprocal.hs:469-472 and
// src/RangeProof.hs:103-106: "takes a random linear combination, computes the scalars in the same manner as
// the bulletproof, and then performs a single ec inner product").  Parity is defined in SURVEY.md §8(c):
//   result = sum_k rho_k * MSM(T_k),   T_k = exactly the term list verifyWith builds for proof k
//                                       (src/Bulletproof.hs:362-368, :375-377)
// so every per-proof scalar is the one bppp_nl_verify computes (expandChallenges, NormArgument.hs:73-81,
// :131-145), scaled by rho_k; scalars that sit on the SHARED basis (G, H, g) are summed over k on the device
// (the "B x 774 mat-vec" of SURVEY.md §8d), per-proof points (initCom, responses) keep their own scalars.
// One MSM of (nlen + llen + 1) + B*(ninit + 2k) terms then decides all B proofs: it is infinity iff every
// proof verifies (up to the 2^-256 soundness slack of the random combination).
#include <string.h>
#include <algorithm>
#include <vector>
#include "ctx.hpp"
#include "fe.hip.h"
#include "fr26.hip.h"
#include "modinv.hip.h"
#include "hostmath.hpp"

namespace bppp {
int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);

static constexpr int KT = 2;    // proofs per partial sum at most (more, shorter wavefronts: these kernels are latency-bound)
// batches that leave SIMDs idle anyway take ONE proof per partial sum: half the chain per lane, twice the (cheap) partial rows to add
static inline uint32_t vb_kt(size_t batch) { return batch <= 1024 ? 1u : (uint32_t)KT; }
BPPP_DI fe frm(const fe &a, const fe &b) { return fe_mul<1>(a, b); }

// factor table per proof: fac[b][r] = q_b^(2^r) (r < k), fac[b][k + r] = e_{b, first-round-first r}, and qF2 = (q^(2^k))^2
__global__ void __launch_bounds__(64) k_vb_factors(const uint32_t *__restrict__ q, const uint32_t *__restrict__ es, uint32_t batch, int k,
                                                   uint32_t *__restrict__ fac, uint32_t *__restrict__ qf2) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  fe qp = fe_load(q + (size_t)b * 8);
  for (int r = 0; r < k; r++) {
    fe_store(fac + ((size_t)b * 2 * k + r) * 8, qp);
    fe_store(fac + ((size_t)b * 2 * k + k + r) * 8, fe_load(es + ((size_t)b * k + (k - 1 - r)) * 8));   // es is last round first
    qp = fe_sqr<1>(qp);
  }
  fe_store(qf2 + (size_t)b * 8, fe_sqr<1>(qp));
}

// tensor'(vs, es, qs)[i] for one proof: vs[i >> k] * prod_r (bit r of i ? e_r : q_r)   (src/Bulletproof.hs:94-95)
BPPP_DI fe tensor_at(const uint32_t *vs, uint32_t nvs, const uint32_t *fac, int k, uint32_t i, bool use_q) {
  uint32_t hi = i >> k;
  if (hi >= nvs) return fe_zero();                 // zipWithDef' default (src/Utils.hs:182-184)
  fe acc = fe_load(vs + (size_t)hi * 8);
  for (int r = 0; r < k; r++) {
    bool bit = (i >> r) & 1u;
    if (bit) acc = fe_mul<1>(acc, fe_load(fac + (size_t)(k + r) * 8));
    else if (use_q) acc = fe_mul<1>(acc, fe_load(fac + (size_t)r * 8));
  }
  return acc;
}

// partial[kt][i] = sum_{b in tile kt} rho_b * (pub[b][i] - tensor_b[i])      (shared-basis scalars), one position per lane
__global__ void __launch_bounds__(256) k_vb_shared1(const uint32_t *__restrict__ rho, const uint32_t *__restrict__ pub, const uint32_t *__restrict__ wit,
                                                    uint32_t nvs, const uint32_t *__restrict__ fac, uint32_t batch, uint32_t len, int k, int use_q,
                                                    uint32_t ktile, uint32_t *__restrict__ partial) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, kt = blockIdx.y;
  if (i >= len) return;
  fe acc = fe_zero();
  uint32_t b0 = kt * ktile, b1 = min(batch, b0 + ktile);
  for (uint32_t b = b0; b < b1; b++) {
    fe t = tensor_at(wit + (size_t)b * nvs * 8, nvs, fac + (size_t)b * 2 * k * 8, k, i, use_q != 0);
    fe d = fe_sub<1>(fe_load(pub + ((size_t)b * len + i) * 8), t);
    acc = fe_add<1>(acc, fe_mul<1>(fe_load(rho + (size_t)b * 8), d));
  }
  fe_store(partial + ((size_t)kt * len + i) * 8, acc);
}
// Same sums with 4 consecutive basis positions i0..i0+3 per lane (k >= 2): they share the tensor factors of rounds >= 2, so the
// lane multiplies those once and expands the two low rounds in registers; rho is folded into the base value.  (8 positions per
// lane halved the multiplications again but left one wavefront per SIMD: 160 us per launch here against 240 us.)
__global__ void __launch_bounds__(64) k_vb_shared4(const uint32_t *__restrict__ rho, const uint32_t *__restrict__ pub, const uint32_t *__restrict__ wit,
                                                   uint32_t nvs, const uint32_t *__restrict__ fac, uint32_t batch, uint32_t len, int k, int use_q,
                                                   uint32_t ktile, uint32_t *__restrict__ partial) {
  // Fr in 10 x 26-bit limbs (fr26.hip.h: 413 instructions per multiplication against 785): values are canonical 8 x 32 in memory,
  // converted once on load; the accumulators stay lazily reduced (magnitude 3 per proof) until the store
  const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, kt = blockIdx.y;
  if (i0 >= len) return;
  fr a0 = fr_zero(), a1 = a0, a2 = a0, a3 = a0;          // named scalars, not arrays: arrays of field elements end up in scratch here
  uint32_t b0 = kt * ktile, b1 = min(batch, b0 + ktile);
  for (uint32_t b = b0; b < b1; b++) {
    const uint32_t *f = fac + (size_t)b * 2 * k * 8;
    const fr r = fr_load(rho + (size_t)b * 8);
    fr t0 = fr_zero(), t1 = t0, t2 = t0, t3 = t0;
    const uint32_t *pb = pub + ((size_t)b * len + i0) * 8;
    const fr p0 = fr_load(pb), p1 = i0 + 1 < len ? fr_load(pb + 8) : fr_zero(), p2 = i0 + 2 < len ? fr_load(pb + 16) : fr_zero(),
             p3 = i0 + 3 < len ? fr_load(pb + 24) : fr_zero();
    uint32_t hi = i0 >> k;
    if (hi < nvs) {                                        // zipWithDef' default 0 beyond the tensor (src/Utils.hs:182-184)
      // every load of this proof is issued before the first multiplication (the factor of round rr is e or q by the bit of i0:
      // the address is computed, not branched on), so the wavefront waits for memory once per proof instead of once per factor
      const fr w0 = fr_load(wit + ((size_t)b * nvs + hi) * 8);
      const fr e0 = fr_load(f + (size_t)k * 8), q0 = fr_load(f), e1 = fr_load(f + (size_t)(k + 1) * 8), q1 = fr_load(f + 8);
      fr base = fr_mul(r, w0);
      for (int rr = 2; rr < k; rr++) {
        const bool bit = (i0 >> rr) & 1u;
        if (bit || use_q) base = fr_mul(base, fr_load(f + (size_t)((bit ? k : 0) + rr) * 8));
      }
      t0 = base;
      t1 = fr_mul(e0, t0); if (use_q) t0 = fr_mul(q0, t0);
      t2 = fr_mul(e1, t0); t3 = fr_mul(e1, t1); if (use_q) { t0 = fr_mul(q1, t0); t1 = fr_mul(q1, t1); }
    }
#define VB_ACC(O, A, P, T) if (i0 + (O) < len) A = fr_add(A, fr_sub<1>(fr_mul(r, P), T));       /* KT = 2 proofs: magnitude <= 6 */
    VB_ACC(0, a0, p0, t0) VB_ACC(1, a1, p1, t1) VB_ACC(2, a2, p2, t2) VB_ACC(3, a3, p3, t3)
#undef VB_ACC
  }
  uint32_t *po = partial + ((size_t)kt * len + i0) * 8;
#define VB_ST(O, A) if (i0 + (O) < len) fr_store(po + (O) * 8, A);
  VB_ST(0, a0) VB_ST(1, a1) VB_ST(2, a2) VB_ST(3, a3)
#undef VB_ST
}
static_assert(KT * 3 <= 16, "k_vb_shared4: the accumulators' magnitude must stay within fr_normalize's range");

// column sums of partial[ntiles][len]: block (x, y) adds the tiles [y*per, (y+1)*per) of 64 columns (its 4 wavefronts take
// every 4th tile) into out[y][len].  Called twice: ntiles -> SUM_GROUPS rows -> 1 row; a single pass over all tiles had only
// len/64 workgroups walking 1024 dependent loads each (100-150 us for 17 MB).
static constexpr uint32_t SUM_GROUPS = 32;
__global__ void __launch_bounds__(256) k_vb_sum_partials(const uint32_t *__restrict__ partial, uint32_t ntiles, uint32_t per, uint32_t len,
                                                         uint32_t *__restrict__ out) {
  __shared__ uint32_t lds[256 * 8];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t i = blockIdx.x * 64 + lane;
  const uint32_t t0 = blockIdx.y * per, t1 = min(ntiles, t0 + per);
  fe acc = fe_zero();
  if (i < len)
    for (uint32_t t = t0 + wv; t < t1; t += 4) acc = fe_add<1>(acc, fe_load(partial + ((size_t)t * len + i) * 8));
  for (int q = 0; q < 8; q++) lds[threadIdx.x * 8 + q] = acc.v[q];
  __syncthreads();
  if (wv == 0 && i < len) {
    for (int w = 1; w < 4; w++) { fe o; for (int q = 0; q < 8; q++) o.v[q] = lds[(w * 64 + lane) * 8 + q]; acc = fe_add<1>(acc, o); }
    fe_store(out + ((size_t)blockIdx.y * len + i) * 8, acc);
  }
}

// sum_j c_j tensor'(wit_lin, es)[j] of every proof (NormArgument.hs:76-78; tensor': src/Bulletproof.hs:94-95, zero beyond it), first
// half: lane (b, g) sums the four consecutive positions 4g .. 4g + 3, which share the factors of rounds >= 2 (as k_vb_shared4): llen = 261
// is 66 lanes x 11 multiplications per proof, packed densely over the batch (a block per proof left three of its four wavefronts idle)
__global__ void __launch_bounds__(256) k_vb_lin_partial(const uint32_t *__restrict__ wit_lin, uint32_t fl, const uint32_t *__restrict__ pub_c, uint32_t llen,
                                                        const uint32_t *__restrict__ fac, int k, uint32_t batch, uint32_t G4, uint32_t *__restrict__ partial) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (uint64_t)batch * G4) return;
  const uint32_t b = (uint32_t)(idx / G4), j0 = 4u * (uint32_t)(idx % G4);
  const uint32_t *f = fac + (size_t)b * 2 * k * 8;
  fr acc = fr_zero();
  if (k >= 2) {
    const uint32_t hi = j0 >> k;
    if (hi < fl) {
      const uint32_t *pc = pub_c + ((size_t)b * llen + j0) * 8;
      fr base = fr_load(wit_lin + ((size_t)b * fl + hi) * 8);
      const fr e0 = fr_load(f + (size_t)k * 8), e1 = fr_load(f + (size_t)(k + 1) * 8);
      const fr c0 = fr_load(pc), c1 = j0 + 1 < llen ? fr_load(pc + 8) : fr_zero(), c2 = j0 + 2 < llen ? fr_load(pc + 16) : fr_zero(),
               c3 = j0 + 3 < llen ? fr_load(pc + 24) : fr_zero();
      for (int rr = 2; rr < k; rr++) if ((j0 >> rr) & 1u) base = fr_mul(base, fr_load(f + (size_t)(k + rr) * 8));
      const fr t1 = fr_mul(e0, base), t2 = fr_mul(e1, base), t3 = fr_mul(e1, t1);
      acc = fr_add(fr_add(fr_mul(c0, base), fr_mul(c1, t1)), fr_add(fr_mul(c2, t2), fr_mul(c3, t3)));       // magnitude 4
    }
  } else {
    for (uint32_t j = j0; j < min(llen, j0 + 4); j++) {     // k < 2: position by position (the tensor index differs inside the group)
      const uint32_t h2 = j >> k;
      if (h2 >= fl) continue;
      fr tl = fr_load(wit_lin + ((size_t)b * fl + h2) * 8);
      if (k == 1 && (j & 1u)) tl = fr_mul(tl, fr_load(f + (size_t)k * 8));
      acc = fr_add(acc, fr_mul(fr_load(pub_c + ((size_t)b * llen + j) * 8), tl));
    }
  }
  fr_store(partial + idx * 8, acc);
}
BPPP_DI fe fe_shfl_down(const fe &a, int d) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], d, 64);
  return r;
}
// second half, one wavefront per proof: gs[b] = rho_b * (sp_b - sc_b), sc_b = sum_i (qF^2)^(i+1) vs_i^2 + the sum above
// (NormArgument.hs:135, :76-78), and the per-proof tail scalars rho_b * [init..., e_r, e_r^2 - 1 ...]
__global__ void __launch_bounds__(64) k_vb_proof(const uint32_t *__restrict__ rho, const uint32_t *__restrict__ sp, const uint32_t *__restrict__ qf2,
                                                 const uint32_t *__restrict__ wit_norm, uint32_t fn, const uint32_t *__restrict__ partial, uint32_t G4, int k,
                                                 const uint32_t *__restrict__ init_sc, uint32_t ninit, const uint32_t *__restrict__ es,
                                                 uint32_t *__restrict__ gs, uint32_t *__restrict__ tail) {
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const fr r26 = fr_load(rho + (size_t)b * 8);
  fe acc = fe_zero();
  for (uint32_t g = t; g < G4; g += 64) acc = fe_add<1>(acc, fe_load(partial + ((size_t)b * G4 + g) * 8));
  for (int d = 32; d >= 1; d >>= 1) acc = fe_add<1>(acc, fe_shfl_down(acc, d));       // lane 0 ends with the sum of all 64
  if (t == 0) {
    fr sc = fr_from_fe(acc);
    const fr q2 = fr_load(qf2 + (size_t)b * 8);
    fr w = q2;
    for (uint32_t i = 0; i < fn; i++) {
      const fr v = fr_load(wit_norm + ((size_t)b * fn + i) * 8);
      sc = fr_addr(sc, fr_mul(w, fr_sqr(v)));
      w = fr_mul(w, q2);
    }
    fr_store(gs + (size_t)b * 8, fr_mul(r26, fr_sub<1>(fr_load(sp + (size_t)b * 8), sc)));
  }
  // tail: ninit init scalars then (e, e^2 - 1) per response, in the order the responses are stored
  uint32_t per = ninit + 2 * (uint32_t)k;
  for (uint32_t m = t; m < per; m += 64) {
    fr v;
    if (m < ninit) v = fr_load(init_sc + ((size_t)b * ninit + m) * 8);
    else {
      uint32_t rr = (m - ninit) >> 1;
      const fr e = fr_load(es + ((size_t)b * k + rr) * 8);
      v = ((m - ninit) & 1u) ? fr_sub<1>(fr_sqr(e), fr_one()) : e;         // makeEs e = (e, e^2 - 1) (NormArgument.hs:109)
    }
    fr_store(tail + ((size_t)b * per + m) * 8, fr_mul(r26, v));
  }
}
__global__ void __launch_bounds__(256) k_vb_sum_gs(const uint32_t *__restrict__ gs, uint32_t batch, uint32_t *__restrict__ out) {
  __shared__ uint32_t lds[256 * 8];
  const uint32_t t = threadIdx.x;
  fe acc = fe_zero();
  for (uint32_t b = t; b < batch; b += 256) acc = fe_add<1>(acc, fe_load(gs + (size_t)b * 8));
  for (int i = 0; i < 8; i++) lds[t * 8 + i] = acc.v[i];
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)t < d) {
      fe x, y;
      for (int i = 0; i < 8; i++) { x.v[i] = lds[t * 8 + i]; y.v[i] = lds[(t + d) * 8 + i]; }
      x = fe_add<1>(x, y);
      for (int i = 0; i < 8; i++) lds[t * 8 + i] = x.v[i];
    }
    __syncthreads();
  }
  if (t == 0) { fe x; for (int i = 0; i < 8; i++) x.v[i] = lds[i]; fe_store(out, x); }
}

// the MSM's point list [G | H | g | per proof: init points, responses] in one pass (16 bytes per lane) instead of five strided copies
__global__ void __launch_bounds__(256) k_vb_gather_points(const uint4 *__restrict__ G, uint32_t nlen, const uint4 *__restrict__ H, uint32_t llen,
                                                          const uint4 *__restrict__ g, const uint4 *__restrict__ init_pts, uint32_t ninit,
                                                          const uint4 *__restrict__ resp, uint32_t nresp, uint64_t T, uint4 *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * T) return;
  const uint64_t t = i >> 2; const uint32_t part = (uint32_t)(i & 3u);
  const uint32_t shared = nlen + llen + 1, per = ninit + nresp;
  const uint4 *src;
  if (t < nlen) src = G + t * 4;
  else if (t < nlen + llen) src = H + (t - nlen) * 4;
  else if (t < shared) src = g;
  else {
    const uint64_t u = t - shared, b = u / per; const uint32_t m = (uint32_t)(u % per);
    src = m < ninit ? init_pts + (b * ninit + m) * 4 : resp + (b * nresp + (m - ninit)) * 4;
  }
  out[i] = src[part];
}

// Validation of the untrusted per-proof arrays: every scalar canonical (< n), rho non-zero, every point the infinity encoding
// or on y^2 = x^3 + 7 with canonical coordinates.  flags[0] |= 1 (scalar) / 2 (point) / 4 (rho = 0).
__global__ void __launch_bounds__(256) k_vb_validate_scalars(const uint32_t *__restrict__ s, uint64_t n, int nonzero, uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe v = fe_load(s + i * 8), t;
  if (!raw_sub(t, v, fr_modulus())) atomicOr(flags, 1u);        // no borrow: v >= n
  if (nonzero && fe_is_zero(v)) atomicOr(flags, 4u);
}
__global__ void __launch_bounds__(256) k_vb_validate_points(const uint32_t *__restrict__ p, uint64_t n, uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_load(p + i * 16), y = fe_load(p + i * 16 + 8), t;
  if (fe_is_zero(x) && fe_is_zero(y)) return;
  bool ok = raw_sub(t, x, fp_modulus()) && raw_sub(t, y, fp_modulus());
  fe seven = fe_zero(); seven.v[0] = 7;
  fe rhs = fe_add<0>(fe_mul<0>(fe_sqr<0>(x), x), seven);
  ok = ok && fe_eq(fe_sqr<0>(y), rhs);
  if (!ok) atomicOr(flags, 2u);
}


// ------------------------------------------------------------------------------------------------ inner-product flavour
// The same batch check for B arguments of src/Bulletproof/InnerProductArgument.hs.  Per proof b, with r_b the argument of makeNorm
// (:194-206; q = r^4), fm = fn / 2 final pairs, es LAST round first:
//   witness through makeNorm 1 (RangeProof.hs:81):  vx_j = (w_2j + w_2j+1) / 2,  vy_j = (w_2j+1 - w_2j) / 2
//   tsX = tensor' vx (1/es) (iterate (^2) q),  tsY = tensor' vy es (repeat 1)                                   (:118-119)
//   the verifier's scalars on the TRANSFORMED basis g'_j = g_2j+1 + r g_2j, h'_j = g_2j+1 - r g_2j are px_j - tsX_j and
//   py_j - tsY_j with px = pub_2j / (2r) + pub_2j+1 / 2, py = -pub_2j / (2r) + pub_2j+1 / 2 (makeNorm of the public vector).
// r differs per proof, so the transformed basis cannot be shared — but the commit is linear in the points:
//   (px - tsX) g' + (py - tsY) h' = [pub_2j - r (tsX_j - tsY_j)] g_2j + [pub_2j+1 - (tsX_j + tsY_j)] g_2j+1
// i.e. the basis change folds into the scalars on the ORIGINAL shared basis (no scalar multiplication per pair, no 1 / r), and the
// sums over the batch land on [G | H | g] exactly as in the norm-linear flavour.
//   Linear (:172-181): challenges inverted, tl = tensor' (n x) (1/es) (repeat 1); scalar on H_j = pub_lin_x_j - tl_j
//   sc = 4 sum_j qF^(j+1) vx_j vy_j + sum_j c_j tl_j,  qF = q^(2^k)  (:116, :176-178; s = 4 from makeNorm);  g: sp - sc
//   responses: makeEs e = (1 / e, e) (:68)
// facx[b] = [q^(2^r) (r < k) | 1 / e_r (first round first)], facy[b] = [unused | e_r]; v[b] = [vx (fm) | vy (fm)]
__global__ void __launch_bounds__(64) k_ipvb_factors(const uint32_t *__restrict__ r_in, const uint32_t *__restrict__ es, const uint32_t *__restrict__ wit_norm,
                                                     uint32_t batch, int k, uint32_t fm, uint32_t *__restrict__ facx, uint32_t *__restrict__ facy,
                                                     uint32_t *__restrict__ qf, uint32_t *__restrict__ v, uint32_t *__restrict__ flags) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const fe r = fe_load(r_in + (size_t)b * 8), r2 = fe_sqr<1>(r);
  fe qp = fe_sqr<1>(r2);
  uint32_t *fx = facx + (size_t)b * 2 * k * 8, *fy = facy + (size_t)b * 2 * k * 8;
  // 1 / e for the k challenges: one inversion (Montgomery's trick along the lane); a zero challenge has no inverse: flagged
  fe run = fe_one();
  for (int i = 0; i < k; i++) {
    const fe e = fe_load(es + ((size_t)b * k + (k - 1 - i)) * 8);         // first round first
    if (fe_is_zero(e)) atomicOr(flags, 8u);
    fe_store(fx + (size_t)i * 8, qp);
    fe_store(fy + (size_t)i * 8, fe_one());
    fe_store(fy + (size_t)(k + i) * 8, e);
    fe_store(fx + (size_t)(k + i) * 8, run);                               // prefix product e_0 .. e_{i-1}
    run = fe_mul<1>(run, e);
    qp = fe_sqr<1>(qp);
  }
  fe_store(qf + (size_t)b * 8, qp);
  fe inv = fe_modinv<1>(run);
  for (int i = k - 1; i >= 0; i--) {
    const fe e = fe_load(fy + (size_t)(k + i) * 8), pre = fe_load(fx + (size_t)(k + i) * 8);
    fe_store(fx + (size_t)(k + i) * 8, fe_mul<1>(inv, pre));
    inv = fe_mul<1>(inv, e);
  }
  fe half = fe_zero();                                                     // (n + 1) / 2
  { const fe n = fr_modulus(); uint32_t carry = 1;
    fe t; for (int i = 0; i < 8; i++) { const uint64_t x = (uint64_t)n.v[i] + carry; t.v[i] = (uint32_t)x; carry = (uint32_t)(x >> 32); }
    for (int i = 0; i < 8; i++) half.v[i] = (t.v[i] >> 1) | (i < 7 ? t.v[i + 1] << 31 : carry << 31); }
  for (uint32_t j = 0; j < fm; j++) {
    const fe w0 = fe_load(wit_norm + ((size_t)b * 2 * fm + 2 * j) * 8), w1 = fe_load(wit_norm + ((size_t)b * 2 * fm + 2 * j + 1) * 8);
    fe_store(v + ((size_t)b * 2 * fm + j) * 8, fe_mul<1>(half, fe_add<1>(w0, w1)));
    fe_store(v + ((size_t)b * 2 * fm + fm + j) * 8, fe_mul<1>(half, fe_sub<1>(w1, w0)));
  }
}
// partial[kt][i] = sum_{b in tile kt} rho_b * (coefficient of proof b on G_i), one (tile, position) per lane
__global__ void __launch_bounds__(256) k_ipvb_norm(const uint32_t *__restrict__ rho, const uint32_t *__restrict__ r_in, const uint32_t *__restrict__ pub,
                                                   const uint32_t *__restrict__ v, uint32_t fm, const uint32_t *__restrict__ facx, const uint32_t *__restrict__ facy,
                                                   uint32_t batch, uint32_t nlen, int k, uint32_t ntiles, uint32_t *__restrict__ partial) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)ntiles * nlen) return;
  const uint32_t kt = (uint32_t)(g / nlen), i = (uint32_t)(g % nlen), j = i >> 1;
  fe acc = fe_zero();
  const uint32_t b0 = kt * KT, b1 = min(batch, b0 + KT);
  for (uint32_t b = b0; b < b1; b++) {
    const fe tx = tensor_at(v + (size_t)b * 2 * fm * 8, fm, facx + (size_t)b * 2 * k * 8, k, j, true);
    const fe ty = tensor_at(v + ((size_t)b * 2 * fm + fm) * 8, fm, facy + (size_t)b * 2 * k * 8, k, j, false);
    const fe p = fe_load(pub + ((size_t)b * nlen + i) * 8);
    const fe d = (i & 1u) ? fe_sub<1>(p, fe_add<1>(tx, ty)) : fe_sub<1>(p, fe_mul<1>(fe_load(r_in + (size_t)b * 8), fe_sub<1>(tx, ty)));
    acc = fe_add<1>(acc, fe_mul<1>(fe_load(rho + (size_t)b * 8), d));
  }
  fe_store(partial + ((size_t)kt * nlen + i) * 8, acc);
}
// per proof: gs[b] = rho_b (sp_b - sc_b) and the tail scalars rho_b [init ..., 1 / e_r, e_r ...] in the order the responses are stored
__global__ void __launch_bounds__(64) k_ipvb_proof(const uint32_t *__restrict__ rho, const uint32_t *__restrict__ sp, const uint32_t *__restrict__ qf,
                                                   const uint32_t *__restrict__ v, uint32_t fm, const uint32_t *__restrict__ wit_lin, uint32_t fl,
                                                   const uint32_t *__restrict__ pub_c, uint32_t llen, const uint32_t *__restrict__ facx, int k,
                                                   const uint32_t *__restrict__ init_sc, uint32_t ninit, const uint32_t *__restrict__ es,
                                                   uint32_t *__restrict__ gs, uint32_t *__restrict__ tail) {
  // one wavefront per proof (the vectors of a single 64-bit proof are a handful of elements: a 256-thread block with an LDS tree was
  // eight barriers around four mostly idle wavefronts); Fr in 10 x 26-bit limbs
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const uint32_t *f = facx + (size_t)b * 2 * k * 8;
  const fr r = fr_load(rho + (size_t)b * 8);
  fr acc26 = fr_zero();
  for (uint32_t j = t; j < llen; j += 64) {
    const uint32_t hi = j >> k;                             // tensor'(wit_lin, 1 / es)[j] (src/Bulletproof.hs:94-95), zero beyond it
    if (hi >= fl) continue;
    fr tl = fr_load(wit_lin + ((size_t)b * fl + hi) * 8);
    for (int rr = 0; rr < k; rr++) if ((j >> rr) & 1u) tl = fr_mul(tl, fr_load(f + (size_t)(k + rr) * 8));
    acc26 = fr_addr(acc26, fr_mul(fr_load(pub_c + ((size_t)b * llen + j) * 8), tl));
  }
  fe acc = fr_to_fe(acc26);
  for (int d = 32; d >= 1; d >>= 1) acc = fe_add<1>(acc, fe_shfl_down(acc, d));       // lane 0 ends with the sum of all 64
  if (t == 0) {
    const fr sc = fr_from_fe(acc);
    const fr q = fr_load(qf + (size_t)b * 8);
    fr w = q, nrm = fr_zero();
    for (uint32_t j = 0; j < fm; j++) {
      nrm = fr_addr(nrm, fr_mul(w, fr_mul(fr_load(v + ((size_t)b * 2 * fm + j) * 8), fr_load(v + ((size_t)b * 2 * fm + fm + j) * 8))));
      w = fr_mul(w, q);
    }
    nrm = fr_dblr(fr_dblr(nrm));                                            // s = 4 (makeNorm)
    fr_store(gs + (size_t)b * 8, fr_mul(r, fr_sub<2>(fr_load(sp + (size_t)b * 8), fr_add(sc, nrm))));
  }
  const uint32_t per = ninit + 2 * (uint32_t)k;
  for (uint32_t m = t; m < per; m += 64) {
    fr val;
    if (m < ninit) val = fr_load(init_sc + ((size_t)b * ninit + m) * 8);
    else {
      const uint32_t rr = (m - ninit) >> 1;                                 // stored last round first; facx holds 1 / e first round first
      val = ((m - ninit) & 1u) ? fr_load(es + ((size_t)b * k + rr) * 8) : fr_load(f + (size_t)(k + (k - 1 - rr)) * 8);
    }
    fr_store(tail + ((size_t)b * per + m) * 8, fr_mul(r, val));
  }
}

}  // namespace bppp

using namespace bppp;

// `validate` = false: the caller made every input itself on the device (csrc/rp.hip: decoded points are on the curve or infinity and
// decoded / derived scalars canonical by construction), so the fifteen validation launches are skipped
namespace bppp {
int nl_verify_batch_run(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                        const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_q,
                        const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                        const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                        const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8], bool validate) {
  if (!ctx || !out_xy) return BPPP_ERR_ARG;
  if (!batch) { memset(out_xy, 0, 64); return BPPP_OK; }
  if (!d_g_xy || !d_rho || !d_q || !d_sp || (nlen && (!d_norm_g_xy || !d_pub_norm)) || (llen && (!d_lin_h_xy || !d_pub_lin_c || !d_pub_lin_x)) ||
      (k && (!d_es || !d_responses_xy)) || (fn && !d_wit_norm) || (fl && !d_wit_lin) || (ninit && (!d_init_scalars || !d_init_points_xy)) || k > 30 ||
      batch >= (1u << 24) || nlen >= (1u << 24) || llen >= (1u << 24))
    return fail(ctx, BPPP_ERR_ARG, "nl_verify_batch: bad arguments");
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  const size_t per = ninit + 2 * k, shared = nlen + llen + 1, T = shared + batch * per;
  const uint32_t ktile = vb_kt(batch), ntiles = (uint32_t)((batch + ktile - 1) / ktile);
  const size_t maxlen = nlen > llen ? nlen : llen;
  const size_t npartial = std::max((size_t)ntiles * maxlen, batch * ((llen + 3) / 4));      // the tile sums, then k_vb_lin_partial's group sums
  // scratch (separate from the MSM workspace, which msm_run carves from ctx->ws)
  size_t words = (batch * 2 * (k ? k : 1) + batch + npartial + (size_t)SUM_GROUPS * maxlen + batch + T + 64) * 8 + T * 16 + 64;
  { int rc0 = ensure_scratch(ctx, words * 4); if (rc0) return rc0; }
  uint32_t *buf = (uint32_t *)ctx->ws2;
  uint32_t *fac = buf, *qf2 = fac + batch * 2 * (k ? k : 1) * 8, *partial = qf2 + batch * 8, *partial2 = partial + npartial * 8, *gs = partial2 + (size_t)SUM_GROUPS * maxlen * 8,
           *sc = gs + batch * 8, *pts = sc + (T + 32) * 8, *flags = pts + T * 16;
  int rc = BPPP_OK;
  do {
    // untrusted inputs first (asynchronous; the flag word is read after the MSM has synchronised the stream)
    if (validate && hipMemsetAsync(flags, 0, 4, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify_batch: memset"); break; }
    auto vs = [&](const void *p, uint64_t n, int nz) {
      if (n) k_vb_validate_scalars<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)p, n, nz, flags);
    };
    auto vp = [&](const void *p, uint64_t n) {
      if (n) k_vb_validate_points<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)p, n, flags);
    };
    if (validate) {
      vs(d_rho, batch, 1); vs(d_q, batch, 0); vs(d_sp, batch, 0); vs(d_pub_norm, batch * nlen, 0); vs(d_pub_lin_c, batch * llen, 0);
      vs(d_pub_lin_x, batch * llen, 0); vs(d_es, batch * k, 0); vs(d_wit_norm, batch * fn, 0); vs(d_wit_lin, batch * fl, 0);
      vs(d_init_scalars, batch * ninit, 0);
      vp(d_g_xy, 1); vp(d_norm_g_xy, nlen); vp(d_lin_h_xy, llen); vp(d_init_points_xy, batch * ninit); vp(d_responses_xy, batch * 2 * k);
    }
    k_vb_factors<<<dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st>>>((const uint32_t *)d_q, (const uint32_t *)d_es, (uint32_t)batch, (int)k, fac, qf2);
    if (nlen) {
      if (k >= 2) k_vb_shared4<<<dim3((unsigned)((nlen + 255) / 256), ntiles), dim3(64), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_norm, (const uint32_t *)d_wit_norm, (uint32_t)fn, fac, (uint32_t)batch, (uint32_t)nlen, (int)k, 1, ktile, partial);
      else k_vb_shared1<<<dim3((unsigned)((nlen + 255) / 256), ntiles), dim3(256), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_norm, (const uint32_t *)d_wit_norm, (uint32_t)fn, fac, (uint32_t)batch, (uint32_t)nlen, (int)k, 1, ktile, partial);
      { const uint32_t per = (ntiles + SUM_GROUPS - 1) / SUM_GROUPS, groups = (ntiles + per - 1) / per;
        k_vb_sum_partials<<<dim3((unsigned)((nlen + 63) / 64), groups), dim3(256), 0, st>>>(partial, ntiles, per, (uint32_t)nlen, partial2);
        k_vb_sum_partials<<<dim3((unsigned)((nlen + 63) / 64), 1), dim3(256), 0, st>>>(partial2, groups, groups, (uint32_t)nlen, sc); }
    }
    if (llen) {
      if (k >= 2) k_vb_shared4<<<dim3((unsigned)((llen + 255) / 256), ntiles), dim3(64), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_lin_x, (const uint32_t *)d_wit_lin, (uint32_t)fl, fac, (uint32_t)batch, (uint32_t)llen, (int)k, 0, ktile, partial);
      else k_vb_shared1<<<dim3((unsigned)((llen + 255) / 256), ntiles), dim3(256), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_lin_x, (const uint32_t *)d_wit_lin, (uint32_t)fl, fac, (uint32_t)batch, (uint32_t)llen, (int)k, 0, ktile, partial);
      { const uint32_t per = (ntiles + SUM_GROUPS - 1) / SUM_GROUPS, groups = (ntiles + per - 1) / per;
        k_vb_sum_partials<<<dim3((unsigned)((llen + 63) / 64), groups), dim3(256), 0, st>>>(partial, ntiles, per, (uint32_t)llen, partial2);
        k_vb_sum_partials<<<dim3((unsigned)((llen + 63) / 64), 1), dim3(256), 0, st>>>(partial2, groups, groups, (uint32_t)llen, sc + nlen * 8); }
    }
    const uint32_t G4 = (uint32_t)((llen + 3) / 4);                 // (the partial-sum buffer is free again: at most llen / 4 of its maxlen / 2 words per proof)
    if (G4) k_vb_lin_partial<<<dim3((unsigned)(((uint64_t)batch * G4 + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)d_wit_lin, (uint32_t)fl, (const uint32_t *)d_pub_lin_c,
                                                                                                      (uint32_t)llen, fac, (int)k, (uint32_t)batch, G4, partial);
    k_vb_proof<<<dim3((unsigned)batch), dim3(64), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_sp, qf2, (const uint32_t *)d_wit_norm, (uint32_t)fn, partial, G4, (int)k,
                                                           (const uint32_t *)d_init_scalars, (uint32_t)ninit, (const uint32_t *)d_es, gs, sc + shared * 8);
    k_vb_sum_gs<<<dim3(1), dim3(256), 0, st>>>(gs, (uint32_t)batch, sc + (nlen + llen) * 8);
    // points: [G | H | g | per proof: init points, responses]
    k_vb_gather_points<<<dim3((unsigned)((4 * T + 255) / 256)), dim3(256), 0, st>>>((const uint4 *)d_norm_g_xy, (uint32_t)nlen, (const uint4 *)d_lin_h_xy, (uint32_t)llen,
                                                                                   (const uint4 *)d_g_xy, (const uint4 *)d_init_points_xy, (uint32_t)ninit,
                                                                                   (const uint4 *)d_responses_xy, (uint32_t)(2 * k), (uint64_t)T, (uint4 *)pts);
    if (hipGetLastError() != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "nl_verify_batch: assembling the MSM failed"); break; }
    rc = msm_run(ctx, sc, pts, T, 1, 1, 0, out_xy);
    if (rc || !validate) break;                 // (msm_run returns with the stream drained; nothing was flagged without the validation pass)
    uint32_t hflags = 0;
    if (hipMemcpyAsync(&hflags, flags, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(ctx, BPPP_ERR_HIP, "nl_verify_batch: reading the validation flags failed"); break;
    }
    if (hflags & 2u) rc = fail(ctx, BPPP_ERR_POINT, "nl_verify_batch: a point is not on the curve");
    else if (hflags & 1u) rc = fail(ctx, BPPP_ERR_ARG, "nl_verify_batch: a scalar is not canonical (>= n)");
    else if (hflags & 4u) rc = fail(ctx, BPPP_ERR_ARG, "nl_verify_batch: a weight rho is zero (it would drop its proof from the combination)");
  } while (0);
  hipStreamSynchronize(st);
  return rc;
}
}  // namespace bppp

extern "C" int bppp_nl_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                                           const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_q,
                                           const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                                           const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                                           const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]) {
  return bppp::nl_verify_batch_run(ctx, batch, nlen, llen, k, fn, fl, ninit, d_g_xy, d_norm_g_xy, d_lin_h_xy, d_rho, d_q, d_sp, d_pub_norm, d_pub_lin_c, d_pub_lin_x, d_es,
                                   d_wit_norm, d_wit_lin, d_init_scalars, d_init_points_xy, d_responses_xy, out_xy, true);
}

// verifyBPM for B arguments of the inner-product flavour (src/Bulletproof/InnerProductArgument.hs; verifyBPM src/Bulletproof.hs:370-378):
// same layout and result contract as bppp_nl_verify_batch_device; d_r holds the per-proof argument of makeNorm (the range proofs'
// challenge q, src/RangeProof/TypedReciprocal.hs:356), fn counts SCALARS of the final norm witness (even).
namespace bppp {
int ip_verify_batch_run(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                        const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_r,
                        const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                        const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                        const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8], bool validate) {
  if (!ctx || !out_xy) return BPPP_ERR_ARG;
  if (!batch) { memset(out_xy, 0, 64); return BPPP_OK; }
  if (!d_g_xy || !d_rho || !d_r || !d_sp || (nlen && (!d_norm_g_xy || !d_pub_norm)) || (llen && (!d_lin_h_xy || !d_pub_lin_c || !d_pub_lin_x)) ||
      (k && (!d_es || !d_responses_xy)) || (fn && !d_wit_norm) || (fl && !d_wit_lin) || (ninit && (!d_init_scalars || !d_init_points_xy)) || k > 30 || (fn & 1) ||
      batch >= (1u << 24) || nlen >= (1u << 24) || llen >= (1u << 24))
    return fail(ctx, BPPP_ERR_ARG, "ip_verify_batch: bad arguments");
  if (ctx_closed(ctx)) return BPPP_ERR_ARG;
  hipSetDevice(ctx->device);
  hipStream_t st = ctx->stream;
  const size_t per = ninit + 2 * k, shared = nlen + llen + 1, T = shared + batch * per, fm = fn / 2, kk = k ? k : 1;
  const uint32_t ktile = (uint32_t)KT, ntiles = (uint32_t)((batch + KT - 1) / KT);
  const size_t maxlen = nlen > llen ? nlen : llen;
  size_t words = (2 * batch * 2 * kk + batch + batch * (fn ? fn : 2) + (size_t)ntiles * maxlen + (size_t)SUM_GROUPS * maxlen + batch + T + 64) * 8 + T * 16 + 64;
  { int rc0 = ensure_scratch(ctx, words * 4); if (rc0) return rc0; }
  uint32_t *buf = (uint32_t *)ctx->ws2;
  uint32_t *facx = buf, *facy = facx + batch * 2 * kk * 8, *qf = facy + batch * 2 * kk * 8, *v = qf + batch * 8, *partial = v + batch * (fn ? fn : 2) * 8,
           *partial2 = partial + (size_t)ntiles * maxlen * 8, *gs = partial2 + (size_t)SUM_GROUPS * maxlen * 8, *sc = gs + batch * 8, *pts = sc + (T + 32) * 8,
           *flags = pts + T * 16;
  int rc = BPPP_OK;
  do {
    if (hipMemsetAsync(flags, 0, 4, st) != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify_batch: memset"); break; }
    auto vs = [&](const void *p, uint64_t n, int nz) {
      if (n) k_vb_validate_scalars<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)p, n, nz, flags);
    };
    auto vp = [&](const void *p, uint64_t n) {
      if (n) k_vb_validate_points<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)p, n, flags);
    };
    if (validate) {
      vs(d_rho, batch, 1); vs(d_r, batch, 0); vs(d_sp, batch, 0); vs(d_pub_norm, batch * nlen, 0); vs(d_pub_lin_c, batch * llen, 0);
      vs(d_pub_lin_x, batch * llen, 0); vs(d_es, batch * k, 0); vs(d_wit_norm, batch * fn, 0); vs(d_wit_lin, batch * fl, 0);
      vs(d_init_scalars, batch * ninit, 0);
      vp(d_g_xy, 1); vp(d_norm_g_xy, nlen); vp(d_lin_h_xy, llen); vp(d_init_points_xy, batch * ninit); vp(d_responses_xy, batch * 2 * k);
    }
    k_ipvb_factors<<<dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st>>>((const uint32_t *)d_r, (const uint32_t *)d_es, (const uint32_t *)d_wit_norm, (uint32_t)batch,
                                                                            (int)k, (uint32_t)fm, facx, facy, qf, v, flags);
    const uint32_t sper = (ntiles + SUM_GROUPS - 1) / SUM_GROUPS, groups = (ntiles + sper - 1) / sper;
    if (nlen) {
      const uint64_t lanes = (uint64_t)ntiles * nlen;
      k_ipvb_norm<<<dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_r, (const uint32_t *)d_pub_norm, v, (uint32_t)fm,
                                                                               facx, facy, (uint32_t)batch, (uint32_t)nlen, (int)k, ntiles, partial);
      k_vb_sum_partials<<<dim3((unsigned)((nlen + 63) / 64), groups), dim3(256), 0, st>>>(partial, ntiles, sper, (uint32_t)nlen, partial2);
      k_vb_sum_partials<<<dim3((unsigned)((nlen + 63) / 64), 1), dim3(256), 0, st>>>(partial2, groups, groups, (uint32_t)nlen, sc);
    }
    if (llen) {
      if (k >= 2) k_vb_shared4<<<dim3((unsigned)((llen + 255) / 256), ntiles), dim3(64), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_lin_x, (const uint32_t *)d_wit_lin, (uint32_t)fl, facx, (uint32_t)batch, (uint32_t)llen, (int)k, 0, ktile, partial);
      else k_vb_shared1<<<dim3((unsigned)((llen + 255) / 256), ntiles), dim3(256), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_pub_lin_x, (const uint32_t *)d_wit_lin, (uint32_t)fl, facx, (uint32_t)batch, (uint32_t)llen, (int)k, 0, ktile, partial);
      k_vb_sum_partials<<<dim3((unsigned)((llen + 63) / 64), groups), dim3(256), 0, st>>>(partial, ntiles, sper, (uint32_t)llen, partial2);
      k_vb_sum_partials<<<dim3((unsigned)((llen + 63) / 64), 1), dim3(256), 0, st>>>(partial2, groups, groups, (uint32_t)llen, sc + nlen * 8);
    }
    k_ipvb_proof<<<dim3((unsigned)batch), dim3(64), 0, st>>>((const uint32_t *)d_rho, (const uint32_t *)d_sp, qf, v, (uint32_t)fm, (const uint32_t *)d_wit_lin, (uint32_t)fl,
                                                              (const uint32_t *)d_pub_lin_c, (uint32_t)llen, facx, (int)k, (const uint32_t *)d_init_scalars, (uint32_t)ninit,
                                                              (const uint32_t *)d_es, gs, sc + shared * 8);
    k_vb_sum_gs<<<dim3(1), dim3(256), 0, st>>>(gs, (uint32_t)batch, sc + (nlen + llen) * 8);
    k_vb_gather_points<<<dim3((unsigned)((4 * T + 255) / 256)), dim3(256), 0, st>>>((const uint4 *)d_norm_g_xy, (uint32_t)nlen, (const uint4 *)d_lin_h_xy, (uint32_t)llen,
                                                                                   (const uint4 *)d_g_xy, (const uint4 *)d_init_points_xy, (uint32_t)ninit,
                                                                                   (const uint4 *)d_responses_xy, (uint32_t)(2 * k), (uint64_t)T, (uint4 *)pts);
    if (hipGetLastError() != hipSuccess) { rc = fail(ctx, BPPP_ERR_HIP, "ip_verify_batch: assembling the MSM failed"); break; }
    rc = msm_run(ctx, sc, pts, T, 1, 1, 0, out_xy);
    if (rc) break;
    uint32_t hflags = 0;
    if (hipMemcpyAsync(&hflags, flags, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(ctx, BPPP_ERR_HIP, "ip_verify_batch: reading the validation flags failed"); break;
    }
    if (hflags & 2u) rc = fail(ctx, BPPP_ERR_POINT, "ip_verify_batch: a point is not on the curve");
    else if (hflags & 1u) rc = fail(ctx, BPPP_ERR_ARG, "ip_verify_batch: a scalar is not canonical (>= n)");
    else if (hflags & 4u) rc = fail(ctx, BPPP_ERR_ARG, "ip_verify_batch: a weight rho is zero (it would drop its proof from the combination)");
    else if (hflags & 8u) rc = fail(ctx, BPPP_ERR_ARG, "ip_verify_batch: a challenge is zero (makeEs needs its inverse)");
  } while (0);
  hipStreamSynchronize(st);
  return rc;
}
}  // namespace bppp

extern "C" int bppp_ip_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                                           const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_r,
                                           const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                                           const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                                           const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]) {
  return bppp::ip_verify_batch_run(ctx, batch, nlen, llen, k, fn, fl, ninit, d_g_xy, d_norm_g_xy, d_lin_h_xy, d_rho, d_r, d_sp, d_pub_norm, d_pub_lin_c, d_pub_lin_x, d_es,
                                   d_wit_norm, d_wit_lin, d_init_scalars, d_init_points_xy, d_responses_xy, out_xy, true);
}
