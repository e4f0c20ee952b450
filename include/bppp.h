/*
 * bppp.h — C ABI of libbppp_hip.so: the MI355X (gfx950) implementation of the Bulletproofs++
 * hot path of Liam-Eagen/BulletproofsPP.  Plain pointers and sizes only; no C++/torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the Haskell `foreign import ccall` stubs that bind
 * these symbols behind the reference's own typeclasses (FastInnerProduct / BPOpening).
 *
 * Data formats (all little-endian, limb 0 least significant):
 *   scalar  Fr : 4 x uint64, canonical integer in [0, n)              (as FastPrime's 4 words,
 *                src/Data/Field/Galois/FastPrime/Internal.hs:152-176; Encoding.hs:75-86)
 *   point      : 8 x uint64 = affine x[4] ++ y[4], canonical in [0, p); infinity = all zero
 *                (what `toA` yields, src/Commitment.hs:172-176; (0,0) is not on y^2 = x^3 + 7)
 *   reduced scalar (ReducedScalar (Prime p) = Integer, Commitment.hs:270): sign flag + 3 x uint64
 *                magnitude (< 2^130; rationalReducedScalarLength = 129, Commitment.hs:286)
 *
 * Buffers named d_* are DEVICE pointers (HBM-resident, 16-byte aligned); everything else is host
 * memory.  All calls are synchronous with respect to the caller on return (outputs are valid),
 * matching a Haskell `foreign import ccall safe`.  A context is bound to one GPU and one HIP
 * stream and is not thread-safe; use one context per thread/GPU.
 *
 * Every function returns BPPP_OK (0) or a negative BPPP_ERR_* code; bppp_last_error() gives text.
 *
 * Handle lifetime: every child handle (bppp_nl, bppp_nlb, bppp_ip, bppp_trrp, bppp_rp, bppp_basis) holds a reference on its context.
 * bppp_ctx_destroy closes the context (further calls through it or its children fail with BPPP_ERR_ARG) and drops the caller's
 * reference; the stream and workspaces are released when the last child is destroyed, so finalisers may run in any order.
 *
 * Untrusted input: the verifier entry points that take DECODED values (bppp_nl_verify, bppp_ip_verify, bppp_nl_verify_batch_device)
 * check that proof-supplied scalars are canonical (< n) and points are on the curve (or the infinity encoding) and return
 * BPPP_ERR_ARG / BPPP_ERR_POINT otherwise; the batch weights rho must be non-zero.  The entry points that take the reference's FILES
 * (bppp_rp_verify_batch*, bppp_rp_verify_shard_device) decode them as the reference does: `get` of Binary (Prime p) ends in toP
 * (src/Encoding.hs:76-80), i.e. a coordinate >= p or a scalar >= n is REDUCED, not refused, so a proof has more than one accepted
 * byte encoding (the files are malleable exactly as the reference's are); an x with no curve point is BPPP_RP_MALFORMED.
 *
 * Side channels: the PROVER entry points are not constant-time.  The fixed-base comb walk (csrc/comb.hip) indexes its HBM table by
 * the signed digits of secret scalars (blindings included) and skips zero scalars per lane, and the bucket method sorts by secret
 * digits; memory access pattern and running time depend on the witness.  The reference is not constant-time either (Integer /
 * GMP arithmetic, testBit-driven additions, src/Commitment.hs:325-335); a deployment that shares the GPU with an adversary needs
 * its own isolation.
 */
#ifndef BPPP_H
#define BPPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#pragma GCC visibility push(default)

#define BPPP_OK 0
#define BPPP_ERR_ARG (-1)      /* bad length / null pointer / unsupported parameter */
#define BPPP_ERR_HIP (-2)      /* a HIP runtime call or kernel launch failed */
#define BPPP_ERR_NODEVICE (-3) /* no gfx950 GPU visible */
#define BPPP_ERR_POINT (-4)    /* a proof-supplied point is not on the curve (verifier entry points validate their inputs) */

typedef struct bppp_ctx bppp_ctx;

/* ---- context ---------------------------------------------------------------------------- */
int bppp_ctx_create(int device, bppp_ctx **out);
void bppp_ctx_destroy(bppp_ctx *ctx);
/* run all work of this context on an existing hipStream_t (e.g. torch's current stream);
 * NULL = the context's own stream. */
int bppp_ctx_set_stream(bppp_ctx *ctx, void *hip_stream);
const char *bppp_last_error(const bppp_ctx *ctx);
const char *bppp_version(void);

/* ---- a1: FastInnerProduct.innerProduct / commit --------------------------------------------
 * Replaces `innerProduct :: [(Scalar v, v)] -> v` (src/Commitment.hs:325-335) as reached through
 * `commit` (src/Commitment.hs:416-417): out = sum_i scalars[i] * points[i], as the canonical
 * affine point.  Zero scalars and points at infinity are allowed (dotWith pads with both,
 * Commitment.hs:423-424).  n = 0 returns infinity (the reference would crash on `head []`,
 * Commitment.hs:328 — documented deviation).  Algorithm: signed-digit Pippenger bucket method
 * (a different algorithm from the reference's 256-row Straus loop; same group element). */
int bppp_msm(bppp_ctx *ctx, const uint64_t *scalars, const uint64_t *points_xy, size_t n, uint64_t out_xy[8]);
/* same, inputs already resident in HBM; window_bits = 0 lets the library choose. */
int bppp_msm_device(bppp_ctx *ctx, const void *d_scalars, const void *d_points_xy, size_t n, int window_bits,
                    uint64_t out_xy[8]);
/* `batch` independent MSMs of n terms each in one pass (one per proof: verifyBPM's single commit,
 * src/Bulletproof.hs:377).  d_scalars is [batch][n]; d_points_xy is [batch][n] when shared_points == 0, one [n] basis used
 * by every instance when shared_points == 1, and [batch / d][n] (one basis per d consecutive instances, e.g. the X and R
 * commitments of one proof) when shared_points == d >= 2.  out_xy is [batch][8] on the host. */
int bppp_msm_batch_device(bppp_ctx *ctx, const void *d_scalars, const void *d_points_xy, size_t n, size_t batch,
                          int shared_points, int window_bits, uint64_t *out_xy);
/* Sum of n affine points (complete group law): the local tail of a sharded MSM after the ranks all-gathered their partial points
 * (SURVEY.md 8e).  Host arithmetic, microseconds for the 2..64 points it is meant for. */
int bppp_sum_points(bppp_ctx *ctx, const uint64_t *points_xy, size_t n, uint64_t out_xy[8]);

/* ---- registered basis with fixed-base precomputation (SURVEY.md 8(b) "Ownership": basis points "may be registered once and
 * referenced by handle since G, H are fixed per setup", src/RangeProof/TypedReciprocal.hs:348-359) ---------------------------
 * The handle owns a copy of the n points in HBM plus the table T[w][i] = 2^(c w) P_i (W = 256 / c + 1 rows of n affine points,
 * W * n * 64 bytes; built on the device at creation).  bppp_msm_basis is `innerProduct` (src/Commitment.hs:325-335) over the first
 * n_terms registered points — `batch` instances, d_scalars [batch][n_terms] in HBM, out_xy [batch][8] on the host — with all
 * windows sharing ONE bucket set: one bucket reduction per instance instead of W, no window combine.  Results equal
 * bppp_msm_batch_device over the same points bit for bit.  window_bits = 0 lets the library choose for `batch_hint` instances
 * of n terms.  The arbitrary-point entry points above stay the general route (and the headline of bench.py). */
typedef struct bppp_basis bppp_basis;
int bppp_basis_create(bppp_ctx *ctx, const uint64_t *points_xy, size_t n, int window_bits, size_t batch_hint, bppp_basis **out);
int bppp_basis_create_device(bppp_ctx *ctx, const void *d_points_xy, size_t n, int window_bits, size_t batch_hint, bppp_basis **out);
void bppp_basis_destroy(bppp_basis *basis);
int bppp_basis_info(const bppp_basis *basis, size_t *n, int *window_bits, size_t *table_bytes);
int bppp_msm_basis(bppp_basis *basis, const void *d_scalars, size_t n_terms, size_t batch, uint64_t *out_xy);
/* For MANY instances over a SHORT basis (the prover's commitments: thousands of MSMs of ~775 terms over a setup's points) the handle can
 * also hold every multiple of every window, tab[w][i][d-1] = d 2^(c w) P_i: bppp_msm_basis of >= 64 instances is then one mixed addition
 * per non-zero signed c-bit digit into one accumulator — no sort, no buckets, no reduction, no doubling.  window_bits = 0 takes the widest
 * window (<= 18) whose table fits budget_bytes (W * n * 2^(c-1) * 64 bytes: 27.6 GB for 774 points at c = 16, 4.1 GB at c = 13).  Same
 * results bit for bit. */
int bppp_basis_enable_comb(bppp_basis *basis, int window_bits, size_t budget_bytes, int *window_bits_out, size_t *table_bytes);

/* ---- a7: SplitScalar.rationalReduceScalar (host) ------------------------------------------
 * Replaces rationalReduceScalar for `Prime p` (src/Commitment.hs:242-255, instance :269-288):
 * returns (a, b) with x = a / b (mod n), following the reference's egcd step-for-step (its
 * choice of (a, b) fixes the collapsed basis points, so it must match exactly). */
int bppp_rational_reduce(const uint64_t x[4], uint64_t a_mag[3], int *a_neg, uint64_t b_mag[3], int *b_neg);

/* ---- a8: projectivePairIP via collapsePoints, over a whole vector ---------------------------
 * Replaces `collapsePoints b a gL gR = projectivePairIP (b, gL) (a, gR)` (src/Bulletproof.hs:213-214,
 * src/Commitment.hs:343-353) mapped over adjacent pairs by mapHalves (src/Bulletproof.hs:88-90):
 * out[j] = b * pts[2j] + a * pts[2j+1] for j < ceil(n/2); an odd tail pairs with infinity.
 * The same (a, b) is used for every pair (one uniform add/double schedule per wavefront). */
int bppp_fold_points(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg,
                     const uint64_t *points_xy, size_t n, uint64_t *out_xy);
int bppp_fold_points_device(bppp_ctx *ctx, const uint64_t b_mag[3], int b_neg, const uint64_t a_mag[3], int a_neg,
                            const void *d_points_xy, size_t n, void *d_out_xy);

/* ---- a7 / a8, Eisenstein variant: the reference's FastPrime configuration (src/Commitment.hs:293-306) -------------------------
 * ReducedScalar = Eis Integer: a reduced scalar is a0 + a1*w (w^3 = 1; w acts on scalars as lambda, on points as the
 * endomorphism (x, y) -> (beta x, y), src/Data/Curve/CM.hs:25-27) with components of about 65 bits
 * (rationalReducedScalarLength = 65).  bppp_rational_reduce_eis is the class default rationalReduceScalar (Commitment.hs:242-255)
 * over that instance: reducedChar = conjEis . charEis, reduceScalar = decomposeEis (FastPrime.hs:186-205), Euclid with the
 * nearest-integer quotRem of Integral (Eis a) (src/Data/Field/Eis.hs:72-82), stop at the first r with (normEis r)^2 <= 2n.
 * Components: mag[2*k], mag[2*k+1] = the two 64-bit limbs of |component k|, neg[k] its sign (k = 0: rational part, 1: w part).
 * bppp_fold_points_eis_device is projectivePairIP of that configuration (Commitment.hs:343-353 with the FastInnerProduct instance
 * :374-398, 65 rows) mapped over adjacent pairs: out[j] = b' * pts[2j] + a' * pts[2j+1].  Same cost as the integer fold
 * (two 66-row walks instead of one 130-row walk): provided for parity with that configuration, not as the fast path. */
int bppp_rational_reduce_eis(const uint64_t x[4], uint64_t a_mag[4], int a_neg[2], uint64_t b_mag[4], int b_neg[2]);
int bppp_fold_points_eis_device(bppp_ctx *ctx, const uint64_t b_mag[4], const int b_neg[2], const uint64_t a_mag[4], const int a_neg[2],
                                const void *d_points_xy, size_t n, void *d_out_xy);

/* ---- a10/a11/a16: scalar halves of the Norm / Linear round (NL flavour) ---------------------
 * makeScalarsComs scalar sums (src/Bulletproof/NormArgument.hs:113-118 via foldXR :20-29):
 *   sx = sum_j q^(4j) xL_j xR_j,  sr = sum_j q^(4j) xR_j^2   over adjacent pairs (odd tail: xR = 0)
 * The caller applies the 2 n^2 q^3 / n^2 q^4 factors (host glue, one multiplication each). */
int bppp_norm_round_sums_device(bppp_ctx *ctx, const void *d_x, size_t n, const uint64_t q4[4], uint64_t sx[4],
                                uint64_t sr[4]);
/* Linear makeScalarsComs sums (NormArgument.hs:56-59): sx = sum cL xR + cR xL, sr = sum cR xR */
int bppp_lin_round_sums_device(bppp_ctx *ctx, const void *d_c, const void *d_x, size_t n, uint64_t sx[4],
                               uint64_t sr[4]);
/* X / R opening scalars of the Norm round (NormArgument.hs:117): d_xw[2j] = q xR_j,
 * d_xw[2j+1] = qinv xL_j (length 2*ceil(n/2)); d_rw[j] = xR_j (length ceil(n/2)). */
int bppp_norm_round_openings_device(bppp_ctx *ctx, const void *d_x, size_t n, const uint64_t q[4],
                                    const uint64_t qinv[4], void *d_xw, void *d_rw);
/* Linear X / R opening scalars (NormArgument.hs:59): d_xw[2j] = xR_j, d_xw[2j+1] = xL_j; d_rw[j] = xR_j */
int bppp_lin_round_openings_device(bppp_ctx *ctx, const void *d_x, size_t n, void *d_xw, void *d_rw);
/* scalar vector fold of collapse (NormArgument.hs:129 / :71): out[j] = u * x[2j] + v * x[2j+1]
 * with u = b0^-1, v = e q b0^-1 (norm) or e b0^-1 (linear x) or (u, v) = (b0, a0) (linear c). */
int bppp_fold_scalars_device(bppp_ctx *ctx, const uint64_t u[4], const uint64_t v[4], const void *d_x, size_t n,
                             void *d_out);

/* ---- a5: batchInverse ----------------------------------------------------------------------
 * Replaces batchInverse (src/Data/Field/BatchInverse.hs:14-24; used by normalizes, src/Commitment.hs:125, :153, and by
 * the range-proof phases, src/RangeProof/TypedReciprocal.hs:193-195): out[i] = x[i]^-1, with 0 -> 0, over n canonical
 * field elements in HBM.  modulus: 0 = Fq (coordinates), 1 = Fr (scalars).  Montgomery's trick runs along each lane
 * (8 values per inversion); d_out may alias d_x. */
int bppp_batch_inverse_device(bppp_ctx *ctx, const void *d_x, size_t n, int modulus, void *d_out);

/* ---- a15: tensor' (challenge expansion) -----------------------------------------------------
 * Replaces the list instance of tensor' (src/Bulletproof.hs:94-95) as used by expandChallenges
 * (NormArgument.hs:73-81, :131-145): out[i * 2^k + t] = bs[i] * prod over rounds of (q_r or e_r)
 * selected by the bits of t.  es (k challenges, LAST ROUND FIRST as verifyBPM holds them,
 * Bulletproof.hs:374) and qs (k weights, first round first) are host arrays of k scalars. */
int bppp_tensor_device(bppp_ctx *ctx, const uint64_t *bs, size_t nb, const uint64_t *es, const uint64_t *qs,
                       size_t k, void *d_out);

/* ---- a10-a14: the norm-linear argument with device-resident state (BPOpening / BPCollection) ---
 * `bppp_nl` is the device counterpart of `PedersenScalarVector (NormLinear f) v s`
 * (src/Commitment.hs:487-501, src/Bulletproof/NormArgument.hs:153-162): the norm vector x with basis G,
 * the linear vector (c, x) with basis H, the scalar s on g, and the deferred normalisations
 * (BPFrame''.nrmlz'', src/Bulletproof.hs:167).  The Fiat-Shamir oracle stays with the caller
 * (injected in the reference too, src/ZKP.hs:73-77). */
typedef struct bppp_nl bppp_nl;
/* makeNormLinearBP' 1 q cs nss ngs lss lgs (NormArgument.hs:162) inside makePSV s g (Commitment.hs:490-491) */
int bppp_nl_create(bppp_ctx *ctx, const uint64_t s[4], const uint64_t g_xy[8], const uint64_t q[4], const uint64_t *norm_x,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy,
                   size_t llen, bppp_nl **out);
void bppp_nl_destroy(bppp_nl *nl);
int bppp_nl_lengths(const bppp_nl *nl, size_t *nlen, size_t *llen);
/* first half of proveRoundM (src/Bulletproof.hs:346-350): makeScalarsComs (NormArgument.hs:113-118, :56-59,
 * Bulletproof.hs:258-261) and ac = commit(sX, X-opening), bc = commit(sR, R-opening) as one batched MSM. */
int bppp_nl_round_commit(bppp_nl *nl, uint64_t sX[4], uint64_t X_xy[8], uint64_t sR[4], uint64_t R_xy[8]);
/* second half of proveRoundM (Bulletproof.hs:352-354): (e0, e1) = makeEs e = (e, e^2 - 1); s += e0 sX + e1 sR;
 * collapse e (NormArgument.hs:123-129, :64-71): scalar folds, basis folds by collapsePoints, q <- q^2. */
int bppp_nl_round_collapse(bppp_nl *nl, const uint64_t e[4]);
/* getWitness (NormArgument.hs:121, :62; Bulletproof.hs:264): normalisation applied; lengths from bppp_nl_lengths */
int bppp_nl_get_witness(bppp_nl *nl, uint64_t *norm_w, uint64_t *lin_w);
/* raw current state (vectors NOT multiplied by the normalisations) for parity checks; any pointer may be NULL */
int bppp_nl_download(bppp_nl *nl, uint64_t *norm_x, uint64_t *norm_g_xy, uint64_t *lin_c, uint64_t *lin_x, uint64_t *lin_h_xy,
                     uint64_t s[4], uint64_t q[4], uint64_t norm_nrmlz[4], uint64_t lin_nrmlz[4]);
/* verifyBPM (src/Bulletproof.hs:370-378) after the challenges are known: expandChallenges
 * (NormArgument.hs:73-81, :131-145; Bulletproof.hs:268-269) on the device, then the single commit over
 * verifyWith's term list (Bulletproof.hs:362-368): wit' ++ initCom ++ [e0 X_j, e1 R_j].  `es` and
 * `responses_xy` (k pairs X, R) are LAST ROUND FIRST as the reference holds them (Bulletproof.hs:359, :374).
 * out_xy is the committed point; the proof verifies iff it is infinity (all zero). */
int bppp_nl_verify(bppp_ctx *ctx, const uint64_t q[4], const uint64_t sp[4], const uint64_t g_xy[8], const uint64_t *pub_norm,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *pub_lin_c, const uint64_t *pub_lin_x, const uint64_t *lin_h_xy,
                   size_t llen, const uint64_t *es, size_t k, const uint64_t *wit_norm, size_t fn, const uint64_t *wit_lin, size_t fl,
                   const uint64_t *init_scalars, const uint64_t *init_points_xy, size_t ninit, const uint64_t *responses_xy, uint64_t out_xy[8]);

/* The injected Fiat-Shamir oracle (MonadZKP.oracle, src/ZKP.hs:57, :96-101; app/Main.hs:75-80 is the CLI's SHA-256 one):
 * called with the WHOLE transcript so far, newest commitments first, as affine points; writes the first challenge. */
typedef void (*bppp_oracle_fn)(void *user, const uint64_t *transcript_xy, size_t npoints, uint64_t challenge[4]);
/* proveBPM (src/Bulletproof.hs:357-359): n_rounds x proveRoundM with the caller's oracle; responses and challenges come out
 * LAST ROUND FIRST.  transcript_xy (capacity transcript_cap points) holds *ntranscript earlier commitments on entry and
 * receives the responses (newest first). */
int bppp_nl_prove(bppp_nl *nl, size_t n_rounds, bppp_oracle_fn oracle, void *user, uint64_t *transcript_xy, size_t *ntranscript,
                  size_t transcript_cap, uint64_t *responses_xy, uint64_t *es);
/* the challenge derivation of verifyBPM (src/Bulletproof.hs:374) with the same oracle contract */
int bppp_nl_verify_challenges(bppp_oracle_fn oracle, void *user, const uint64_t *responses_xy, size_t k, uint64_t *transcript_xy,
                              size_t *ntranscript, size_t transcript_cap, uint64_t *es);

/* ---- lockstep batch prover: `batch` norm-linear arguments of one shape advance round by round together ----------------
 * Same functions as bppp_nl_* (proveRoundM, src/Bulletproof.hs:346-355) with a leading batch dimension on every array:
 * one proof cannot fill the chip (its MSMs have < 800 terms, its basis fold is one ~1 ms dependency chain), B proofs
 * share every launch.  The starting basis (g, G, H) is shared; q, s and the vectors are per proof ([batch][...]). */
typedef struct bppp_nlb bppp_nlb;
int bppp_nlb_create(bppp_ctx *ctx, size_t batch, const uint64_t *s, const uint64_t g_xy[8], const uint64_t *q, const uint64_t *norm_x,
                    const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy, size_t llen,
                    bppp_nlb **out);
void bppp_nlb_destroy(bppp_nlb *nlb);
int bppp_nlb_lengths(const bppp_nlb *nlb, size_t *batch, size_t *nlen, size_t *llen);
int bppp_nlb_round_commit(bppp_nlb *nlb, uint64_t *sX, uint64_t *X_xy, uint64_t *sR, uint64_t *R_xy);   /* [batch][4], [batch][8] */
int bppp_nlb_round_collapse(bppp_nlb *nlb, const uint64_t *es);                                         /* [batch][4] */
int bppp_nlb_get_witness(bppp_nlb *nlb, uint64_t *norm_w, uint64_t *lin_w, uint64_t *s);

/* ---- a12: the inner-product flavour (src/Bulletproof/InnerProductArgument.hs; the CLI's default, app/Parse.hs:100)
 * Same contract as bppp_nl_*.  `r` is the argument of makeNorm (:194-206; q = r^4): the norm vector (nlen scalars on nlen
 * points) is re-expressed as ceil(nlen/2) inner-product pairs with the basis change g' = g1 + r g0, h' = g1 - r g0 done
 * on the device (one full scalar multiplication per pair, :204).  makeEs e = (1/e, e) (:68). */
typedef struct bppp_ip bppp_ip;
int bppp_ip_create(bppp_ctx *ctx, const uint64_t s[4], const uint64_t g_xy[8], const uint64_t r[4], const uint64_t *norm_s,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy,
                   size_t llen, bppp_ip **out);
void bppp_ip_destroy(bppp_ip *ip);
int bppp_ip_lengths(const bppp_ip *ip, size_t *ip_len, size_t *llen);   /* ip_len = number of (x, y) pairs */
/* makeScalarsComs (:70-81, :155-158 via foldLR :17-26) + the two commits of proveRoundM (src/Bulletproof.hs:348-350) */
int bppp_ip_round_commit(bppp_ip *ip, uint64_t sL[4], uint64_t L_xy[8], uint64_t sR[4], uint64_t R_xy[8]);
/* s += e^-1 sL + e sR; collapse (:86-101, :162-170) */
int bppp_ip_round_collapse(bppp_ip *ip, const uint64_t e[4]);
/* getWitness: norm_w holds 2*ip_len scalars (nx x - ny y, nx x + ny y) (:222-223); lin_w llen scalars; s the PSV scalar */
int bppp_ip_get_witness(bppp_ip *ip, uint64_t *norm_w, uint64_t *lin_w, uint64_t s[4]);
/* verifyBPM for this flavour: basis change, expandChallenges (:103-124, :172-181), one MSM; out must be infinity */
int bppp_ip_verify(bppp_ctx *ctx, const uint64_t r[4], const uint64_t sp[4], const uint64_t g_xy[8], const uint64_t *pub_norm,
                   const uint64_t *norm_g_xy, size_t nlen, const uint64_t *pub_lin_c, const uint64_t *pub_lin_x, const uint64_t *lin_h_xy,
                   size_t llen, const uint64_t *es, size_t k, const uint64_t *wit_norm, size_t fn, const uint64_t *wit_lin, size_t fl,
                   const uint64_t *init_scalars, const uint64_t *init_points_xy, size_t ninit, const uint64_t *responses_xy, uint64_t out_xy[8]);

/* ---- batch verifier (BASELINE config 5) --------------------------------------------------------
 * No reference implementation exists (TODO at src/RangeProof/TypedReciprocal.hs:469-472); semantics per
 * out = sum_b rho[b] * MSM(T_b) with T_b the verifyWith term list of proof b
 * (src/Bulletproof.hs:362-368, :375-377).  All proofs share the shape (nlen, llen, k rounds, fn, fl, ninit)
 * and the basis (G, H, g); q, sp, the public vectors, challenges, final openings, initCom terms and responses
 * are per proof, stored [batch][...] contiguously in HBM.  Scalars on the shared basis are summed over the
 * batch on the device; one MSM of (nlen + llen + 1) + batch * (ninit + 2k) terms decides every proof:
 * out is infinity iff all verify (rho random, rho[0] = 1 by convention). */
int bppp_nl_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                                const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_q,
                                const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                                const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                                const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]);

/* The same batch check for the inner-product flavour (src/Bulletproof/InnerProductArgument.hs; verifyBPM src/Bulletproof.hs:370-378 with
 * expandChallenges :103-124, :172-181 and makeEs e = (1/e, e) :68).  Same layout and result as bppp_nl_verify_batch_device; d_r holds the
 * per-proof argument of makeNorm (:194-206; the range proofs pass their challenge q), fn counts the SCALARS of the final norm witness
 * (two per inner-product pair, as getWitness :222-223 lays them out).  makeNorm's basis change g' = g1 + r g0, h' = g1 - r g0 depends
 * on the proof, so it is folded into the scalars on the ORIGINAL basis — x' g' + y' h' = r (x' - y') g0 + (x' + y') g1 — and the
 * per-proof scalars still sum onto the one shared basis [G | H | g]: no scalar multiplication per basis pair (the reference's TODO at
 * InnerProductArgument.hs:187-190, :225-227) and one MSM of (nlen + llen + 1) + batch * (ninit + 2k) terms for the whole batch. */
int bppp_ip_verify_batch_device(bppp_ctx *ctx, size_t batch, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl, size_t ninit,
                                const void *d_g_xy, const void *d_norm_g_xy, const void *d_lin_h_xy, const void *d_rho, const void *d_r,
                                const void *d_sp, const void *d_pub_norm, const void *d_pub_lin_c, const void *d_pub_lin_x,
                                const void *d_es, const void *d_wit_norm, const void *d_wit_lin, const void *d_init_scalars,
                                const void *d_init_points_xy, const void *d_responses_xy, uint64_t out_xy[8]);

/* ---- the optional endomorphism (GLV) path ---------------------------------------------------------------
 * bppp_glv_decompose_device: decomposeFastPrimeEis (src/Data/Field/Galois/FastPrime.hs:186-205) for n canonical scalars:
 * x = a + b*lambda (mod n), the reference's own (a, b) including its one-step rounding.  d_a_mag / d_b_mag: [n][4] magnitudes,
 * d_signs: [n] uint32, bit 0 = a negative, bit 1 = b negative (bit 2 would flag a magnitude over 256 bits; it cannot happen).
 * bppp_msm_glv_device: the same group element as bppp_msm_device through that decomposition — 2n half-length terms
 * |a| * (+-P), |b| * (+-(beta x, y)) (SplitScalar / FastInnerProduct of FastPrime, src/Commitment.hs:293-306, :374-398; cmConj,
 * src/Data/Curve/CM.hs:25-27).  Not faster at 2^20 (same number of bucket additions); provided for parity with that option. */
int bppp_glv_decompose_device(bppp_ctx *ctx, const void *d_scalars, size_t n, void *d_a_mag, void *d_b_mag, void *d_signs);
int bppp_msm_glv_device(bppp_ctx *ctx, const void *d_scalars, const void *d_points_xy, size_t n, uint64_t out_xy[8]);

/* ---- range-proof verifier: public scalars from challenges, on the device ----------------------------------------------
 * The scalar work of verifyTRRPM (src/RangeProof/TypedReciprocal.hs:449-467) for a batch of proofs of ONE setup: makePhase2s on
 * the unit witness (:185-205), makeSharedCoeffs (:213-216), makePublicConsts (:246-274), makeBpCoeffs (:391-396) and the opening
 * scalars of TranscriptTRRP (:293-297).  bppp_trrp_create uploads the setup's static structure:
 *   per norm position i < nlen (the verifier's Phase1 list, types first when has_types):
 *     pos_kind[i]  = 0 typing | 1 inline | 2 shared, | 0x100 if the range is an output (typing) | 0x200 if it is assumed (typing)
 *     pos_range[i] = index of the range;  pos_slot[i] = index of the digit's base in the sorted base list (base map x^(3+2 slot));
 *     pos_sym[i]   = index into syms of the inline symbol, or 0xFFFFFFFF;  pos_coeff[i] = the digit coefficient b (4 limbs)
 *   per range: range_min (4 limbs, canonical mod n), range_assumed;  syms: the distinct values s whose 1/(e+s) is needed
 *   cs_slot / cs_sym [llen - 6]: base slot and symbol of each shared-digit linear weight;  public amounts: is_out, amount, type symbol.
 * flavour 0 = NL (q0 = q^2), 1 = IP (q0 = -q^2).
 * bppp_trrp_public_device: d_challenges is [batch][7][4] = (e, x, r0, q, x', r1, t) per proof; outputs, all in HBM:
 *   d_q [batch][4], d_sp [batch][4], d_pub_norm [batch][nlen][4], d_pub_lin_c [batch][llen][4] and
 *   d_init_scalars [batch][4 + nranges][4] in the proof's commitment order blCom, rCom, dmCom, mCom, nComs...
 * — exactly the arrays bppp_nl_verify_batch_device reads (its pub_lin_x is all zero for these proofs).  Asynchronous on the
 * context's stream. */
typedef struct bppp_trrp bppp_trrp;
int bppp_trrp_create(bppp_ctx *ctx, int flavour, int has_types, size_t nlen, size_t llen, size_t nranges, const uint32_t *pos_kind,
                     const uint32_t *pos_range, const uint32_t *pos_slot, const uint32_t *pos_sym, const uint64_t *pos_coeff,
                     const uint64_t *range_min, const uint32_t *range_assumed, size_t nsyms, const uint64_t *syms, const uint32_t *cs_slot,
                     const uint32_t *cs_sym, size_t npub, const uint32_t *pub_is_out, const uint64_t *pub_amount, const uint32_t *pub_sym,
                     bppp_trrp **out);
void bppp_trrp_destroy(bppp_trrp *t);
int bppp_trrp_public_device(bppp_trrp *t, size_t batch, const void *d_challenges, void *d_q, void *d_sp, void *d_pub_norm, void *d_pub_lin_c,
                            void *d_init_scalars);

/* ---- the range-proof layer end to end: encoded proofs in, accept / reject out ------------------
 * `bppp_rp` is one typed-reciprocal setup (setup, src/RangeProof/TypedReciprocal.hs:332-359) resident on the device: the ranges
 * with their digit coefficients (makeRangeData :103-120), the Phase1 layout (:133-169), the round count (optimalWitnessSize,
 * src/Bulletproof/NormArgument.hs:165-178) and the REGISTERED BASIS g, G, H — uploaded once, referenced by every later call
 * (G, H are fixed per setup, TypedReciprocal.hs:348-359).  Argument flavour: 0 = norm-linear (Bulletproof.NormArgument), 1 = inner product
 * (Bulletproof.InnerProductArgument, the CLI's default, app/Parse.hs:100).  Both have the batch verifier and the lockstep batch prover
 * (bppp_rp_prove_batch; flavour 1 proves with every commitment as an MSM over the registered ORIGINAL basis — makeNorm's basis change and
 * every point fold are carried in the scalars, same bytes as the folding route; with the handle's comb table in place the whole proof,
 * argument included, is one stream of kernels (csrc/ipb.hip), before that the field algebra of this flavour runs on the host cores).
 *
 * bppp_rp_create: `ranges` as the schema gives them (app/Parse.hs:125-172): base, min, max (plain INTEGERS in 256-bit two's complement — a minimum may be negative, examples/rec_test — max exclusive
 * as in makeRangeData), flags.  `pubs`: the public (isOutput, type, amount) triples.  `points_xy` = h : g : hs ++ gs, the stream the
 * CLI takes from getPoints (app/Main.hs:68-72, :260); at least 2 + lin_len + norm_len points (validated: on the curve).
 * `oracle_tag` (may be NULL = the reference's input) is prepended to every hashed message (domain separation for tests).
 *
 * The Fiat-Shamir oracle of these entry points is the CLI's shaOracle (app/Main.hs:64-80) computed natively: challenge n =
 * decode (SHA-256 (tag <> show n <> show (length ps) <> foldMap (show x <> show y) ps)) over the WHOLE transcript, newest first
 * (src/ZKP.hs:96-101); `show` of a coordinate = its decimal integer (parity of that text with galois-field's Show instance is
 * unpinned), `decode` = Binary (Prime p): four big-endian 64-bit words, least significant first (Encoding.hs:75-79).
 * The injectable-oracle route stays available: bppp_trrp_public_device + bppp_nl_verify_batch_device from caller-made challenges. */
#define BPPP_RP_SHARED 1u  /* isShared  */
#define BPPP_RP_OUTPUT 2u  /* isOutput  */
#define BPPP_RP_ASSUMED 4u /* isAssumed */
typedef struct bppp_rp_range { uint32_t base; uint32_t flags; uint64_t min[4]; uint64_t max[4]; } bppp_rp_range;
typedef struct bppp_rp_public { uint32_t is_output; uint32_t reserved; uint64_t type[4]; uint64_t amount[4]; } bppp_rp_public;
typedef struct bppp_rp_shape {
  size_t nranges, norm_len, lin_len, rounds, final_norm, final_lin;
  size_t coms_bytes;            /* the commitments file of one proof: sign bytes + 32 per input commitment (Encoding.hs:130-134) */
  size_t proof_bytes;           /* the proof file: final witness scalars, sign bytes, 4 + 2*rounds x coordinates (RangeProof.hs:60-66) */
  size_t challenges_per_proof;  /* 7 + rounds */
} bppp_rp_shape;
typedef struct bppp_rp bppp_rp;
int bppp_rp_create(bppp_ctx *ctx, int flavour, int has_types, const bppp_rp_range *ranges, size_t nranges, const bppp_rp_public *pubs, size_t npub,
                   const uint64_t *points_xy, size_t npoints, const char *oracle_tag, bppp_rp **out);
/* RangeProof.Binary (src/RangeProof/Binary.hs) behind the same handle: setupBRP (:143-156).  `ranges`: base must be 2 and SHARED must not
 * be set (app/Parse.hs:141-146 refuses both); `conserve` = the schema's "conserved" (inputs, outputs and `net_public` — the net public
 * amount, inputs minus outputs, a plain integer in two's complement — must balance: witnessBRP :158-166 yields a witness only then);
 * `points_xy` = [h, g, h0, h1] ++ gs, at least 4 + nrmLen points (:147-148).  Every entry point that takes a bppp_rp then serves the
 * binary protocol: bppp_rp_info (lin_len = 2; proof file = final witness scalars, then blCom, dCom and the responses;
 * challenges_per_proof = 4 + rounds: q, x, r, t), bppp_rp_verify_batch* / _shard_device (verifyBRPM :206-222 with its two oracle
 * calls, then verifyBPM) and bppp_rp_prove_batch (proveBRPM :169-204 + proveBPM in lockstep; `types` is ignored, a binary proof is
 * untyped; from its first 1024 proofs on — COMB_MIN — the handle keeps a comb table of its basis, 21.5 GB for the 4099 points of 64 outputs of
 * 64 bits, and a proof is one stream of kernels, csrc/brpprove_dev.hip; before that the field algebra runs on the host cores; same bytes).  Both sides take the round count from optimalWitnessSize (the reference's prover uses integerLog 2 nrmLen - 1, which
 * agrees wherever its own proofs verify — SURVEY.md App. D-1). */
int bppp_rp_create_binary(bppp_ctx *ctx, int flavour, int conserve, const bppp_rp_range *ranges, size_t nranges, const uint64_t net_public[4],
                          const uint64_t *points_xy, size_t npoints, const char *oracle_tag, bppp_rp **out);
void bppp_rp_destroy(bppp_rp *rp);
int bppp_rp_info(const bppp_rp *rp, bppp_rp_shape *out);
/* Tuning knobs of one handle.  Every knob has a measured default; the BPPP_RP_* environment variables of the
 * same names are read ONCE, at bppp_rp_create, as initial values — no entry point reads the environment per call.  None changes a
 * result (asserted byte for byte by the tests).
 *   COMB_MIN        first batch size — or cumulative number of proofs proved — at which the PROVER builds its fixed-base comb table
 *                   (default 1024).  The table is a persistent allocation that lives until bppp_rp_destroy.
 *   COMB_BUDGET     bytes the table may take (default 32 GiB; additionally never more than half of the HBM free at build time);
 *                   0 = never build one (bucket MSMs and the point-folding argument serve every batch).
 *   COMB_BITS       force the table's window width (4..18; skips the budget), 0 = widest that fits.
 *   SPLIT_MIN       smallest prove batch run as two half-batches in flight on a twin context (default 4096); 0 = never split.
 *   HOST_ORACLE_MAX largest batch whose Fiat-Shamir hashing runs on host cores (defaults: 8 proofs verifying, 64 proving; UINT64_MAX
 *                   restores them).  Verifying 2 .. HOST_ORACLE_MAX proofs starts, once per handle, a pool of at most 15 worker threads
 *                   that sleep between calls and end with bppp_rp_destroy.
 *   FOLD_POINTS     1 = the point-folding argument although a table exists;  HOST_ALGEBRA 1 = prover's field algebra on the host;
 *   TIMING          1 = phase times on stderr. */
#define BPPP_RP_OPT_COMB_MIN 1
#define BPPP_RP_OPT_COMB_BUDGET 2
#define BPPP_RP_OPT_COMB_BITS 3
#define BPPP_RP_OPT_SPLIT_MIN 4
#define BPPP_RP_OPT_HOST_ORACLE_MAX 5
#define BPPP_RP_OPT_FOLD_POINTS 6
#define BPPP_RP_OPT_HOST_ALGEBRA 7
#define BPPP_RP_OPT_TIMING 8
int bppp_rp_set_option(bppp_rp *rp, int option, uint64_t value);
/* Host-only helpers (no context, no GPU): the shape `setup` gives a schema (nrmLen, linLen, rounds = optimalWitnessSize, file sizes);
 * `digits` of one value in one range (src/RangeProof/TypedReciprocal.hs:125-127: greedy mixed-radix digits, the first one binary when
 * the range needs a bit; cap = capacity of out_digits); and the CLI's hash-to-field `hash = decode . SHA.hash` (app/Main.hs:64-65:
 * SHA-256, digest read through Binary (Prime p)) — hashToScalar p s = bppp_hash_to_scalar (p <> s) (app/Main.hs:83-84). */
int bppp_rp_shape_of(int flavour, int has_types, const bppp_rp_range *ranges, size_t nranges, bppp_rp_shape *out);
int bppp_rp_digits(const bppp_rp_range *range, const uint64_t amount[4], uint32_t *out_digits, size_t cap, size_t *ndigits, int *has_bit);
int bppp_hash_to_scalar(const uint8_t *data, size_t len, uint64_t out[4]);

/* Batch verification, end to end: for `batch` proofs of this setup, given as the reference's FILES — coms_files [batch][coms_bytes],
 * proof_files [batch][proof_bytes] — decodeProof (src/RangeProof.hs:68-85, src/Encoding.hs:97-128: x-only points, square roots and
 * sign selection on the device), verifyM (src/RangeProof.hs:103-105): verifyTRRPM with its three oracle calls
 * (TypedReciprocal.hs:447-467) and verifyBPM with one per round (src/Bulletproof.hs:370-378), all SHA-256 on the device, then ONE
 * combined MSM over sum_b rho_b T_b.  `seed` = 32 bytes of the VERIFIER's fresh secret randomness (never a constant
 * outside tests); rho_b = decode(SHA-256(seed <> le64(index_offset + b) <> t_b <> e_last_b <> final witness scalars of b)): bound to
 * the seed, to the proof's position in the whole job and to every byte of the proof (t and e_last are transcript hashes over all its
 * commitments and responses), never fixed to 1.  *accept = 1 iff every proof decodes and the combination is the identity.
 * proof_status (may be NULL, [batch]): BPPP_RP_VALID / _INVALID / _MALFORMED (an x coordinate with no curve point: `Nothing` in
 * the reference); when the batch is rejected the culprits are found by bisection over sub-batches (each a combined MSM).
 * challenges_out (may be NULL, [batch][7 + rounds][4]): (e, x, r0, q, x', r1, t) then the argument's challenges LAST ROUND FIRST
 * (src/Bulletproof.hs:374) — what the injected-oracle route would have been given; for parity tests.
 * combined_xy (may be NULL): the combined point sum_b rho_b MSM(T_b) itself (infinity = all zero).
 * _device: the files are already in HBM (the timed configuration of bench.py); the host variant uploads them first.
 * bppp_rp_verify_shard_device: one rank's share of a job sharded proof-per-GPU: this rank holds proofs
 * [index_offset, index_offset + batch) of the job and every rank passes the SAME seed; the ranks all-gather their 64-byte combined
 * points (RCCL) and add them with bppp_sum_points: the job verifies iff that sum is the identity and no rank saw a malformed proof.
 * Because the weights are indexed by the position in the JOB (and bound to the proofs), error terms cannot cancel between ranks: the
 * sum over ranks is the same random linear combination a single rank would have formed over all proofs.  bppp_rp_verify_batch_device
 * is the shard with index_offset = 0. */
#define BPPP_RP_VALID 0u
#define BPPP_RP_INVALID 1u
#define BPPP_RP_MALFORMED 2u
int bppp_rp_verify_batch(bppp_rp *rp, size_t batch, const uint8_t *coms_files, const uint8_t *proof_files, const uint8_t seed[32], int *accept,
                         uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy);
int bppp_rp_verify_batch_device(bppp_rp *rp, size_t batch, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32], int *accept,
                                uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy);
int bppp_rp_verify_shard_device(bppp_rp *rp, size_t batch, uint64_t index_offset, const void *d_coms_files, const void *d_proof_files, const uint8_t seed[32],
                                int *accept, uint32_t *proof_status, uint64_t *challenges_out, uint64_t *combined_xy);

/* Batch prover: `batch` proofs of this setup in lockstep — proveM of RangeProof (src/RangeProof.hs:93-97) = proveTRRPM
 * (src/RangeProof/TypedReciprocal.hs:399-446; blinding algebra src/RangeProof/Internal.hs:118-196) followed by proveBPM
 * (src/Bulletproof.hs:357-359), then encodeProof' (src/RangeProof.hs:60-66).  Per proof b: amounts / types / blinds are
 * [batch][nranges][4] (amount: a plain integer inside its range, two's complement when negative; type, blinding: canonical scalars), and the prover's randomness
 * is the CLI's: random n = decode(SHA-256(prefix_b <> show n)) for n = 0, 1, ... (hashToScalar, app/Main.hs:83-87, :189;
 * ZKPT.random, src/ZKP.hs:88-92) with prefix_b = rand_prefix[b * prefix_len ..].  The oracle is the setup's shaOracle (see above).
 * Outputs are the reference's files: coms_files [batch][coms_bytes], proof_files [batch][proof_bytes].  Every commitment is
 * computed on the device (input commitments through a fixed-base table of g, H0, H1; the four range-proof commitments of all
 * proofs and the round commitments of the argument), and so are the per-proof field algebra, the randomness and the transcript
 * hashing; the host cores extract the digits of the plain amounts.  MEMORY: at the first batch of 1024 proofs or more, or once it has proved that many in smaller batches (bppp_rp_set_option COMB_MIN), the handle
 * builds a fixed-base comb table over the setup's basis [g | H | G] and keeps it until it is destroyed — the widest window (<= 18
 * bits) whose table fits 32 GB: c = 16, 27.6 GB for the 774 points of 64by64, built in ~0.3 s (BPPP_RP_COMB_GB=<GB> changes the budget:
 * 64 GB (c = 17) and 128 GB (c = 18) measured level with it — the gathers over longer table rows cost what the fewer additions save;
 * BPPP_RP_COMB_BITS=<c> forces a width: c = 13 is 4.1 GB and ~15 % more additions; BPPP_RP_NO_COMB=1 keeps the bucket route and
 * the point-folding argument, which smaller batches use anyway).  A batch of 4096 proofs or more runs as two half-batches in
 * flight, the second on a twin handle with its own context that this handle creates and owns and that shares the table
 * (BPPP_RP_SPLIT_MIN=<n> moves the threshold, BPPP_RP_NO_SPLIT=1 disables it; a RangeProof.Binary handle splits from 1024 proofs,
 * BPPP_RP_SPLIT_MIN_BINARY); a handle serves one call at a time.  Same
 * randomness and inputs => byte-identical files to the host protocol code (bulletproofspp_amd/rangeproof.py: prove +
 * encoding.encode_proof), on every one of these routes, which the tests assert. */
int bppp_rp_prove_batch(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *types, const uint64_t *blinds, const uint8_t *rand_prefix,
                        size_t prefix_len, uint8_t *coms_files, uint8_t *proof_files);

/* ---- harness utility: pointX of getPoints (app/Main.hs:68-72) -------------------------------
 * For each candidate x (n x 4 uint64 in HBM) writes the affine point (x, y) with y the EVEN root of
 * x^3 + 7, or the infinity encoding when x^3 + 7 is a non-residue or x >= p.  (Which root
 * galois-field's `sr` returns cannot be confirmed offline — SURVEY.md 8c; even-y is this build's
 * documented choice.)  Used to make synthetic bases on the GPU. */
int bppp_lift_x_device(bppp_ctx *ctx, const void *d_x, size_t n, void *d_points_xy);

/* ---- device memory helpers (so a non-HIP host language can keep vectors resident) ---------- */
int bppp_device_alloc(bppp_ctx *ctx, size_t bytes, void **d_ptr);
int bppp_device_free(bppp_ctx *ctx, void *d_ptr);
int bppp_upload(bppp_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int bppp_download(bppp_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* Page-locked host memory for the buffers a caller hands to the host-buffer entry points (bppp_msm, bppp_rp_verify_batch,
 * bppp_rp_prove_batch, bppp_upload ...): from such a buffer the copies to the device are DMA transfers that run beside the kernels;
 * from ordinary (pageable) memory every copy first passes through a staging buffer at the host's memcpy rate.  Optional: every entry
 * point takes either kind (a Haskell binding would wrap these in a ForeignPtr with bppp_host_free as its finalizer). */
int bppp_host_alloc(bppp_ctx *ctx, size_t bytes, void **ptr);
int bppp_host_free(bppp_ctx *ctx, void *ptr);

/* ---- measurement hooks ---------------------------------------------------------------------
 * When enabled, each MSM call brackets its stages with hipEvents on the context's stream; the
 * accumulated per-stage milliseconds and launch counts can be read back (bench.py's roofline). */
#define BPPP_STAGE_DIGITS 0      /* k_digits */
#define BPPP_STAGE_SORT 1        /* k_hist .. k_scatter + bucket memset */
#define BPPP_STAGE_ACC_POINTS 2  /* k_acc_points alone: the dominant kernel (roofline) */
#define BPPP_STAGE_ACC_RECORDS 3 /* k_merge + k_merge_heavy (buckets that straddle lanes) */
#define BPPP_STAGE_REDUCE 4      /* k_reduce1 + k_reduce2 */
#define BPPP_STAGE_FINISH 5      /* window combine + copy-out */
#define BPPP_NUM_STAGES 6
int bppp_profile_enable(bppp_ctx *ctx, int on);
int bppp_profile_read(bppp_ctx *ctx, double ms[BPPP_NUM_STAGES], uint64_t *calls, int reset);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* BPPP_H */
