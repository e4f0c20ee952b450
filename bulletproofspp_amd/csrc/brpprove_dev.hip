// brpprove_dev.hip — RangeProof.Binary's prover for B proofs in lockstep as ONE stream of kernels: field algebra, randomness and transcript on
// the device, every commitment a comb MSM over the setup's basis [g | h0 h1 | G], the argument device-resident (csrc/nlb.hip fixed-basis mode or
// csrc/ipb.hip, by the setup's flavour).  The host extracts the binary digits of the plain amounts, uploads them and waits once, at the end.
//
// proveBRPM (src/RangeProof/Binary.hs:169-204), restated with a leading batch dimension (the host-core version — csrc/rpprove.hip's
// prove_batch_binary — stays as the cross-check behind BPPP_RP_HOST_ALGEBRA; both write the same bytes, tests/test_gpu_native_binary.py):
//   k_brpp_row_d     dWit = (sBl; lBl0, 0; ds) (:171-178): the two blinding draws and the digits, straight into the MSM input
//   k_brpp_phase2    one workgroup per proof: makePublicConsts' norm vector p_i = x^(2(j+1)) b_i q0^-(i+1) - 1/2 (:73-98), the blinding vector bls,
//                    makePolyTerms' constant and linear coefficient of |bls + T (ds + pub)|^2_q (src/RangeProof/Internal.hs:69-80), lin1 =
//                    (sBl - 2 bl1) / r, the row of blWit = (bl0; blBl, lin1; bls) (:179-189)
//   k_brp_public     (csrc/rp.hip, the verifier's kernel) t^2 pubSc, t pubNrm, [0, r t], 2 t^2 inputCoeffs (:127-129) once t is known
//   k_brpp_combine   bpWit = blWit + t (pub' + dWit + 2 t sum_j ic_j nWit_j) (:190-201), written where the argument starts from
// Draw order (ZKPT.random counter, src/ZKP.hs:88-92): 0 sBl, 1 lBl0, 2 .. nlen + 1 bls, nlen + 2 blBl.
#include <string.h>
#include <vector>
#include "fr26.hip.h"
#include "modinv.hip.h"
#include "rp_internal.hpp"
#include "brp.hpp"
#include "comb.hpp"
#include "rpprove_dev.hpp"
#include "rpp_transcript.hpp"

namespace bppp {

// rows [B][T], T = 3 + nlen
__global__ void __launch_bounds__(256) k_brpp_row_d(BrpDims D, uint32_t batch, uint32_t nd, const uint32_t *__restrict__ rnd, const uint8_t *__restrict__ bits,
                                                    uint32_t *__restrict__ rows) {
  const uint32_t T = 3 + D.nlen;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (uint64_t)batch * T) return;
  const uint32_t b = (uint32_t)(g / T), j = (uint32_t)(g % T);
  fe v = fe_zero();
  if (j < 2) v = fe_load(rnd + ((size_t)b * nd + j) * 8);
  else if (j >= 3 && j - 3 < D.nlive) v.v[0] = bits[(size_t)b * D.nlive + (j - 3)];
  fe_store(rows + g * 8, v);
}

BPPP_DI fr brpp_pow(fr base, uint32_t e) { fr a = fr_one(); while (e) { if (e & 1u) a = fr_mul(a, base); base = fr_sqr(base); e >>= 1; } return a; }

// aux [B][2]: bl0, lin1 (the combination needs them again)
__global__ void __launch_bounds__(256) k_brpp_phase2(BrpDims D, uint32_t nd, const uint32_t *__restrict__ pos_range, const uint32_t *__restrict__ pos_coeff,
                                                     const uint32_t *__restrict__ rnd, const uint8_t *__restrict__ bits, const uint32_t *__restrict__ ch,
                                                     uint32_t *__restrict__ row_bl, uint32_t *__restrict__ aux) {
  extern __shared__ uint32_t lds[];               // [nr] x^(2(j+1)), then [256][2] partial sums
  uint32_t *x2s = lds, *part = lds + (size_t)D.nr * 8;
  const uint32_t b = blockIdx.x, l = threadIdx.x, T = 3 + D.nlen;
  const uint32_t *c = ch + (size_t)b * 56;
  const fe q8 = fe_load(c), x8 = fe_load(c + 8), r8 = fe_load(c + 16);
  fe q0 = fe_sqr<1>(q8);                          // qPowers': powers' (q^2) for the norm-linear argument (NormArgument.hs:148),
  if (D.flavour) q0 = fe_neg<1>(q0);              // powers' (-q^2) for the inner-product one (InnerProductArgument.hs:231)
  const fe q0i = fe_modinv<1>(q0);                // every lane the same division steps: no divergence
  const fr xx = fr_sqr(fr_from_fe(x8));
  {
    fr xj = brpp_pow(xx, l + 1);
    const fr step = brpp_pow(xx, 256);
    for (uint32_t j = l; j < D.nr; j += 256) { const fe v = fr_to_fe(xj); for (int k = 0; k < 8; k++) x2s[j * 8 + k] = v.v[k]; xj = fr_mul(xj, step); }
  }
  __syncthreads();
  fe half8 = fe_zero();                           // (n + 1) / 2
  { const fe n = fr_modulus(); uint32_t carry = 1; fe tt;
    for (int i = 0; i < 8; i++) { const uint64_t s_ = (uint64_t)n.v[i] + carry; tt.v[i] = (uint32_t)s_; carry = (uint32_t)(s_ >> 32); }
    for (int i = 0; i < 8; i++) half8.v[i] = (tt.v[i] >> 1) | (i < 7 ? tt.v[i + 1] << 31 : carry << 31); }
  const fr q0r = fr_from_fe(q0), q0ir = fr_from_fe(q0i), half = fr_from_fe(half8);
  fr qp = brpp_pow(q0r, l + 1), qi = brpp_pow(q0ir, l + 1), bl0 = fr_zero(), bl1 = fr_zero();
  const fr qs = brpp_pow(q0r, 256), qis = brpp_pow(q0ir, 256);
  const uint32_t *r = rnd + (size_t)b * nd * 8;
  uint32_t *row = row_bl + (size_t)b * T * 8;
  for (uint32_t i = l; i < D.nlen; i += 256) {
    const fe bl8 = fe_load(r + (size_t)(2 + i) * 8);
    const fr bl = fr_from_fe(bl8);
    bl0 = fr_addr(bl0, fr_mul(qp, fr_sqr(bl)));
    if (i < D.nlive) {
      fe xv; for (int k = 0; k < 8; k++) xv.v[k] = x2s[pos_range[i] * 8 + k];
      fr pv = fr_sub<1>(fr_mul(fr_mul(fr_from_fe(xv), fr_load(pos_coeff + (size_t)i * 8)), qi), half);      // magnitude 3
      if (bits[(size_t)b * D.nlive + i]) pv = fr_add(pv, fr_one());                                          // ds_i + pub_i
      bl1 = fr_addr(bl1, fr_mul(qp, fr_mul(bl, pv)));
    }
    fe_store(row + (size_t)(3 + i) * 8, bl8);
    qp = fr_mul(qp, qs); qi = fr_mul(qi, qis);
  }
  {
    const fe a0 = fr_to_fe(bl0), a1 = fr_to_fe(bl1);
    for (int k = 0; k < 8; k++) { part[(l * 2) * 8 + k] = a0.v[k]; part[(l * 2 + 1) * 8 + k] = a1.v[k]; }
  }
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)l < d)
      for (int s = 0; s < 2; s++) {
        fe a, o;
        for (int k = 0; k < 8; k++) { a.v[k] = part[(l * 2 + s) * 8 + k]; o.v[k] = part[((l + d) * 2 + s) * 8 + k]; }
        a = fe_add<1>(a, o);
        for (int k = 0; k < 8; k++) part[(l * 2 + s) * 8 + k] = a.v[k];
      }
    __syncthreads();
  }
  if (l == 0) {
    fe s0, s1;
    for (int k = 0; k < 8; k++) { s0.v[k] = part[k]; s1.v[k] = part[8 + k]; }
    const fe r_inv = fe_modinv<1>(r8);
    const fe lin1 = fe_mul<1>(r_inv, fe_sub<1>(fe_load(r), fe_dbl<1>(s1)));       // (sBl - 2 bl1) / r
    fe_store(row, s0); fe_store(row + 8, fe_load(r + (size_t)(2 + D.nlen) * 8)); fe_store(row + 16, lin1);
    fe_store(aux + (size_t)b * 16, s0); fe_store(aux + (size_t)b * 16 + 8, lin1);
  }
}

// a_s [B], a_lx [B][2], a_nx [B][nlen] from: rnd, bits, aux (bl0, lin1), the inputs (v, bl) and k_brp_public's outputs for the proof's t —
// p_sp = t^2 pubSc, p_norm = t pubNrm, p_init[2 + j] = 2 t^2 ic_j
__global__ void __launch_bounds__(256) k_brpp_combine(BrpDims D, uint32_t batch, uint32_t nd, const uint32_t *__restrict__ rnd, const uint8_t *__restrict__ bits,
                                                      const uint32_t *__restrict__ aux, const uint32_t *__restrict__ in_sc, const uint32_t *__restrict__ ch,
                                                      const uint32_t *__restrict__ p_sp, const uint32_t *__restrict__ p_norm, const uint32_t *__restrict__ p_init,
                                                      uint32_t *__restrict__ a_s, uint32_t *__restrict__ a_lx, uint32_t *__restrict__ a_nx) {
  const uint32_t T = 3 + D.nlen;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (uint64_t)batch * T) return;
  const uint32_t b = (uint32_t)(g / T), j = (uint32_t)(g % T);
  const uint32_t *r = rnd + (size_t)b * nd * 8;
  const fe t8 = fe_load(ch + (size_t)b * 56 + 48);
  if (j >= 3) {
    const uint32_t i = j - 3;
    fr v = fr_add(fr_load(r + (size_t)(2 + i) * 8), fr_load(p_norm + ((size_t)b * D.nlen + i) * 8));
    if (i < D.nlive && bits[(size_t)b * D.nlive + i]) v = fr_add(v, fr_from_fe(t8));
    fr_store(a_nx + ((size_t)b * D.nlen + i) * 8, v);
    return;
  }
  if (j == 2) { fe_store(a_lx + ((size_t)b * 2 + 1) * 8, fe_load(aux + (size_t)b * 16 + 8)); return; }
  // j = 0: bl0 + t^2 pubSc + t sBl + sum_j (2 t^2 ic_j) v_j;   j = 1: blBl + t lBl0 + sum_j (2 t^2 ic_j) bl_j
  const fr t = fr_from_fe(t8);
  fr acc = fr_mul(t, fr_load(r + (size_t)j * 8));
  for (uint32_t k = 0; k < D.nr; k++)
    acc = fr_addr(acc, fr_mul(fr_load(p_init + ((size_t)b * (2 + D.nr) + 2 + k) * 8), fr_load(in_sc + (((size_t)b * D.nr + k) * 3 + j) * 8)));
  if (j == 0) fr_store(a_s + (size_t)b * 8, fr_add(fr_add(acc, fr_load(aux + (size_t)b * 16)), fr_load(p_sp + (size_t)b * 8)));
  else fr_store(a_lx + (size_t)b * 16, fr_add(acc, fr_load(r + (size_t)(2 + D.nlen) * 8)));
}

int brp_device_prove(bppp_rp *rp, const BrpHostInputs &in, BrpOutputs &out) {
  bppp_ctx *ctx = rp->ctx;
  hipStream_t st = ctx->stream;
  const bppp_rps::Setup &S = rp->st;
  const bppp_brp_tabs *tb = rp->btabs;
  if (!rp->comb || !tb) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: the device prover needs the comb table of the setup");
  const BrpDims D = tb->D;
  const size_t B = in.batch, nr = S.rds.size(), nlen = S.nlen, nlive = S.nlive, k = S.rounds, T = 3 + nlen, nd = nlen + 3;
  if (S.llen != 2 || rp->comb->T != T) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: binary setup and comb table disagree");
  uint32_t *in_sc = nullptr, *in_pt = nullptr, *rnd = nullptr, *row_d = nullptr, *row_bl = nullptr, *aux = nullptr, *ch = nullptr, *es = nullptr, *tstart = nullptr,
           *ptbuf = nullptr, *a_s = nullptr, *a_q = nullptr, *a_lx = nullptr, *a_nx = nullptr, *p_sp = nullptr, *p_norm = nullptr, *p_cs = nullptr, *p_init = nullptr,
           *d_resp = nullptr, *d_com = nullptr, *cscratch = nullptr;
  uint8_t *bits = nullptr, *text = nullptr, *prefix = nullptr, *hdrs = nullptr;
  const uint32_t stride = rp->D.text_stride;
  for (int pass = 0; pass < 2; pass++) {
    Carver cv(pass ? rp->pwork : nullptr, rp->pwork_bytes);
    in_sc = cv.take<uint32_t>(B * nr * 24); in_pt = cv.take<uint32_t>(B * nr * 16); bits = cv.take<uint8_t>(B * nlive + 16);
    rnd = cv.take<uint32_t>(B * nd * 8); row_d = cv.take<uint32_t>(B * T * 8); row_bl = cv.take<uint32_t>(B * T * 8); aux = cv.take<uint32_t>(B * 16);
    ch = cv.take<uint32_t>(B * 56); es = cv.take<uint32_t>(B * 8); tstart = cv.take<uint32_t>(B); ptbuf = cv.take<uint32_t>(B * (1 + nr) * 16);
    a_s = cv.take<uint32_t>(B * 8); a_q = cv.take<uint32_t>(B * 8); a_lx = cv.take<uint32_t>(B * 16); a_nx = cv.take<uint32_t>(B * nlen * 8);
    p_sp = cv.take<uint32_t>(B * 8); p_norm = cv.take<uint32_t>(B * nlen * 8); p_cs = cv.take<uint32_t>(B * 16); p_init = cv.take<uint32_t>(B * (2 + nr) * 8);
    text = cv.take<uint8_t>(B * (size_t)stride + 64); prefix = cv.take<uint8_t>(B * in.prefix_len + 16); hdrs = cv.take<uint8_t>(RppTranscript::hdr_bytes(2 + k) + 16);
    d_resp = cv.take<uint32_t>(k * B * 32 + 16); d_com = cv.take<uint32_t>(2 * B * 16 + 16);
    cscratch = cv.take<uint32_t>(comb_rows_scratch_bytes(B) / 4 + 16);
    if (!pass) { int rc = rpp_ensure_pwork(rp, cv.off); if (rc) return rc; }
  }
  BPPP_HIP(ctx, hipMemcpyAsync(in_sc, in.in_sc, B * nr * 96, hipMemcpyHostToDevice, st));
  if (nlive) BPPP_HIP(ctx, hipMemcpyAsync(bits, in.bits, B * nlive, hipMemcpyHostToDevice, st));
  if (in.prefix_len) BPPP_HIP(ctx, hipMemcpyAsync(prefix, in.prefix, B * in.prefix_len, hipMemcpyHostToDevice, st));
  // the oracle calls of proveBRPM: oracle' (dCom : nComs) -> q x r (:179), oracle [blCom] -> t (:189); small batches hash on the host cores
  RppTranscript tr;
  int rc = tr.begin(rp, B, {RppCall{(uint32_t)(1 + nr), 3, 0}, RppCall{1, 1, 6}}, k, B <= rp->opt.host_oracle_prove, text, tstart, hdrs, ch, es); if (rc) return rc;
  auto comb = [&](const uint32_t *rows, uint32_t *dst, int hint) -> int {
    int r_ = comb_msm(rp->comb, rows, B, dst, st, hint, 0, cscratch, comb_rows_scratch_bytes(B));
    return r_ ? fail(ctx, r_, bppp_last_error(rp->comb->ctx)) : BPPP_OK;
  };
  uint32_t *c_d = d_com, *c_bl = d_com + B * 16;
  rc = rpp_draws(ctx, prefix, in.prefix_len, B, nd, rnd); if (rc) return rc;
  { const uint64_t n = (uint64_t)B * T; k_brpp_row_d<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(D, (uint32_t)B, (uint32_t)nd, rnd, bits, row_d); }
  BPPP_HIP(ctx, hipGetLastError());
  rc = rpp_commit_inputs(rp, in_sc, B * nr, in_pt); if (rc) return rc;                 // scalarRPW' (Internal.hs:56-57): v g + bl h0
  BPPP_HIP(ctx, hipMemcpyAsync(out.input_coms, in_pt, B * nr * 64, hipMemcpyDeviceToHost, st));
  rc = comb(row_d, c_d, COMB_ROWS_ANY); if (rc) return rc;                   // bits and a few blinders: most scalars are 0 or 1
  BPPP_HIP(ctx, hipMemcpy2DAsync(ptbuf, (1 + nr) * 64, c_d, 64, 64, B, hipMemcpyDeviceToDevice, st));
  BPPP_HIP(ctx, hipMemcpy2DAsync(ptbuf + 16, (1 + nr) * 64, in_pt, nr * 64, nr * 64, B, hipMemcpyDeviceToDevice, st));
  rc = tr.call(ptbuf, 0); if (rc) return rc;
  const size_t lds2 = (nr + 2 * 256) * 32;
  if (lds2 > 160 * 1024) return fail(ctx, BPPP_ERR_ARG, "rp_prove_batch: too many ranges for the device prover");
  if (lds2 > 64 * 1024) BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_brpp_phase2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  k_brpp_phase2<<<dim3((unsigned)B), dim3(256), lds2, st>>>(D, (uint32_t)nd, tb->pos_range, tb->pos_coeff, rnd, bits, ch, row_bl, aux);
  BPPP_HIP(ctx, hipGetLastError());
  rc = comb(row_bl, c_bl, COMB_ROWS_DENSE); if (rc) return rc;
  rc = tr.call(c_bl, 1); if (rc) return rc;
  rc = brp_public_device(rp, B, ch, a_q, p_sp, p_norm, p_cs, p_init); if (rc) return rc;
  { const uint64_t n = (uint64_t)B * T;
    k_brpp_combine<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(D, (uint32_t)B, (uint32_t)nd, rnd, bits, aux, in_sc, ch, p_sp, p_norm, p_init, a_s, a_lx, a_nx); }
  BPPP_HIP(ctx, hipGetLastError());
  std::vector<uint64_t> hcom;
  rc = rpp_argument_stream(rp, tr, 2, B, a_s, a_q, a_nx, p_cs, a_lx, d_resp, out.resp, out.wit_norm, out.wit_lin, d_com, 2 * B, hcom); if (rc) return rc;
  memcpy(out.c_d, hcom.data(), B * 64); memcpy(out.c_bl, hcom.data() + B * 8, B * 64);
  return BPPP_OK;
}

}  // namespace bppp
