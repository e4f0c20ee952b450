// rpprove_dev.hpp — interface between the two halves of the batch range-proof prover: csrc/rpprove.hip (entry point, witness digits,
// commitments, encoding; the host-algebra reference path) and csrc/rpprove_dev.hip (field algebra and transcript on the device).
#pragma once
#include <stddef.h>
#include <string>
#include <vector>
#include <stdint.h>
#include "rp_internal.hpp"

namespace bppp {

struct PDims { uint32_t nlen, llen, nr, T, nd, has_types, maxb; };     // T = 1 + llen + nlen (one commitment row), nd = random scalars per proof,
// maxb = entries of the per-proof reciprocal table 1 / (e + s), s < maxb: the power of two at or above the setup's widest digit base (16 .. 2048)

struct RppHostInputs {
  size_t batch;
  const uint64_t *in_sc;     // [batch][nr][3][4]: amount, type, blinding as field elements
  const uint32_t *dig;       // [batch][nlen]: the digit of every norm position (unused at typing positions)
  const uint32_t *mul;       // [batch][nlen]: the inline multiplicity (0 elsewhere)
  const uint32_t *mss;       // [batch][llen - 6]: the shared multiplicities
  const uint8_t *prefix; size_t prefix_len;
};
struct RppOutputs {          // host arrays
  uint64_t *input_coms;      // [batch][nr][8]
  uint64_t *c_dm, *c_m, *c_r, *c_bl;   // [batch][8] each
  uint64_t *resp;            // [batch][rounds][16]: (X, R), last round first
  uint64_t *wit_norm, *wit_lin;        // [batch][fn][4], [batch][fl][4]
};

int rpp_device_prove(bppp_rp *rp, const RppHostInputs &in, RppOutputs &out);

// RangeProof.Binary (csrc/brpprove_dev.hip): proveBRPM + proveBPM of B proofs as one stream of kernels over the handle's comb table
struct BrpHostInputs {
  size_t batch;
  const uint64_t *in_sc;     // [batch][nr][3][4]: amount, blinding, 0 as field elements (scalarRPW', src/RangeProof/Internal.hs:56-57)
  const uint8_t *bits;       // [batch][nlive]: the binary digit of every live norm position (makeDigits, src/RangeProof/Binary.hs:56-69)
  const uint8_t *prefix; size_t prefix_len;
};
struct BrpOutputs {          // host arrays
  uint64_t *input_coms;      // [batch][nr][8]
  uint64_t *c_d, *c_bl;      // [batch][8] each
  uint64_t *resp;            // [batch][rounds][16]: (X, R) resp. (L, R), last round first
  uint64_t *wit_norm, *wit_lin;
};
int brp_device_prove(bppp_rp *rp, const BrpHostInputs &in, BrpOutputs &out);

struct RppTranscript;
// proveBPM of the setup's flavour behind the range-proof phases, device-resident (csrc/rpprove_dev.hip)
int rpp_argument_stream(bppp_rp *rp, RppTranscript &tr, size_t first_call, size_t B, const uint32_t *a_s, const uint32_t *a_q, const uint32_t *a_nx, const uint32_t *a_lc,
                        const uint32_t *a_lx, uint32_t *d_resp, uint64_t *resp_out, uint64_t *wn_out, uint64_t *wl_out, const uint32_t *d_extra, size_t extra_points,
                        std::vector<uint64_t> &extra_out);
// verifyBRPM's public scalars for a batch (k_brp_public, csrc/rp.hip): the binary prover reuses them as the TR prover reuses k_trrp_public
int brp_public_device(bppp_rp *rp, size_t batch, const uint32_t *ch, uint32_t *q, uint32_t *sp, uint32_t *pub_norm, uint32_t *pub_lin_c, uint32_t *init_sc);

// provided by rpprove.hip
int rpp_ensure_pwork(bppp_rp *rp, size_t bytes);
int rpp_commit_inputs(bppp_rp *rp, const uint32_t *d_in_sc, size_t n, uint32_t *d_out);                 // asynchronous on the context's stream
int rpp_commit_rows(bppp_rp *rp, const uint32_t *d_rows, size_t nrows, uint64_t *host_out);              // synchronises the stream
int nlb_create_impl(bppp_ctx *ctx, size_t batch, const uint64_t *s, const uint64_t *g_xy, const uint64_t *q, const uint64_t *norm_x, const uint64_t *norm_g_xy,
                    size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy, size_t llen, bppp_nlb **out, bool on_device,
                    const struct CombTable *comb);

void rpp_host_oracle(const std::string &tag, std::vector<std::string> &groups, size_t &npoints, const uint64_t *pts, size_t m, int count, uint64_t *out);   // csrc/rpprove.hip
bool nlb_fixed_basis(const bppp_nlb *o);
int nlb_round_commit_dev(bppp_nlb *o, uint32_t *d_XR);
int nlb_round_collapse_dev(bppp_nlb *o, const uint32_t *d_es);

}  // namespace bppp

extern "C" {
void bppp_nlb_destroy(bppp_nlb *nlb);
int bppp_nlb_round_commit(bppp_nlb *nlb, uint64_t *sX, uint64_t *X_xy, uint64_t *sR, uint64_t *R_xy);
int bppp_nlb_round_collapse(bppp_nlb *nlb, const uint64_t *es);
int bppp_nlb_get_witness(bppp_nlb *nlb, uint64_t *norm_w, uint64_t *lin_w, uint64_t *s);
}
