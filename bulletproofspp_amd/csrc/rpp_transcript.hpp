// rpp_transcript.hpp — the Fiat-Shamir transcript and the prover's randomness of B proofs in lockstep (csrc/rpp_transcript.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "rp_internal.hpp"

namespace bppp {

// one oracle call of the protocol: `points` new commitments go in front of the transcript, `count` (1 .. 3) challenges come out and land in
// ch[b][first_slot ...] (first_slot < 7) or in es[b] (first_slot = 7: a round challenge of the argument)
struct RppCall { uint32_t points, count, first_slot; };

struct RppTranscript {
  bppp_rp *rp = nullptr;
  size_t B = 0;
  std::vector<RppCall> calls;            // the range-proof layer's calls, then one (2, 1, 7) per round of the argument
  bool host = false;                     // a handful of proofs: the hashing runs on the host cores, the new points and the challenges cross PCIe
  uint8_t *text = nullptr; uint32_t *tstart = nullptr; void *hdrs = nullptr; uint32_t *ch = nullptr, *es = nullptr;     // device; ch [B][7][8], es [B][8]
  uint32_t stride = 0, tend = 0;         // text capacity per proof (rp->D.text_stride), right-aligned with 16 bytes of slack at the end
  std::vector<std::vector<std::string>> groups; std::vector<size_t> np;      // host mode: every proof's transcript so far
  static size_t hdr_bytes(size_t ncalls);                                    // device bytes `d_hdrs` needs
  // uploads the headers of all calls (tag <> show n <> show (length ps)) and resets every proof's text; synchronises the stream once
  int begin(bppp_rp *rp, size_t batch, const std::vector<RppCall> &layer_calls, size_t rounds, bool host_oracle, uint8_t *d_text, uint32_t *d_tstart, void *d_hdrs,
            uint32_t *d_ch, uint32_t *d_es);
  // oracle call number `call_index` of every proof: pts_dev [B][points] affine, in the order the reference conses them.  Asynchronous on the
  // context's stream in device mode.
  int call(const uint32_t *pts_dev, size_t call_index);
};

// rnd[b][c] = hashToScalar (prefix_b <> show c), c < nd (app/Main.hs:83-87); d_prefix [batch][prefix_len].  Asynchronous.
int rpp_draws(bppp_ctx *ctx, const uint8_t *d_prefix, size_t prefix_len, size_t batch, size_t nd, uint32_t *d_rnd);

void rpp_host_oracle(const std::string &tag, std::vector<std::string> &groups, size_t &npoints, const uint64_t *pts, size_t m, int count, uint64_t *out);   // csrc/rpprove.hip

}  // namespace bppp
