// ipb.hpp — the inner-product argument for B proofs in lockstep with device-resident state (csrc/ipb.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "comb.hpp"
#include "rpp_transcript.hpp"

namespace bppp {

// device bytes ipb_prove_stream needs in `work`
size_t ipb_work_bytes(size_t B, size_t nlen, size_t llen);
// proveBPM of the inner-product flavour (src/Bulletproof.hs:357-359 over src/Bulletproof/InnerProductArgument.hs) for B arguments of one
// shape, k rounds queued back to back on the context's stream; every commitment a comb MSM over the setup's ORIGINAL basis [g | H | G].
// in (device, canonical): psv [B] (the PSV scalar), rr [B] (makeNorm's r), nrm [B][nlen], lc / lx [B][llen]; the oracle calls of the rounds
// are tr's calls first_call ...  out (device): resp [k][B][2][16] (L, R per round, in round order), wn [B][fn], wl [B][fl]; *d_flag: one word,
// non-zero when some round challenge was zero (read it after the stream has drained).
int ipb_prove_stream(bppp_ctx *ctx, const CombTable *comb, RppTranscript &tr, size_t first_call, size_t B, size_t nlen, size_t llen, size_t k, size_t fn, size_t fl,
                     const uint32_t *d_psv, const uint32_t *d_rr, const uint32_t *d_nrm, const uint32_t *d_lc, const uint32_t *d_lx, void *work, size_t work_bytes,
                     uint32_t *d_resp, uint32_t *d_wn, uint32_t *d_wl, uint32_t **d_flag_out);

}  // namespace bppp
