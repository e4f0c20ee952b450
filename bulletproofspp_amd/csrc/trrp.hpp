// trrp.hpp — the static structure of a typed-reciprocal setup as it sits on the device (uploaded by bppp_trrp_create): shared by the
// verifier's public-scalar kernel (csrc/trrp.hip) and the prover's phase kernels (csrc/rpprove_dev.hip).
#pragma once
#include <stdint.h>
#include "ctx.hpp"

namespace bppp {
struct TrrpDims { uint32_t nlen, llen, nr, nsyms, npub, has_types, flavour; };
// position kinds (Phase1 constructors, src/RangeProof/TypedReciprocal.hs:56-60)
static constexpr uint32_t K_TYPING = 0, K_INLINE = 1, F_IO = 1u << 8, F_IA = 1u << 9, NO_SYM = 0xFFFFFFFFu;
static constexpr int TRRP_MAX_SLOTS = 16;      // distinct digit bases of one setup (base map x^3, x^5, ...)
}  // namespace bppp

struct bppp_trrp {
  bppp_ctx *ctx;
  bppp::TrrpDims D;
  uint32_t *pos_kind, *pos_range, *pos_slot, *pos_sym, *pos_coeff, *range_min, *range_assumed, *syms, *cs_slot, *cs_sym, *pub_is_out, *pub_amount, *pub_sym;
};
