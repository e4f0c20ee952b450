// brp.hpp — the static structure of a RangeProof.Binary setup as it sits on the device (uploaded by brp_build_tables, csrc/rp.hip): shared by the
// verifier's public-scalar kernel (k_brp_public, csrc/rp.hip) and the prover's phase kernels (csrc/brpprove_dev.hip).
#pragma once
#include <stdint.h>

namespace bppp {
struct BrpDims { uint32_t nlen, nlive, nr, conserve, flavour; };     // nlive: norm positions of the ranges that are not assumed (the rest of nlen stays zero)
}  // namespace bppp

// pos_range / pos_coeff [nlive]: the range a live position belongs to and its coefficient 2^j resp. b_n (setupBRP, src/RangeProof/Binary.hs:143-156);
// range_min [nr], range_flags [nr] (bit 0: output, bit 1: assumed); net_public: one scalar
struct bppp_brp_tabs {
  uint32_t *pos_range = nullptr, *pos_coeff = nullptr, *range_min = nullptr, *range_flags = nullptr, *net_public = nullptr;
  bppp::BrpDims D{};
};
