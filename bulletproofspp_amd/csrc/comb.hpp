// comb.hpp — fixed-base comb over a setup's basis: every signed c-bit digit of a scalar is ONE table addition (csrc/comb.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "ctx.hpp"

namespace bppp {

struct CombTable {
  bppp_ctx *ctx;
  size_t T;            // registered points
  int c, W, D;         // window bits, windows (c * W >= 257), multiples per window D = 2^(c-1)
  uint32_t *tab;       // [W][T][D] affine (16 u32 each): tab[w][i][d - 1] = d * 2^(c w) * P_i
  size_t bytes;
};

// d_points: T affine points in HBM (copied into the table; the caller keeps ownership of the array).  window_bits = 0: chosen so
// that the table stays under `budget_bytes`.
int comb_create(bppp_ctx *ctx, const uint32_t *d_points, size_t T, int window_bits, size_t budget_bytes, CombTable **out);
void comb_destroy(CombTable *t);
// out[inst] = sum_i scalars[inst][i] * P_i for inst < ninst (canonical affine, infinity = zeros); scalars are canonical (< n),
// [ninst][nterms] in HBM over the first nterms registered points (0 = all T).  Asynchronous on `st`.  heavy_first: instances
// 2b / 2b + 1 are a heavy / light pair (dispatch order only).
// d_scratch (optional, scratch_bytes): with fewer than COMB_SPLIT_BELOW instances several wavefronts share an instance and park their partial sums
// there (160 B per wavefront; comb_scratch_bytes(ninst) is enough); without it a small launch is one wavefront per instance.
int comb_msm(const CombTable *t, const uint32_t *d_scalars, size_t ninst, uint32_t *d_out_aff, hipStream_t st, bool heavy_first = false, size_t nterms = 0,
             uint32_t *d_scratch = nullptr, size_t scratch_bytes = 0);
// many instances of a few terms each over the FIRST nterms registered points, one lane per instance: d_scalars [ninst][nterms]
int comb_lanes(const CombTable *t, const uint32_t *d_scalars, size_t nterms, size_t ninst, uint32_t *d_out_aff, hipStream_t st);
static constexpr size_t COMB_SPLIT_BELOW = 8192;      // launches of fewer instances are split into about that many wavefronts
inline size_t comb_scratch_bytes(size_t ninst) { return ninst < COMB_SPLIT_BELOW ? (COMB_SPLIT_BELOW + 64 * (ninst < 1024 ? ninst : 1024) + ninst) * 160 : 0; }

}  // namespace bppp
