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
// what the caller knows about the scalar rows of a launch: nothing; every row long and of full-width scalars (blinded rows); the same in heavy / light pairs
enum { COMB_ROWS_ANY = 0, COMB_ROWS_PAIRS = 1, COMB_ROWS_DENSE = 2 };
// out[inst] = sum_i scalars[inst][i] * P_i for inst < ninst (canonical affine, infinity = zeros); scalars are canonical (< n),
// [ninst][nterms] in HBM over the first nterms registered points (0 = all T).  Asynchronous on `st`.  rows_hint: COMB_ROWS_PAIRS — instances
// 2b / 2b + 1 are a heavy / light pair of full-width rows (the light one zero on an index pattern common to all instances); COMB_ROWS_DENSE — full-width
// rows; either selects the lane-per-instance kernel for >= 512 long rows over a large table (same results, another schedule).
// d_scratch (optional, scratch_bytes): with fewer than COMB_SPLIT_BELOW instances several wavefronts share an instance and park their partial sums
// there (160 B per wavefront; comb_scratch_bytes(ninst) is enough); without it a small launch is one wavefront per instance.
int comb_msm(const CombTable *t, const uint32_t *d_scalars, size_t ninst, uint32_t *d_out_aff, hipStream_t st, int rows_hint = COMB_ROWS_ANY, size_t nterms = 0,
             uint32_t *d_scratch = nullptr, size_t scratch_bytes = 0);
// sums of GROUPS of 2^L consecutive registered points: d_scalars [ninst][T] over [g | lin (l0) | norm (n0)], out[inst][1 + q] = the q-th group's sum (lin groups,
// then norm groups; out rows are out_stride points apart, slot 0 untouched) — the level-L basis of each proof of the lockstep argument
int comb_groups(const CombTable *t, const uint32_t *d_scalars, size_t ninst, size_t l0, size_t n0, int L, uint32_t *d_out_aff, size_t out_stride, hipStream_t st);
// many instances of a few terms each over the FIRST nterms registered points, one lane per instance: d_scalars [ninst][nterms]
int comb_lanes(const CombTable *t, const uint32_t *d_scalars, size_t nterms, size_t ninst, uint32_t *d_out_aff, hipStream_t st);
static constexpr size_t COMB_SPLIT_BELOW = 8192;      // launches of fewer instances are split into about that many wavefronts
static constexpr size_t COMB_ROWS_WAVES = 16384;      // k_comb_msm_rows (lane = instance): about that many wavefronts, 64 partial sums each
inline size_t comb_scratch_bytes(size_t ninst) { return ninst < COMB_SPLIT_BELOW ? (COMB_SPLIT_BELOW + 64 * (ninst < 1024 ? ninst : 1024) + ninst) * 160 : 0; }
// scratch of a launch with a rows_hint (the argument's rounds, the blinded phase rows): room for the partial sums of either route
inline size_t comb_rows_scratch_bytes(size_t ninst) { const size_t a = comb_scratch_bytes(ninst), b = (64 * COMB_ROWS_WAVES + 2 * ninst + 128) * 160; return a > b ? a : b; }

}  // namespace bppp
