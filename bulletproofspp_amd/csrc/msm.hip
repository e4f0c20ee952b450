// msm.hip — signed-digit Pippenger multi-scalar multiplication for gfx950.
//
// Replaces FastInnerProduct.innerProduct (src/Commitment.hs:325-335) reached through `commit`
// (src/Commitment.hs:416-417).  The reference is a 256-row bit-serial Straus loop; this is a
// different algorithm that yields the same group element:
//
//   1. k_digits      reduceScalar's sign fold (Commitment.hs:276-279, :366) + signed c-bit window
//                    recode; coalesced 32-B scalar loads, u16 digits out.
//   2. k_hist / k_scan* / k_scatter
//                    counting sort of (window, |digit|) keys: the bucket histogram and cursors are
//                    staged in LDS (<= 128 KiB per workgroup), so HBM sees only coalesced streams.
//   3. k_acc_points / k_merge / k_merge_heavy
//                    load-balanced bucket accumulation: every lane owns exactly L consecutive
//                    sorted entries (4 bytes each: sign | point index; the bucket of a position follows from the
//                    bucket offsets start[], it is not stored), sums runs of one bucket in an XYZZ register
//                    accumulator with mixed adds and stores complete buckets; a bucket that
//                    straddles lanes is finished by k_merge (one lane per bucket) or, when it spans
//                    many lanes (skewed scalars), by one wavefront in k_merge_heavy.
//   4. k_reduce_marg / k_reduce_tail
//                    sum_m m*B_m per window by MARGINAL SUMS (m = LO*hi + lo: plain row and column sums, every lane busy,
//                    then two short weighted sums); k_reduce1/2 (per-lane running sums + wavefront suffix scans) for windows
//                    under 256 buckets, k_reduce_groups for thousands of small windows.
//   5. window combine: Horner over <= 65 window sums — on the host for one MSM (a 256-doubling
//                    dependency chain), in k_window_combine for batches.
//
// Integer / carry-chain work on the VALU; nothing here is a dense contraction, so no MFMA.
#include <algorithm>
#include <vector>
#include "ctx.hpp"
#include "ec.hip.h"
#include "hostmath.hpp"

namespace bppp {


struct RecodeK { uint32_t k[9]; };

// ------------------------------------------------------------------------------------------------
// 1. digits
// Window widths: the first `acnt` windows are c bits wide, the others c - 1 (balanced windows, make_plan; acnt = W: all c).  A digit of a
// narrow window is stored re-biased by 2^(c-2), so that every consumer reads  stored - 2^(c-1)  as the signed digit whatever the width.
__global__ void __launch_bounds__(256) k_digits(const uint32_t *__restrict__ scalars, uint64_t total, uint32_t n, uint32_t stride, int c,
                                                int W, int acnt, RecodeK K, uint16_t *__restrict__ dig,
                                                unsigned long long *__restrict__ negmask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = i < total;
  bool neg = false;
  if (valid) {
    fe s = fe_load(scalars + 8 * i);
    fe t, tmp;
    raw_sub(t, fr_modulus(), s);               // n - s
    neg = raw_sub(tmp, t, s) != 0;             // t < s  <=>  s > n - s   (reduceScalar, Commitment.hs:279)
    uint32_t sp[9];
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      cy += (uint64_t)(neg ? t.v[k] : s.v[k]) + K.k[k];
      sp[k] = (uint32_t)cy; cy >>= 32;
    }
    sp[8] = (uint32_t)cy + K.k[8];
    uint32_t inst = (uint32_t)(i / n), j = (uint32_t)(i % n);
    size_t base = (size_t)inst * W * stride + j;
    for (int w = 0; w < W; w++) {
      const int cw = w < acnt ? c : c - 1;
      const uint32_t mask = (1u << cw) - 1u, bias = w < acnt ? 0u : (1u << (c - 2));
      dig[base + (size_t)w * stride] = (uint16_t)((sp[0] & mask) + bias);
#pragma unroll
      for (int k = 0; k < 8; k++) sp[k] = (sp[k] >> cw) | (sp[k + 1] << (32 - cw));
      sp[8] >>= cw;
    }
  }
  unsigned long long m = __ballot(valid && neg);
  if ((threadIdx.x & 63) == 0 && (i - (i & 63)) < total) negmask[i >> 6] = m;
}

// ------------------------------------------------------------------------------------------------
// 2. counting sort by (instance, window, |digit|)
// Both sort kernels read 8 consecutive u16 digits per lane with ONE 16-byte load (n is padded to a multiple of 8 in the
// digit buffer's row stride), so each lane has 8 independent LDS atomics / stores in flight instead of a dependent
// load -> atomic -> store chain per entry (the scattered stores are latency-, not bandwidth-bound).
__global__ void k_hist(const uint16_t *__restrict__ dig, uint32_t n, uint32_t stride, int c, int CH, uint32_t *__restrict__ blockhist) {
  extern __shared__ uint32_t lh[];
  const int M = 1 << (c - 1);
  const uint32_t nbw = blockIdx.x, ch = blockIdx.y;
  for (int t = threadIdx.x; t < M; t += blockDim.x) lh[t] = 0;
  __syncthreads();
  uint32_t per = (((n + CH - 1) / CH) + 7u) & ~7u, lo = ch * per, hi = min(n, lo + per);
  const uint16_t *d = dig + (size_t)nbw * stride;
  for (uint32_t j0 = lo + threadIdx.x * 8; j0 < hi; j0 += blockDim.x * 8) {
    uint4 pk = *reinterpret_cast<const uint4 *>(d + j0);
    uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int v = (int)((w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu) - M;
      if (j0 + k < hi && v) atomicAdd(&lh[(v < 0 ? -v : v) - 1], 1u);
    }
  }
  __syncthreads();
  uint32_t *out = blockhist + ((size_t)nbw * CH + ch) * M;
  for (int t = threadIdx.x; t < M; t += blockDim.x) out[t] = lh[t];
}

// per flat bucket: exclusive prefix over the chunks (in place) and the bucket total
__global__ void k_chunk_prefix(uint32_t *__restrict__ blockhist, int M, int CH, uint64_t FB, uint32_t *__restrict__ count) {
  uint64_t fb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fb >= FB) return;
  uint64_t nbw = fb / M, mb = fb % M;
  uint32_t run = 0;
  for (int ch = 0; ch < CH; ch++) {
    size_t at = ((size_t)nbw * CH + ch) * M + mb;
    uint32_t t = blockhist[at];
    blockhist[at] = run;
    run += t;
  }
  count[fb] = run;
}

// three-kernel exclusive scan over `count` (tiles of 4096)
static constexpr int SCAN_TILE = 4096;
__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ tile_sums) {
  __shared__ uint32_t ws[4];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int k = 0; k < SCAN_TILE / 256; k++) {
    uint64_t i = base + k * 256 + threadIdx.x;
    if (i < n) s += in[i];
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t *__restrict__ tile_sums, uint32_t ntiles, uint32_t *__restrict__ total_out) {
  // single block: exclusive scan of tile sums in place
  __shared__ uint32_t part[1024];
  uint32_t per = (ntiles + 1023) / 1024, lo = threadIdx.x * per, hi = min(ntiles, lo + per);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += tile_sums[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = threadIdx.x >= (uint32_t)d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (uint32_t i = lo; i < hi; i++) { uint32_t t = tile_sums[i]; tile_sums[i] = run; run += t; }
  if (threadIdx.x == 1023) *total_out = part[1023];
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ in, uint64_t n, const uint32_t *__restrict__ tile_off,
                                                    uint32_t *__restrict__ out) {
  // block-level exclusive scan of one tile (16 per thread), plus the tile offset
  __shared__ uint32_t wsum[4];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t v[16], s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  uint32_t inc = s;
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += t; }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += wsum[w];
  uint32_t run = tile_off[blockIdx.x] + woff + inc - s;
#pragma unroll
  for (int k = 0; k < 16; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

__global__ void k_scatter(const uint16_t *__restrict__ dig, const unsigned long long *__restrict__ negmask, uint32_t n, uint32_t stride, int c,
                          int CH, int W, const uint32_t *__restrict__ blockhist, const uint32_t *__restrict__ start,
                          uint32_t *__restrict__ sorted, uint32_t flat_stride) {
  extern __shared__ uint32_t lh[];
  const int M = 1 << (c - 1);
  const uint32_t nbw = blockIdx.x, ch = blockIdx.y;
  const uint32_t *bh = blockhist + ((size_t)nbw * CH + ch) * M;
  // flat_stride != 0: the windows of an instance share one bucket set and the entry indexes the table row of its window
  const uint32_t *st = start + (size_t)(flat_stride ? nbw / W : nbw) * M;
  const uint32_t idx_base = flat_stride ? (nbw % W) * flat_stride : 0u;
  for (int t = threadIdx.x; t < M; t += blockDim.x) lh[t] = st[t] + bh[t];
  __syncthreads();
  uint32_t per = (((n + CH - 1) / CH) + 7u) & ~7u, lo = ch * per, hi = min(n, lo + per);
  const uint16_t *d = dig + (size_t)nbw * stride;
  const uint32_t inst = nbw / W;
  for (uint32_t j0 = lo + threadIdx.x * 8; j0 < hi; j0 += blockDim.x * 8) {
    uint4 pk = *reinterpret_cast<const uint4 *>(d + j0);
    uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
    uint64_t flat0 = (uint64_t)inst * n + j0;
    // sign bits of the 8 scalars: they may straddle two 64-bit words when inst*n is not a multiple of 8
    unsigned long long m0 = negmask[flat0 >> 6], m1 = negmask[(flat0 + 7) >> 6];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int v = (int)((w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu) - M;
      if (j0 + k < hi && v) {
        uint32_t mb = (v < 0 ? -v : v) - 1;
        uint64_t flat = flat0 + k;
        unsigned long long mw = ((flat >> 6) == (flat0 >> 6)) ? m0 : m1;
        uint32_t sneg = (uint32_t)((mw >> (flat & 63)) & 1ull);
        uint32_t sg = (v < 0 ? 1u : 0u) ^ sneg;
        uint32_t pos = atomicAdd(&lh[mb], 1u);
        sorted[pos] = (sg << 31) | (idx_base + j0 + k);   // 4 bytes: the bucket is implied by start[] (position -> bucket), not stored
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3. load-balanced accumulation
// Lane g owns sorted positions [g*L, (g+1)*L).  A bucket b = [start[b], start[b]+count[b]) that lies
// inside one lane is summed and stored there.  A bucket spanning lanes g0 < g1 leaves partial sums:
//   lane g0 (run starts inside the lane)      -> tail slot 2*g0+1
//   lanes g0 < g <= g1 (run continues from g-1) -> head slot 2*g
// which k_merge (one lane per bucket, short spans) or k_merge_heavy (one wavefront per bucket)
// add up.  Slots are addressed from start/count alone: no keys are stored, no compaction needed.
__global__ void __launch_bounds__(256) k_acc_points(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ start, uint32_t FB,
                                                    const uint32_t *__restrict__ points, uint32_t n, uint32_t WM, int shared_pts,
                                                    int L, uint64_t G, uint32_t *__restrict__ buckets, uint32_t *__restrict__ rec_pt) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t total = start[FB];
  const uint64_t pos0 = g * (uint64_t)L;
  if (pos0 >= total) return;
  const uint32_t pos1 = (uint32_t)min((uint64_t)total, pos0 + L);
  // the bucket that holds position pos0: the largest fb with start[fb] <= pos0 (it is non-empty: start[fb + 1] > pos0)
  uint32_t lo = 0, hi = FB - 1;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (start[mid] <= (uint32_t)pos0) lo = mid; else hi = mid - 1;
  }
  uint32_t cur = lo, end = start[cur + 1];
  const bool from_prev = start[cur] < (uint32_t)pos0;         // the first run continues one that began in an earlier lane
  bool first = true;
  uint32_t e = sorted[pos0];
  xyzz acc = xyzz_inf();
  for (uint32_t p = (uint32_t)pos0; p < pos1; p++) {
    const uint32_t e_next = (p + 1 < pos1) ? sorted[p + 1] : 0u;
    if (p == end) {
      // the finished run ends inside this lane: head partial if it came from the previous lane, else complete
      if (first && from_prev) xyzz_store(rec_pt + (2 * g) * XYZZ_WORDS, acc);
      else xyzz_store(buckets + (size_t)cur * XYZZ_WORDS, acc);
      first = false; acc = xyzz_inf();
      cur++; end = start[cur + 1];
      if (end <= p) {                                         // an empty bucket — there may be thousands in a row (the unused half of a narrow
        uint32_t lo2 = cur + 1, hi2 = FB - 1;                 // window, an empty carry window): bisect for the bucket that holds p
        while (lo2 < hi2) {
          const uint32_t mid = lo2 + (hi2 - lo2 + 1) / 2;
          if (start[mid] <= p) lo2 = mid; else hi2 = mid - 1;
        }
        cur = lo2; end = start[cur + 1];
      }
    }
    const uint32_t idx = e & 0x7FFFFFFFu;
    const bool sg = (e >> 31) & 1u;
    // shared_pts: 1 = one point array for every instance, 0 = one per instance, d >= 2 = one per d consecutive instances
    size_t pidx = shared_pts == 1 ? (size_t)idx : (size_t)((cur / WM) / (shared_pts ? shared_pts : 1)) * n + idx;
    aff P = aff_cneg(aff_load(points + pidx * 16), sg);
    xyzz_madd(acc, P);
    e = e_next;
  }
  const bool hi_part = first && from_prev, ti = end > pos1;   // the last run goes on in the next lane
  if (hi_part) xyzz_store(rec_pt + (2 * g) * XYZZ_WORDS, acc);
  else if (ti) xyzz_store(rec_pt + (2 * g + 1) * XYZZ_WORDS, acc);
  else xyzz_store(buckets + (size_t)cur * XYZZ_WORDS, acc);
}

static constexpr int MERGE_SERIAL_MAX = 8;   // longer spans go to the wavefront-cooperative kernels
static constexpr int HEAVY_CHUNK = 256;      // partial sums per wavefront in k_merge_heavy

// partial t of a bucket whose first lane is g0: t = 0 is g0's tail slot, t >= 1 the head slot of lane g0+t
BPPP_DI const uint32_t *partial_ptr(const uint32_t *rec_pt, uint64_t g0, uint64_t t) {
  return rec_pt + (t == 0 ? 2 * g0 + 1 : 2 * (g0 + t)) * XYZZ_WORDS;
}

__global__ void __launch_bounds__(256) k_merge(const uint32_t *__restrict__ start, const uint32_t *__restrict__ count, uint64_t FB, int L,
                                               const uint32_t *__restrict__ rec_pt, uint32_t *__restrict__ buckets,
                                               uint2 *__restrict__ heavy_items, uint4 *__restrict__ heavy_buckets, uint32_t *__restrict__ heavy_count) {
  uint64_t fb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fb >= FB) return;
  uint32_t cnt = count[fb];
  if (!cnt) return;
  uint32_t s = start[fb];
  uint64_t g0 = s / (uint32_t)L, g1 = (uint64_t)(s + cnt - 1) / (uint32_t)L;
  if (g0 == g1) return;                       // summed and stored by its lane
  uint32_t np = (uint32_t)(g1 - g0 + 1);      // number of partial sums
  if (np > MERGE_SERIAL_MAX + 1) {
    // skewed scalars / a narrow top window: hand the bucket to wavefronts, HEAVY_CHUNK partials each
    uint32_t nch = (np + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
    uint32_t base = atomicAdd(&heavy_count[0], nch);
    for (uint32_t j = 0; j < nch; j++) heavy_items[base + j] = make_uint2((uint32_t)fb, j);
    if (nch > 1) heavy_buckets[atomicAdd(&heavy_count[1], 1u)] = make_uint4((uint32_t)fb, base, nch, 0u);
    return;
  }
  xyzz acc = xyzz_load(partial_ptr(rec_pt, g0, 0));
  for (uint32_t t = 1; t < np; t++) {
    xyzz p = xyzz_load(partial_ptr(rec_pt, g0, t));
    xyzz_add(acc, p);
  }
  xyzz_store(buckets + fb * XYZZ_WORDS, acc);
}

// one wavefront per (heavy bucket, chunk): lanes stride over the chunk's partials, then a shuffle tree
__global__ void __launch_bounds__(64) k_merge_heavy(const uint32_t *__restrict__ start, const uint32_t *__restrict__ count, int L,
                                                    const uint32_t *__restrict__ rec_pt, uint32_t *__restrict__ buckets,
                                                    const uint2 *__restrict__ heavy_items, const uint32_t *__restrict__ heavy_count,
                                                    uint32_t *__restrict__ chunk_sums) {
  const uint32_t nitems = heavy_count[0], lane = threadIdx.x;
  for (uint32_t h = blockIdx.x; h < nitems; h += gridDim.x) {
    uint2 it = heavy_items[h];
    uint32_t fb = it.x, s = start[fb], cnt = count[fb];
    uint64_t g0 = s / (uint32_t)L, g1 = (uint64_t)(s + cnt - 1) / (uint32_t)L;
    uint32_t np = (uint32_t)(g1 - g0 + 1);
    uint32_t t0 = it.y * HEAVY_CHUNK, t1 = min(np, t0 + HEAVY_CHUNK);
    xyzz acc = xyzz_inf();
    for (uint32_t t = t0 + lane; t < t1; t += 64) {
      xyzz p = xyzz_load(partial_ptr(rec_pt, g0, t));
      xyzz_add(acc, p);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      xyzz o = xyzz_shfl_down(acc, d);
      if ((int)lane + d < 64) xyzz_add(acc, o);
    }
    if (lane == 0) {
      if (np <= HEAVY_CHUNK) xyzz_store(buckets + (size_t)fb * XYZZ_WORDS, acc);
      else xyzz_store(chunk_sums + (size_t)h * XYZZ_WORDS, acc);
    }
  }
}
// buckets with more than one chunk: one wavefront adds the chunk sums
__global__ void __launch_bounds__(64) k_merge_heavy2(const uint4 *__restrict__ heavy_buckets, const uint32_t *__restrict__ heavy_count,
                                                     const uint32_t *__restrict__ chunk_sums, uint32_t *__restrict__ buckets) {
  const uint32_t nb = heavy_count[1], lane = threadIdx.x;
  for (uint32_t h = blockIdx.x; h < nb; h += gridDim.x) {
    uint4 hb = heavy_buckets[h];
    xyzz acc = xyzz_inf();
    for (uint32_t j = lane; j < hb.z; j += 64) {
      xyzz p = xyzz_load(chunk_sums + (size_t)(hb.y + j) * XYZZ_WORDS);
      xyzz_add(acc, p);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      xyzz o = xyzz_shfl_down(acc, d);
      if ((int)lane + d < 64) xyzz_add(acc, o);
    }
    if (lane == 0) xyzz_store(buckets + (size_t)hb.x * XYZZ_WORDS, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// 4. bucket reduction: per window  sum_{m=1..M} m * B_m
// One wavefront handles 64*Lw consecutive buckets: lane l takes buckets [ (wave*64+l)*Lw, +Lw ).
__global__ void __launch_bounds__(64) k_reduce1(const uint32_t *__restrict__ buckets, int M, int Lw, int WPW, uint32_t *__restrict__ red) {
  const uint32_t nbw = blockIdx.x, wave = blockIdx.y, lane = threadIdx.x;
  uint32_t t = wave * 64 + lane;                 // lane index within the window
  xyzz run = xyzz_inf(), acc = xyzz_inf();
  if ((uint64_t)t * Lw < (uint64_t)M) {
    const uint32_t *b = buckets + ((size_t)nbw * M + (size_t)t * Lw) * XYZZ_WORDS;
    for (int k = Lw - 1; k >= 0; k--) {          // running sum from the top: acc = sum (k+1) * B_k
      xyzz B = xyzz_load(b + (size_t)k * XYZZ_WORDS);
      xyzz_add(run, B);
      xyzz_add(acc, run);
    }
  }
  // inclusive suffix scan of the lane sums S_l over the wavefront
  xyzz suf = run;
  for (int d = 1; d < 64; d <<= 1) {
    xyzz o = xyzz_shfl_down(suf, d);
    if ((int)lane + d < 64) xyzz_add(suf, o);
  }
  // V_l = acc_l + Lw * (l >= 1 ? suf_l : 0);  sum_l l*S_l = sum_{l>=1} suf_l
  xyzz v = (lane >= 1) ? suf : xyzz_inf();
  for (int k = Lw; k > 1; k >>= 1) v = xyzz_dbl(v);
  xyzz_add(v, acc);
  for (int d = 32; d >= 1; d >>= 1) {
    xyzz o = xyzz_shfl_down(v, d);
    if ((int)lane + d < 64) xyzz_add(v, o);
  }
  if (lane == 0) {
    uint32_t *o = red + ((size_t)nbw * WPW + wave) * (2 * XYZZ_WORDS);
    xyzz_store(o, v);          // A_wave
    xyzz_store(o + XYZZ_WORDS, suf);   // S_wave (lane 0's inclusive suffix = whole-wave sum)
  }
}
// across the wavefronts of a window: total = sum_j A_j + (64*Lw) * sum_{j>=1} suffix_j(S)
__global__ void __launch_bounds__(64) k_reduce2(const uint32_t *__restrict__ red, int Lw, int WPW, uint32_t *__restrict__ winsum) {
  const uint32_t nbw = blockIdx.x, lane = threadIdx.x;
  xyzz A = xyzz_inf(), S = xyzz_inf();
  if ((int)lane < WPW) {
    const uint32_t *r = red + ((size_t)nbw * WPW + lane) * (2 * XYZZ_WORDS);
    A = xyzz_load(r); S = xyzz_load(r + XYZZ_WORDS);
  }
  xyzz suf = S;
  for (int d = 1; d < 64; d <<= 1) {
    xyzz o = xyzz_shfl_down(suf, d);
    if ((int)lane + d < 64) xyzz_add(suf, o);
  }
  xyzz v = (lane >= 1) ? suf : xyzz_inf();
  for (int k = 64 * Lw; k > 1; k >>= 1) v = xyzz_dbl(v);
  xyzz_add(v, A);
  for (int d = 32; d >= 1; d >>= 1) {
    xyzz o = xyzz_shfl_down(v, d);
    if ((int)lane + d < 64) xyzz_add(v, o);
  }
  if (lane == 0) xyzz_store(winsum + (size_t)nbw * XYZZ_WORDS, v);
}

// ---- 4'. bucket reduction by marginal sums (windows with M >= 256 buckets).
// A dependent chain of point additions runs at ONE wavefront's instruction rate (~6 us per addition), so the running-sum scheme
// above costs its depth: ~23 chained additions in k_reduce1 at 2 wavefronts per SIMD plus ~21 in k_reduce2 on 17 wavefronts.
// Here the bucket index splits as i = hi * LO + lo (weight i + 1 = LO * hi + (lo + 1)):
//     sum_i (i + 1) B_i  =  LO * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo,     R_hi = sum_lo B,   C_lo = sum_hi B
// R and C are PLAIN sums: k_reduce_marg gives every lane S consecutive (rows) or strided (columns) buckets — no scan, every
// lane busy, one wavefront per SIMD — and finishes them with a short segmented tree.  The two weighted sums left are over HI and
// LO (<= 256) elements: k_reduce_tail, one element per lane (suffix scan + tree, ~18 chained additions).  The factor LO is not
// applied on the device at all: the window combine, which doubles c times per window anyway, takes (W1, W2) and does
// r = (r * 2^(c-a) + W1) * 2^a + W2.
struct MargGeom { int a, LO, HI, PR, PC, SR, SC; uint32_t row_tiles, col_tiles; };

__global__ void __launch_bounds__(64) k_reduce_marg(const uint32_t *__restrict__ buckets, int M, MargGeom Gm, uint32_t *__restrict__ R, uint32_t *__restrict__ C) {
  const uint32_t nbw = blockIdx.y, lane = threadIdx.x;
  const uint32_t *wb = buckets + (size_t)nbw * M * XYZZ_WORDS;
  xyzz acc = xyzz_inf();
  if (blockIdx.x < Gm.row_tiles) {
    const uint32_t r = blockIdx.x * 64 + lane, hi = r / (uint32_t)Gm.PR, part = r % (uint32_t)Gm.PR;
    const bool act = hi < (uint32_t)Gm.HI;
    if (act) {
      const uint32_t *b = wb + ((size_t)hi * Gm.LO + (size_t)part * Gm.SR) * XYZZ_WORDS;
      acc = xyzz_load(b);
      for (int k = 1; k < Gm.SR; k++) { xyzz B = xyzz_load(b + (size_t)k * XYZZ_WORDS); xyzz_add(acc, B); }
    }
    for (int d = Gm.PR >> 1; d >= 1; d >>= 1) {
      xyzz o = xyzz_shfl_down(acc, d);
      if ((int)part + d < Gm.PR) xyzz_add(acc, o);
    }
    if (act && part == 0) xyzz_store(R + ((size_t)nbw * Gm.HI + hi) * XYZZ_WORDS, acc);
  } else {
    const uint32_t q = (blockIdx.x - Gm.row_tiles) * 64 + lane, lo = q / (uint32_t)Gm.PC, part = q % (uint32_t)Gm.PC;
    const bool act = lo < (uint32_t)Gm.LO;
    if (act) {
      const uint32_t *b = wb + ((size_t)part * Gm.SC * Gm.LO + lo) * XYZZ_WORDS;
      acc = xyzz_load(b);
      for (int k = 1; k < Gm.SC; k++) { xyzz B = xyzz_load(b + (size_t)k * Gm.LO * XYZZ_WORDS); xyzz_add(acc, B); }
    }
    for (int d = Gm.PC >> 1; d >= 1; d >>= 1) {
      xyzz o = xyzz_shfl_down(acc, d);
      if ((int)part + d < Gm.PC) xyzz_add(acc, o);
    }
    if (act && part == 0) xyzz_store(C + ((size_t)nbw * Gm.LO + lo) * XYZZ_WORDS, acc);
  }
}

// one workgroup per (window, R | C) — blockIdx.y = 0: R, 1: C — of at most four wavefronts, one per SIMD of its CU: a dependent
// chain of point additions runs at one wavefront's instruction rate, and two chains on one SIMD would each run at half of it.
// Both kinds are launched with 64 * ceil(LO / 64) threads; the spare wavefronts of an R workgroup hold infinity.
__global__ void __launch_bounds__(256) k_reduce_tail(const uint32_t *__restrict__ R, const uint32_t *__restrict__ C, MargGeom Gm, uint32_t *__restrict__ winsum2) {
  __shared__ __attribute__((aligned(16))) uint32_t tot[4 * XYZZ_WORDS];
  const uint32_t nbw = blockIdx.x, lane = threadIdx.x & 63, gw = threadIdx.x >> 6, gn = blockDim.x >> 6;
  const bool isC = blockIdx.y != 0;
  const uint32_t n = isC ? (uint32_t)Gm.LO : (uint32_t)Gm.HI, idx = gw * 64 + lane;
  xyzz suf = xyzz_inf();
  if (idx < n) suf = xyzz_load((isC ? C + ((size_t)nbw * Gm.LO + idx) * XYZZ_WORDS : R + ((size_t)nbw * Gm.HI + idx) * XYZZ_WORDS));
  for (int d = 1; d < 64; d <<= 1) {             // inclusive suffix scan inside the wavefront
    xyzz o = xyzz_shfl_down(suf, d);
    if ((int)lane + d < 64) xyzz_add(suf, o);
  }
  if (lane == 0) xyzz_store(tot + gw * XYZZ_WORDS, suf);
  __syncthreads();
  for (uint32_t w = gw + 1; w < gn; w++) { xyzz t = xyzz_load(tot + w * XYZZ_WORDS); xyzz_add(suf, t); }     // later wavefronts of the group
  // R: sum_hi hi * R_hi = sum_{hi >= 1} suffix(hi);   C: sum_lo (lo + 1) * C_lo = sum_{lo >= 0} suffix(lo)
  xyzz v = (!isC && idx == 0) ? xyzz_inf() : suf;
  for (int d = 32; d >= 1; d >>= 1) {
    xyzz o = xyzz_shfl_down(v, d);
    if ((int)lane + d < 64) xyzz_add(v, o);
  }
  __syncthreads();
  if (lane == 0) xyzz_store(tot + gw * XYZZ_WORDS, v);
  __syncthreads();
  if (lane == 0 && gw == 0) {
    for (uint32_t w = 1; w < gn; w++) { xyzz t = xyzz_load(tot + w * XYZZ_WORDS); xyzz_add(v, t); }
    xyzz_store(winsum2 + ((size_t)nbw * 2 + (isC ? 1 : 0)) * XYZZ_WORDS, v);
  }
}

// Many small windows (batched MSMs of a few hundred terms: M <= 256 buckets per window, thousands of windows): a whole
// wavefront per window leaves most lanes idle and pays two 6-step scans.  Here a GROUP of G lanes (G | 64, G <= M) owns one
// window: lane g sums its Lw = M/G buckets serially, the G lane sums go through a log2(G)-step suffix scan inside the
// group, and the window sum goes straight to `winsum` (no k_reduce2 pass).
__global__ void __launch_bounds__(64) k_reduce_groups(const uint32_t *__restrict__ buckets, int M, int G, uint32_t NB, uint32_t *__restrict__ winsum) {
  const uint32_t lane = threadIdx.x, gl = lane & (uint32_t)(G - 1);
  const uint32_t nbw = (blockIdx.x * 64u + lane) / (uint32_t)G;
  const int Lw = M / G;
  xyzz run = xyzz_inf(), acc = xyzz_inf();
  if (nbw < NB) {
    const uint32_t *b = buckets + ((size_t)nbw * M + (size_t)gl * Lw) * XYZZ_WORDS;
    for (int k = Lw - 1; k >= 0; k--) {
      xyzz B = xyzz_load(b + (size_t)k * XYZZ_WORDS);
      xyzz_add(run, B);
      xyzz_add(acc, run);
    }
  }
  xyzz suf = run;                                // inclusive suffix scan of the lane sums inside the group
  for (int d = 1; d < G; d <<= 1) {
    xyzz o = xyzz_shfl_down(suf, d);
    if ((int)gl + d < G) xyzz_add(suf, o);
  }
  xyzz v = (gl >= 1) ? suf : xyzz_inf();         // sum_g g * S_g = sum_{g>=1} suf_g, times Lw
  if (G > 1) for (int k = Lw; k > 1; k >>= 1) v = xyzz_dbl(v);
  xyzz_add(v, acc);
  for (int d = G >> 1; d >= 1; d >>= 1) {
    xyzz o = xyzz_shfl_down(v, d);
    if ((int)gl + d < G) xyzz_add(v, o);
  }
  if (gl == 0 && nbw < NB) xyzz_store(winsum + (size_t)nbw * XYZZ_WORDS, v);
}

// 5. batched window combine: one lane per MSM instance
// `a` > 0: every window comes as two points (W1, W2) with window value 2^a * W1 + W2 (marginal-sum reduction)
__global__ void __launch_bounds__(64) k_window_combine(const uint32_t *__restrict__ winsum, int W, int c, int a, uint32_t batch, uint32_t *__restrict__ out_aff) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  xyzz r = xyzz_inf();
  for (int w = W - 1; w >= 0; w--) {
    if (a) {
      for (int k = 0; k < c - a; k++) r = xyzz_dbl(r);
      xyzz t1 = xyzz_load(winsum + ((size_t)b * W + w) * 2 * XYZZ_WORDS);
      xyzz_add(r, t1);
      for (int k = 0; k < a; k++) r = xyzz_dbl(r);
      xyzz t2 = xyzz_load(winsum + (((size_t)b * W + w) * 2 + 1) * XYZZ_WORDS);
      xyzz_add(r, t2);
    } else {
      for (int k = 0; k < c; k++) r = xyzz_dbl(r);
      xyzz t = xyzz_load(winsum + ((size_t)b * W + w) * XYZZ_WORDS);
      xyzz_add(r, t);
    }
  }
  aff_store(out_aff + (size_t)b * 16, xyzz_to_aff(r));
}

// ------------------------------------------------------------------------------------------------
// 0. ONE small MSM (a few thousand terms, one instance) in one or two launches: the general pipeline is ~16 launches and four dependency
// chains deep (0.28 ms of kernels for the 858-term MSM of a single range-proof verification, whatever the window); here one workgroup per
// (window, slice of the terms) does all of it — sign fold + digit of its window for every scalar of the slice, a counting sort of the
// entries in LDS, M = 2^(c-1) buckets with 256 / M lanes each (a shuffle tree joins them), and the running-sum reduction sum (m + 1) B_m
// as a suffix scan + tree over M lanes.  c = 6: 43 windows, ~17 dependent point operations deep (0.13 ms).  k_msm_small_join adds the
// slices of a window (one wavefront per window); the window sums go to the host's Horner combine like the general path's.
static constexpr uint32_t MSM_SMALL_MAX = 4096;            // terms per slice (LDS: 6 bytes each)
static constexpr size_t MSM_SMALL_DEFAULT_MAX = 8192;      // terms per MSM on this route unless BPPP_MSM_SMALL_MAX says otherwise: every (window, slice) workgroup
                                                           // pays the ~17-operation reduction chain with most lanes idle, so the general pipeline wins from ~2^14 terms
                                                           // (benchmarks/sweep_small_msm.py: 858 terms 0.36 -> 0.21 ms, 4096 0.38 -> 0.27, 8192 0.40 -> 0.33, 22016 0.45 -> 0.43, 65536 0.52 -> 0.78)
__global__ void __launch_bounds__(256) k_msm_small(const uint32_t *__restrict__ scalars_all, const uint32_t *__restrict__ points_all, uint32_t n_all, uint32_t slice_len,
                                                   int c, int acnt, RecodeK K, uint32_t *__restrict__ winsum) {
  // slice blockIdx.y of the terms: its own bucket set, its own partial window sum (k_msm_small_join adds the slices of a window)
  const uint32_t j0 = blockIdx.y * slice_len, n = min(slice_len, n_all - j0);
  const uint32_t *scalars = scalars_all + (size_t)j0 * 8, *points = points_all + (size_t)j0 * 16;
  extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
  const int M = 1 << (c - 1);
  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  uint32_t *hist = sm, *start = hist + M, *sorted = start + M + 1;
  uint16_t *key = (uint16_t *)(sorted + n);
  uint32_t *bsum = sm + (((size_t)(2 * M + 1) + n + (n + 1) / 2 + 3) & ~(size_t)3);    // [M + 4] bucket sums, then the wavefronts' results
  for (int t = tid; t < M; t += 256) hist[t] = 0;
  __syncthreads();
  // balanced windows (as make_plan's): windows [0, acnt) are c bits wide, the others c - 1 and use half of the M buckets — with uniform widths
  // the top window (3 real bits at c = 6) had 4 buckets of n / 4 entries and its workgroup set the kernel's time
  const int cw = (int)w < acnt ? c : c - 1;
  const uint32_t bit0 = (int)w < acnt ? (uint32_t)c * w : (uint32_t)(c * acnt + (c - 1) * ((int)w - acnt)), mask = (1u << cw) - 1u;
  const int half = 1 << (cw - 1);
  for (uint32_t j = tid; j < n; j += 256) {
    const fe s_ = fe_load(scalars + (size_t)j * 8);
    fe t, tmp;
    raw_sub(t, fr_modulus(), s_);
    const bool neg = raw_sub(tmp, t, s_) != 0;                     // s > n - s (reduceScalar, Commitment.hs:279)
    uint32_t sp[10];
    uint64_t cy = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { cy += (uint64_t)(neg ? t.v[q] : s_.v[q]) + K.k[q]; sp[q] = (uint32_t)cy; cy >>= 32; }
    sp[8] = (uint32_t)cy + K.k[8]; sp[9] = 0;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) if ((bit0 >> 5) == (uint32_t)q) { lo = sp[q]; hi = sp[q + 1]; }
    const uint32_t sh = bit0 & 31u;
    const uint32_t d = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & mask;
    const int v = (int)d - half;
    uint16_t kx = 0xFFFFu;
    if (v) {
      const uint32_t mb = (uint32_t)(v < 0 ? -v : v) - 1u;
      kx = (uint16_t)((mb << 1) | (((v < 0) != neg) ? 1u : 0u));
      atomicAdd(&hist[mb], 1u);
    }
    key[j] = kx;
  }
  __syncthreads();
  if (tid < 64) {                                                   // exclusive prefix of hist[M] (M <= 128): lane owns M / 64 or one bucket
    const int per = M > 64 ? M / 64 : 1;
    uint32_t s0 = 0;
    for (int q = 0; q < per; q++) { const int b = (int)lane * per + q; if (b < M) s0 += hist[b]; }
    uint32_t inc = s0;
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(inc, dd, 64); if ((int)lane >= dd) inc += o; }
    uint32_t run = inc - s0;
    for (int q = 0; q < per; q++) { const int b = (int)lane * per + q; if (b < M) { const uint32_t h = hist[b]; start[b] = run; hist[b] = run; run += h; } }
    if (lane == 63) start[M] = inc;
  }
  __syncthreads();
  for (uint32_t j = tid; j < n; j += 256) {
    const uint16_t kx = key[j];
    if (kx != 0xFFFFu) { const uint32_t pos = atomicAdd(&hist[kx >> 1], 1u); sorted[pos] = ((uint32_t)(kx & 1u) << 31) | j; }
  }
  __syncthreads();
  // ---- buckets: LPB = 256 / M lanes per bucket, entries dealt round-robin, then a shuffle tree inside the group
  const uint32_t LPB = 256u / (uint32_t)M, b = tid / LPB, sub = tid % LPB;
  xyzz acc = xyzz_inf();
  {
    const uint32_t p1 = start[b + 1];
    for (uint32_t p = start[b] + sub; p < p1; p += LPB) {
      const uint32_t e = sorted[p];
      xyzz_madd(acc, aff_cneg(aff_load(points + (size_t)(e & 0x7FFFFFFFu) * 16), (e >> 31) != 0));
    }
  }
  for (uint32_t dd = LPB >> 1; dd >= 1; dd >>= 1) {
    const xyzz o = xyzz_shfl_down(acc, (int)dd);
    if (sub + dd < LPB) xyzz_add(acc, o);
  }
  if (sub == 0) xyzz_store(bsum + (size_t)b * XYZZ_WORDS, acc);
  __syncthreads();
  // ---- sum_m (m + 1) B_m = sum over m of the inclusive suffix sums: suffix scan + tree over the M lanes of each wavefront
  const uint32_t nw = ((uint32_t)M + 63u) / 64u;
  if (tid < 64u * nw) {
    const uint32_t m = tid;
    xyzz suf = m < (uint32_t)M ? xyzz_load(bsum + (size_t)m * XYZZ_WORDS) : xyzz_inf();
    const int Mw = M < 64 ? M : 64;                                 // live lanes of a wavefront
    for (int dd = 1; dd < Mw; dd <<= 1) {
      const xyzz o = xyzz_shfl_down(suf, dd);
      if ((int)lane + dd < Mw) xyzz_add(suf, o);
    }
    xyzz tot = suf;                                                 // lane 0: the whole wavefront's plain sum
    xyzz v = suf;
    for (int dd = Mw >> 1; dd >= 1; dd >>= 1) {
      const xyzz o = xyzz_shfl_down(v, dd);
      if ((int)lane + dd < Mw) xyzz_add(v, o);
    }
    if (lane == 0) {                                                // results go past the M bucket sums the other wavefront may still be loading
      xyzz_store(bsum + (size_t)(M + 2 * (tid >> 6)) * XYZZ_WORDS, v);
      xyzz_store(bsum + (size_t)(M + 2 * (tid >> 6) + 1) * XYZZ_WORDS, tot);
    }
  }
  __syncthreads();
  if (tid == 0) {
    xyzz r = xyzz_load(bsum + (size_t)M * XYZZ_WORDS);              // A_0
    for (uint32_t j = 1; j < nw; j++) {                             // + A_j + 64 j T_j
      xyzz a = xyzz_load(bsum + (size_t)(M + 2 * j) * XYZZ_WORDS), t = xyzz_load(bsum + (size_t)(M + 2 * j + 1) * XYZZ_WORDS);
      for (int q = 0; q < 6; q++) t = xyzz_dbl(t);                  // 64 T_j (nw <= 2: j = 1)
      xyzz_add(r, a); xyzz_add(r, t);
    }
    xyzz_store(winsum + ((size_t)w * gridDim.y + blockIdx.y) * XYZZ_WORDS, r);
  }
}
// window w = the sum of its S slice partials: one wavefront per window
__global__ void __launch_bounds__(64) k_msm_small_join(const uint32_t *__restrict__ partials, uint32_t S, uint32_t *__restrict__ winsum) {
  const uint32_t w = blockIdx.x, lane = threadIdx.x;
  xyzz acc = xyzz_inf();
  for (uint32_t s_ = lane; s_ < S; s_ += 64) {
    const xyzz p = xyzz_load(partials + ((size_t)w * S + s_) * XYZZ_WORDS);
    if (s_ < 64) acc = p; else xyzz_add(acc, p);
  }
  uint32_t top = 1; while (top < S && top < 64) top <<= 1;
  for (uint32_t dd = top >> 1; dd >= 1; dd >>= 1) {
    const xyzz o = xyzz_shfl_down(acc, (int)dd);
    if (lane + dd < 64) xyzz_add(acc, o);
  }
  if (lane == 0) xyzz_store(winsum + (size_t)w * XYZZ_WORDS, acc);
}

// ------------------------------------------------------------------------------------------------
// host orchestration
static int choose_window(size_t n, size_t batch, const MsmTune &tune) {
  // Cost model in units of one mixed addition (~68 ps chip-wide, measured at 2^20; profiles/):
  //   accumulate: one addition per (scalar, window that holds real bits): ceil(255/c) windows (+ half a window when
  //               c divides 255: the top digit then wraps for half the scalars and a carry window appears);
  //   reduce:     ~10 per bucket (the wave-prefix bucket reduction is latency-bound; fitted at c = 13..16); ~3.5 per
  //               bucket for a large batch of small MSMs, which goes through k_reduce_groups (throughput-bound);
  //   heavy top:  when the top window has few real bits its buckets hold n / 2^r entries each and go through the
  //               wave-cooperative merge tree: a flat ~1.5e6 (0.1 ms) once they span many lanes.
  // One large MSM: the reduction is latency-bound and nearly flat in the bucket count (0.12 ms at c = 11 .. 0.26 ms at c = 16),
  // so the model above overprices wide windows; thresholds read off the (n, c, L) table of benchmarks/sweep_window.py
  // (profiles/r02_window_sweep.txt): c = 16 has 17 windows against 20 at c = 13 and no heavy top window.
  // (re-read after the balanced windows of make_plan, which removed the heavy top window of every width that does not divide 256:
  // 9000-12000 terms 0.47 ms at c = 8 against 0.33 at c = 11; 22 016 terms 0.376 at c = 12; 43 782 terms 0.423 at c = 13)
  if (batch == 1 && n >= 4096) return n < 20000 ? 11 : n < 30000 ? 12 : n < 200000 ? 13 : 16;
  const double gcost = tune.gcost > 0 ? tune.gcost : 3.5;               // tuning sweeps override both
  const int cmin = tune.cmin ? std::max(2, tune.cmin) : 4;
  double best = 1e300; int bc = 8;
  for (int c = cmin; c <= 16; c++) {
    int W = 256 / c + 1, full = 254 / c, r = 255 - c * full;       // r = real bits in the top window (1..c)
    double weff = full + 1 + (r == c ? 0.5 : 0.0);
    const bool groups = c <= 9 && (double)batch * W >= 4096.0;
    double cost = weff * (double)n + (groups ? gcost : 10.0) * W * (double)(1u << (c - 1));
    double top_bucket = r == c ? n / 2.0 : (double)n / (double)(1u << (r < 20 ? r : 20));
    if ((r == c || r < c - 1) && top_bucket > 1024.0) cost += 1.5e6;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

struct MsmPlan {
  size_t n, batch; int c, W, acnt, M, CH, hist_threads, Lw, WPW;   // acnt: windows [0, acnt) are c bits wide, the others c - 1
  bool flat;                   // precomputed table 2^(c w) P_i: all windows of an instance share ONE bucket set
  int Wc;                      // windows left for the window combine (1 when flat)
  uint64_t NB, NS, FB, total_max;   // digit rows (batch * W), bucket sets (batch * W, or batch when flat), buckets, sorted entries
  int L; uint64_t G;           // sorted entries per lane, number of lanes
  int RG;                      // k_reduce_groups lanes per window (0 = wave-per-window path)
  bool marg; MargGeom mg;      // marginal-sum reduction (M >= 256, not the grouped path)
  int ntiles;
};

static MsmPlan make_plan(size_t n, size_t batch, int c, bool flat, const MsmTune &tune) {
  MsmPlan p;
  p.n = n; p.batch = batch; p.c = c; p.W = 256 / c + 1; p.M = 1 << (c - 1);
  p.acnt = p.W;
  // Balanced windows (one MSM over arbitrary points, c not a divisor of 256): with uniform widths the top window holds only
  // 256 - c floor(256 / c) real bits, i.e. a few buckets with n / 2^r entries each — the "heavy" merges (0.05 ms at 2^16 terms, c = 13:
  // 256 buckets of 256 entries).  Instead ceil(256 / c) windows of widths c (the low ones) and c - 1 share the 256 bits, so every
  // window's buckets are evenly filled; the extra top window only catches the carry of the rare scalars within 2^-(c-2) of 2^255.
  if (batch == 1 && !flat && c >= 3 && 256 % c != 0 && !tune.no_balance) {
    const int Wr = (256 + c - 1) / c;
    p.acnt = 256 - Wr * (c - 1);
    p.W = Wr + 1;
  }
  p.flat = flat; p.Wc = flat ? 1 : p.W;
  p.NB = (uint64_t)batch * p.W; p.NS = flat ? batch : p.NB; p.FB = p.NS * p.M; p.total_max = p.NB * n;
  p.hist_threads = p.M >= 8192 ? 1024 : 256;
  size_t per_block = (size_t)p.hist_threads * 64;
  p.CH = (int)std::max<size_t>(1, std::min<size_t>(64, (n + per_block - 1) / per_block));
  // the histogram of a wide window (>= 80 KB of LDS) leaves room for ONE k_hist / k_scatter workgroup per CU: a grid a few workgroups
  // over a whole number of rounds (17 windows x 16 chunks on 256 CUs) runs a nearly empty extra round — take the chunk count that fills
  // the rounds instead (15: sort stage 0.301 -> 0.277 ms at 2^20)
  if ((size_t)p.M * 4 > 80 * 1024 && tune.num_cus > 0) {
    const double rounds = (double)p.NB * p.CH / tune.num_cus;
    if (rounds >= 0.75) {
      const uint64_t r = (uint64_t)(rounds + 0.5) ? (uint64_t)(rounds + 0.5) : 1;
      const uint64_t ch = r * (uint64_t)tune.num_cus / p.NB;
      if (ch >= 1 && ch <= 64 && 4 * ch >= 3 * (uint64_t)p.CH) p.CH = (int)ch;
    }
  }
  if (tune.hist_ch >= 1 && tune.hist_ch <= 64) p.CH = tune.hist_ch;
  // bucket reduce geometry: lanes per window T = M / Lw, at most 64 wavefronts per window
  if (p.M <= 64) { p.Lw = 1; p.WPW = 1; }
  else {
    p.Lw = 1;
    while (p.M / p.Lw > 64 * 64) p.Lw <<= 1;     // cap at 4096 lanes per window
    if (p.M / p.Lw >= 1024 && p.Lw < 4) p.Lw = std::min(4, p.M / 1024);  // amortise the wave scan
    if (p.Lw < 1) p.Lw = 1;
    if (tune.lw) { const int v = tune.lw; if (v >= 1 && (v & (v - 1)) == 0 && p.M / v >= 64 && p.M / v <= 4096) p.Lw = v; }
    p.WPW = p.M / p.Lw / 64;
    if (p.WPW < 1) p.WPW = 1;
  }
  p.RG = 0;
  if (p.M <= 256 && p.NS >= 4096) {                            // M is a power of two >= 2
    // lanes per window: the serial part costs 2 additions per bucket, the in-group scan and tree ~3 log2(RG) more per lane, so
    // keep >= 16 buckets per lane (RG = 8 on 8 buckets is 11 additions per bucket, RG = 1 is 2) unless the launch would
    // then be under ~2 wavefronts per SIMD
    const int cap = std::min(8, p.M);
    p.RG = std::max(1, std::min(cap, p.M / 16));
    while (p.RG < cap && p.NS * (uint64_t)p.RG < 131072) p.RG <<= 1;
    if (tune.rg) { const int v = tune.rg; if (v >= 1 && v <= cap && (v & (v - 1)) == 0) p.RG = v; }
  }
  p.marg = !p.RG && p.M >= 256 && p.NS <= 65535 && !tune.reduce_old;
  memset(&p.mg, 0, sizeof p.mg);
  if (p.marg) {
    MargGeom &g = p.mg;
    g.a = (c - 1 + 1) / 2; g.LO = 1 << g.a; g.HI = p.M / g.LO;          // LO >= HI, both <= 256
    // serial length S per lane: about one wavefront per SIMD over the whole launch (2 * FB bucket reads over ~64K lanes)
    int S = 1;
    while (S < 16 && (2.0 * (double)p.FB) / S > 98304.0) S <<= 1;
    if (tune.marg_s) { const int v = tune.marg_s; if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) S = v; }
    g.SR = std::min(S, g.LO); g.SC = std::min(S, g.HI);
    g.PR = g.LO / g.SR; g.PC = g.HI / g.SC;
    while (g.PR > 64) { g.PR >>= 1; g.SR <<= 1; }                        // the segmented tree lives inside one wavefront
    while (g.PC > 64) { g.PC >>= 1; g.SC <<= 1; }
    g.row_tiles = (uint32_t)(((size_t)g.HI * g.PR + 63) / 64); g.col_tiles = (uint32_t)(((size_t)g.LO * g.PC + 63) / 64);
  }
  // slice length: short enough to keep >= ~150K lanes (two to four wavefronts per SIMD), but at least a quarter of a bucket's
  // expected entries, so that a bucket straddles few lanes and stays on the serial merge path (sweep_window.py table)
  if (batch == 1 && !flat) {
    const uint64_t occ = n / (uint64_t)p.M;
    int lo = 8; while (lo < 64 && (uint64_t)(4 * lo) < occ) lo <<= 1;
    p.L = 64; while (p.L > lo && p.total_max / (uint64_t)p.L < 150000) p.L >>= 1;
  } else {
    p.L = 4;
    while (p.L < 64 && p.total_max / (uint64_t)(2 * p.L) >= 65536) p.L <<= 1;
  }
  // one bucket set for all windows: buckets hold W times more entries; a longer slice keeps a bucket within a few lanes (the
  // serial merge path) while the launch still has two wavefronts per SIMD
  // (L = 256 leaves one wavefront per SIMD and the gathers are no longer hidden: 1.39 ms against 1.15 ms at 2^20)
  if (flat) while (p.L < 128 && (double)p.total_max / (double)p.FB > 4.0 * p.L && p.total_max / (uint64_t)(2 * p.L) >= 65536) p.L <<= 1;
  if (tune.lacc >= 1 && tune.lacc <= 4096) p.L = tune.lacc;
  p.G = (p.total_max + p.L - 1) / p.L; if (!p.G) p.G = 1;
  p.ntiles = (int)((p.FB + SCAN_TILE - 1) / SCAN_TILE);
  return p;
}

// sets K = sum_{w<W} 2^(c-1) * 2^(w*c) as 9 x 32-bit limbs
static RecodeK make_recode_k(int c, int W, int acnt = -1) {
  RecodeK K; memset(&K, 0, sizeof K);
  if (acnt < 0) acnt = W;
  int off = 0;
  for (int w = 0; w < W; w++) {
    const int cw = w < acnt ? c : c - 1, bit = off + cw - 1;
    if (bit < 288) K.k[bit >> 5] |= 1u << (bit & 31);
    off += cw;
  }
  return K;
}

int msm_run_ex(bppp_ctx *ctx, const void *d_scalars, const void *d_points, size_t n, size_t batch, int shared_points, int window_bits, uint64_t *out_xy,
               size_t table_stride, uint32_t *d_out_dev = nullptr);
int msm_run(bppp_ctx *ctx, const void *d_scalars, const void *d_points, size_t n, size_t batch, int shared_points,
            int window_bits, uint64_t *out_xy) {
  return msm_run_ex(ctx, d_scalars, d_points, n, batch, shared_points, window_bits, out_xy, 0);
}
// the same batch of MSMs with the results LEFT IN HBM (d_out [batch][16], canonical affine) and nothing waited for: a link in a stream of kernels
// (the lockstep argument's late rounds over per-proof bases, csrc/nlb.hip).  batch > 4.
int msm_batch_dev(bppp_ctx *ctx, const void *d_scalars, const void *d_points, size_t n, size_t batch, int shared_points, int window_bits, uint32_t *d_out) {
  if (!d_out || batch <= 4 || !n) return fail(ctx, BPPP_ERR_ARG, "msm_batch_dev: bad arguments");
  uint64_t dummy[8];
  return msm_run_ex(ctx, d_scalars, d_points, n, batch, shared_points, window_bits, dummy, 0, d_out);
}
// table_stride != 0: d_points is a precomputed table [W][table_stride] of 2^(c w) P_i for the given window_bits = c (bppp_basis):
// all windows of an instance share one bucket set, so there is one bucket reduction per instance and no window combine
int msm_run_ex(bppp_ctx *ctx, const void *d_scalars, const void *d_points, size_t n, size_t batch, int shared_points, int window_bits, uint64_t *out_xy,
               size_t table_stride, uint32_t *d_out_dev) {
  using namespace bppp_host;
  if (!out_xy) return fail(ctx, BPPP_ERR_ARG, "msm: null output");
  if (n == 0 || batch == 0) { memset(out_xy, 0, 64 * (batch ? batch : 1)); return BPPP_OK; }
  if (!d_scalars || !d_points) return fail(ctx, BPPP_ERR_ARG, "msm: null input");
  if (n >= (1ull << 31)) return fail(ctx, BPPP_ERR_ARG, "msm: n must be < 2^31");
  // ---- one small or mid-size MSM: k_msm_small over (window, slice of <= 4096 terms) workgroups, the slices of a window joined by one
  // wavefront each, then the host's Horner combine over the W window sums
  const size_t small_max = ctx->tune.small_max ? (size_t)ctx->tune.small_max : MSM_SMALL_DEFAULT_MAX;
  if (batch == 1 && !table_stride && !window_bits && n <= small_max && !ctx->tune.no_small && !d_out_dev) {
    const int c = ctx->tune.small_c >= 5 && ctx->tune.small_c <= 8 ? ctx->tune.small_c : 6, M = 1 << (c - 1);
    const bool bal = 256 % c != 0 && !ctx->tune.no_balance;
    const int Wr = (256 + c - 1) / c, acnt = bal ? 256 - Wr * (c - 1) : 256 / c + 1, W = bal ? Wr + 1 : 256 / c + 1;
    size_t len = ctx->tune.small_len >= 64 && ctx->tune.small_len <= (int)MSM_SMALL_MAX ? (size_t)ctx->tune.small_len : (n <= 2048 ? 512 : 1024);
    if (n <= len + len / 2) len = n;                                  // a second slice has to pay for the join launch
    const size_t S = (n + len - 1) / len;
    Carver cv0(nullptr, 0);
    cv0.take<uint32_t>((size_t)W * S * XYZZ_WORDS); cv0.take<uint32_t>((size_t)W * XYZZ_WORDS);
    int rc = ensure_workspace(ctx, cv0.off); if (rc) return rc;
    Carver cv(ctx->ws, ctx->ws_bytes);
    uint32_t *partials = cv.take<uint32_t>((size_t)W * S * XYZZ_WORDS), *winsum = cv.take<uint32_t>((size_t)W * XYZZ_WORDS);
    const size_t bytes = (size_t)W * XYZZ_WORDS * 4;
    rc = ensure_pinned(ctx, bytes); if (rc) return rc;
    hipStream_t st = ctx->stream;
    for (int i = 0; i <= 2; i++) prof_mark(ctx, i);
    if (ctx->pre_acc) { auto f = ctx->pre_acc; ctx->pre_acc = nullptr; int rc_ = f(ctx->pre_acc_arg); if (rc_) return rc_; }
    const size_t lds = (((size_t)(2 * M + 1) + len + (len + 1) / 2 + 3) & ~(size_t)3) * 4 + (size_t)(M + 4) * XYZZ_WORDS * 4;
    k_msm_small<<<dim3((unsigned)W, (unsigned)S), dim3(256), lds, st>>>((const uint32_t *)d_scalars, (const uint32_t *)d_points, (uint32_t)n, (uint32_t)len, c,
                                                                       acnt, make_recode_k(c, W, acnt), S > 1 ? partials : winsum);
    if (S > 1) k_msm_small_join<<<dim3((unsigned)W), dim3(64), 0, st>>>(partials, (uint32_t)S, winsum);
    for (int i = 3; i <= 5; i++) prof_mark(ctx, i);
    BPPP_HIP(ctx, hipMemcpyAsync(ctx->pinned, winsum, bytes, hipMemcpyDeviceToHost, st));
    prof_mark(ctx, 6);
    BPPP_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t *ws = (const uint32_t *)ctx->pinned;
    HJac r = hj_inf();
    for (int w = W - 1; w >= 0; w--) {
      const uint32_t *q = ws + (size_t)w * XYZZ_WORDS;
      for (int k = 0; k < (w < acnt ? c : c - 1); k++) r = hj_dbl(r);
      r = hj_add(r, hj_from_xyzz(from_limbs26(q), from_limbs26(q + 10), from_limbs26(q + 20), from_limbs26(q + 30)));
    }
    HAff a = hj_to_aff(r);
    a.x.store(out_xy); a.y.store(out_xy + 4);
    BPPP_HIP(ctx, hipGetLastError());
    prof_collect(ctx, 7);
    return BPPP_OK;
  }
  int c = window_bits ? window_bits : choose_window(n, batch, ctx->tune);
  if (!window_bits && ctx->tune.window_batched) { const int v = ctx->tune.window_batched; if (batch > 4 && v >= 2 && v <= 16) c = v; }   // tuning sweeps
  if (c < 2 || c > 16) return fail(ctx, BPPP_ERR_ARG, "msm: window_bits must be in [2,16]");
  const bool flat = table_stride != 0;
  if (flat && (!window_bits || shared_points != 1 || n > table_stride || (uint64_t)(256 / c + 1) * table_stride >= (1ull << 31)))
    return fail(ctx, BPPP_ERR_ARG, "msm: bad precomputed-table arguments");
  MsmPlan p = make_plan(n, batch, c, flat, ctx->tune);
  if (p.FB >= (1ull << 32) - 1 || p.total_max >= (1ull << 32) - 1)
    return fail(ctx, BPPP_ERR_ARG, "msm: batch*windows*buckets or batch*n*windows exceeds 2^32; split the batch");

  // ---- carve the workspace
  size_t need = 0;
  for (int pass = 0; pass < 2; pass++) {
    Carver cv(pass ? ctx->ws : nullptr, ctx->ws_bytes);
    const uint32_t stride = (uint32_t)((n + 7) & ~(size_t)7);     // digit rows are 16-byte aligned
    uint16_t *dig = cv.take<uint16_t>((size_t)p.NB * stride + 8);
    unsigned long long *negmask = cv.take<unsigned long long>((batch * n + 63) / 64 + 1);
    uint32_t *blockhist = cv.take<uint32_t>((size_t)p.NB * p.CH * p.M);
    uint32_t *count = cv.take<uint32_t>(p.FB + 1);
    uint32_t *start = cv.take<uint32_t>(p.FB + 1);
    uint32_t *tiles = cv.take<uint32_t>(p.ntiles + 1);
    uint32_t *sorted = cv.take<uint32_t>(p.total_max + 4);
    uint32_t *buckets = cv.take<uint32_t>((size_t)p.FB * XYZZ_WORDS);
    uint32_t *rec_pt = cv.take<uint32_t>((size_t)p.G * 2 * XYZZ_WORDS);
    // heavy buckets span > 9 lanes: at most G/9 of them, and at most G/256 + G/9 (bucket, chunk) items
    size_t hmax = (size_t)(p.G / 8 + 2);
    uint2 *heavy_items = cv.take<uint2>(hmax);
    uint4 *heavy_buckets = cv.take<uint4>(hmax);
    uint32_t *chunk_sums = cv.take<uint32_t>(hmax * XYZZ_WORDS);
    uint32_t *heavy_count = cv.take<uint32_t>(4);
    uint32_t *red = cv.take<uint32_t>(p.marg ? (size_t)p.NS * (p.mg.HI + p.mg.LO) * XYZZ_WORDS : (size_t)p.NS * p.WPW * 2 * XYZZ_WORDS);
    uint32_t *winsum = cv.take<uint32_t>((size_t)p.NS * 2 * XYZZ_WORDS);
    uint32_t *out_aff = cv.take<uint32_t>((size_t)batch * 16);
    if (!pass) { need = cv.off; int rc = ensure_workspace(ctx, need); if (rc) return rc; continue; }

    hipStream_t st = ctx->stream;
    const size_t lds = (size_t)p.M * 4;
    if (lds > 64 * 1024) {
      BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      BPPP_HIP(ctx, hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    prof_mark(ctx, 0);
    // 1. digits
    uint64_t total_sc = (uint64_t)batch * n;
    k_digits<<<dim3((unsigned)((total_sc + 255) / 256)), dim3(256), 0, st>>>((const uint32_t *)d_scalars, total_sc, (uint32_t)n, stride, c, p.W, p.acnt,
                                                                            make_recode_k(c, p.W, p.acnt), dig, negmask);
    prof_mark(ctx, 1);
    // 2. sort
    k_hist<<<dim3((unsigned)p.NB, p.CH), dim3(p.hist_threads), lds, st>>>(dig, (uint32_t)n, stride, c, p.CH, blockhist);
    // flat: the (window, chunk) histograms of an instance are W * CH chunks of ONE bucket set (same memory layout)
    k_chunk_prefix<<<dim3((unsigned)((p.FB + 255) / 256)), dim3(256), 0, st>>>(blockhist, p.M, p.flat ? p.W * p.CH : p.CH, p.FB, count);
    k_scan_tile_sums<<<dim3(p.ntiles), dim3(256), 0, st>>>(count, p.FB, tiles);
    k_scan_top<<<dim3(1), dim3(1024), 0, st>>>(tiles, (uint32_t)p.ntiles, start + p.FB);
    k_scan_apply<<<dim3(p.ntiles), dim3(256), 0, st>>>(count, p.FB, tiles, start);
    k_scatter<<<dim3((unsigned)p.NB, p.CH), dim3(p.hist_threads), lds, st>>>(dig, negmask, (uint32_t)n, stride, c, p.CH, p.W, blockhist, start, sorted,
                                                                             p.flat ? (uint32_t)table_stride : 0u);
    BPPP_HIP(ctx, hipMemsetAsync(buckets, 0, (size_t)p.FB * XYZZ_WORDS * 4, st));
    BPPP_HIP(ctx, hipMemsetAsync(heavy_count, 0, 16, st));
    prof_mark(ctx, 2);
    if (ctx->pre_acc) { auto f = ctx->pre_acc; ctx->pre_acc = nullptr; int rc_ = f(ctx->pre_acc_arg); if (rc_) return rc_; }
    // 3. accumulate
    k_acc_points<<<dim3((unsigned)((p.G + 255) / 256)), dim3(256), 0, st>>>(sorted, start, (uint32_t)p.FB, (const uint32_t *)d_points, (uint32_t)n,
                                                                           (uint32_t)(p.Wc * p.M), shared_points, p.L, p.G, buckets, rec_pt);
    prof_mark(ctx, 3);
    k_merge<<<dim3((unsigned)((p.FB + 255) / 256)), dim3(256), 0, st>>>(start, count, p.FB, p.L, rec_pt, buckets, heavy_items, heavy_buckets, heavy_count);
    k_merge_heavy<<<dim3(2048), dim3(64), 0, st>>>(start, count, p.L, rec_pt, buckets, heavy_items, heavy_count, chunk_sums);
    k_merge_heavy2<<<dim3(256), dim3(64), 0, st>>>(heavy_buckets, heavy_count, chunk_sums, buckets);
    prof_mark(ctx, 4);
    // 4. bucket reduce
    if (p.RG) {
      k_reduce_groups<<<dim3((unsigned)((p.NS * p.RG + 63) / 64)), dim3(64), 0, st>>>(buckets, p.M, p.RG, (uint32_t)p.NS, winsum);
    } else if (p.marg) {
      uint32_t *Rm = red, *Cm = red + (size_t)p.NS * p.mg.HI * XYZZ_WORDS;
      k_reduce_marg<<<dim3(p.mg.row_tiles + p.mg.col_tiles, (unsigned)p.NS), dim3(64), 0, st>>>(buckets, p.M, p.mg, Rm, Cm);
      k_reduce_tail<<<dim3((unsigned)p.NS, 2), dim3(64u * (unsigned)((p.mg.LO + 63) / 64)), 0, st>>>(Rm, Cm, p.mg, winsum);
    } else {
      k_reduce1<<<dim3((unsigned)p.NS, p.WPW), dim3(64), 0, st>>>(buckets, p.M, p.Lw, p.WPW, red);
      k_reduce2<<<dim3((unsigned)p.NS), dim3(64), 0, st>>>(red, p.Lw, p.WPW, winsum);
    }
    prof_mark(ctx, 5);
    // 5. window combine
    if (d_out_dev) {
      k_window_combine<<<dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st>>>(winsum, p.Wc, c, p.marg ? p.mg.a : 0, (uint32_t)batch, d_out_dev);
      BPPP_HIP(ctx, hipGetLastError());
      return BPPP_OK;
    }
    if (batch > 4) {
      k_window_combine<<<dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st>>>(winsum, p.Wc, c, p.marg ? p.mg.a : 0, (uint32_t)batch, out_aff);
      BPPP_HIP(ctx, hipMemcpyAsync(out_xy, out_aff, batch * 64, hipMemcpyDeviceToHost, st));
      prof_mark(ctx, 6);
      BPPP_HIP(ctx, hipStreamSynchronize(st));
    } else {
      size_t bytes = (size_t)p.NS * (p.marg ? 2 : 1) * XYZZ_WORDS * 4;
      int rc = ensure_pinned(ctx, bytes); if (rc) return rc;
      BPPP_HIP(ctx, hipMemcpyAsync(ctx->pinned, winsum, bytes, hipMemcpyDeviceToHost, st));
      prof_mark(ctx, 6);
      BPPP_HIP(ctx, hipStreamSynchronize(st));
      const uint32_t *ws = (const uint32_t *)ctx->pinned;
      for (size_t b = 0; b < batch; b++) {
        HJac r = hj_inf();
        auto pt = [&](const uint32_t *q) { return hj_from_xyzz(from_limbs26(q), from_limbs26(q + 10), from_limbs26(q + 20), from_limbs26(q + 30)); };
        for (int w = p.Wc - 1; w >= 0; w--) {
          const int cw = (p.flat || w < p.acnt) ? c : c - 1;     // this window's width = the doublings that separate it from the one above
          if (p.marg) {                               // window value = 2^a * W1 + W2: the doublings are split around W1
            const uint32_t *q = ws + ((size_t)b * p.Wc + w) * 2 * XYZZ_WORDS;
            for (int k = 0; k < cw - p.mg.a; k++) r = hj_dbl(r);
            r = hj_add(r, pt(q));
            for (int k = 0; k < p.mg.a; k++) r = hj_dbl(r);
            r = hj_add(r, pt(q + XYZZ_WORDS));
          } else {
            for (int k = 0; k < cw; k++) r = hj_dbl(r);
            r = hj_add(r, pt(ws + ((size_t)b * p.Wc + w) * XYZZ_WORDS));
          }
        }
        HAff a = hj_to_aff(r);
        a.x.store(out_xy + 8 * b); a.y.store(out_xy + 8 * b + 4);
      }
    }
    BPPP_HIP(ctx, hipGetLastError());
    prof_collect(ctx, 7);
  }
  return BPPP_OK;
}

}  // namespace bppp
