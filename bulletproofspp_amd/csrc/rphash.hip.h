// rphash.hip.h — device helpers of the Fiat-Shamir transcript: `show` of a coordinate (decimal text) and the SHA-256 of
// header <> text with the digest read as a field element (shaOracle, app/Main.hs:64-80).  Shared by the batch verifier
// (csrc/rp.hip) and the batch prover (csrc/rpprove_dev.hip).
#pragma once
#include "fe.hip.h"
#include "sha256.hip.h"

namespace bppp {

// `show` of a field element = its decimal integer.  v is split into nine 9-digit chunks (10^81 > 2^256) by repeated division.
struct Dec { uint32_t ch[9]; uint32_t top, len; };
BPPP_DI uint32_t ndigits9(uint32_t v) {
  return v >= 100000000u ? 9 : v >= 10000000u ? 8 : v >= 1000000u ? 7 : v >= 100000u ? 6 : v >= 10000u ? 5 : v >= 1000u ? 4 : v >= 100u ? 3 : v >= 10u ? 2 : 1;
}
BPPP_DI Dec dec_convert(fe v) {
  Dec d;
#pragma unroll
  for (int c = 0; c < 9; c++) {
    uint64_t rem = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
      const uint64_t cur = (rem << 32) | v.v[i];
      const uint64_t q = cur / 1000000000ull;
      v.v[i] = (uint32_t)q; rem = cur - q * 1000000000ull;
    }
    d.ch[c] = (uint32_t)rem;
  }
  d.top = 0;
#pragma unroll
  for (int c = 1; c < 9; c++) if (d.ch[c]) d.top = c;
  uint32_t tv = 0;
#pragma unroll
  for (int c = 0; c < 9; c++) if ((uint32_t)c == d.top) tv = d.ch[c];
  d.len = 9 * d.top + ndigits9(tv);
  return d;
}
// writes the d.len characters so that they END at `end` (exclusive); returns the start
BPPP_DI uint8_t *dec_write_backward(const Dec &d, uint8_t *end) {
  uint8_t *p = end;
#pragma unroll
  for (int c = 0; c < 9; c++) {
    if ((uint32_t)c > d.top) continue;
    uint32_t v = d.ch[c];
    const uint32_t n = (uint32_t)c == d.top ? ndigits9(v) : 9u;
    for (uint32_t j = 0; j < n; j++) { *--p = (uint8_t)('0' + v % 10u); v /= 10u; }
  }
  return p;
}

BPPP_DI uint32_t load_unaligned_be32(const uint8_t *p) {
  const uintptr_t a = (uintptr_t)p;
  const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3) * 8;
  const uint64_t two = ((uint64_t)q[1] << 32) | q[0];
  return __builtin_bswap32((uint32_t)(two >> sh));
}


// bytes p .. p+3 (big-endian) of the message  hdr <> tx[0 .. mlen - hlen) <> 0x80 <> 0 ...  without a branch: the header comes as 16
// big-endian words, the text through the unaligned read; words that straddle a boundary are masked together.  (A bytewise
// assembly of the first and last blocks cost ~130 dependent byte loads behind divergent branches per hash.)
BPPP_DI uint32_t rp_msg_word(const uint32_t *hdr_be, uint32_t hlen, const uint8_t *tx, uint32_t mlen, uint32_t p) {
  int32_t off = (int32_t)p - (int32_t)hlen;
  off = off < -4 ? -4 : off;                                   // bytes before the text are masked out below; stay next to the buffer
  const uint32_t T = load_unaligned_be32(tx + off);
  const uint32_t H = hdr_be[(p >> 2) & 15u];
  const int32_t kh = (int32_t)hlen - (int32_t)p;               // header bytes in this word: >= 4 all of it, <= 0 none
  const uint32_t hm = kh >= 4 ? ~0u : kh <= 0 ? 0u : (~0u << (8 * (4 - kh)));
  uint32_t w = (H & hm) | (T & ~hm);
  const int32_t k = (int32_t)mlen - (int32_t)p;                // message bytes in this word
  const uint32_t mm = k >= 4 ? ~0u : k <= 0 ? 0u : (~0u << (8 * (4 - k)));
  const uint32_t pad = (k >= 0 && k < 4) ? (0x80u << (8 * (3 - k))) : 0u;
  return (w & mm) | pad;
}

// SHA-256 (hdr <> tx[0 .. tlen)), digest -> Fr by Binary (Prime p) (src/Encoding.hs:75-79), toP.  hdr_be: the header (<= 64 bytes) as 16
// big-endian words, zero-padded, in global memory; tx: any alignment, readable from 4 bytes before it to 8 bytes past its end
// (the callers' buffers have that slack).
BPPP_DI fe rp_hash_to_fr(const uint32_t *hdr_be, uint32_t hlen, const uint8_t *tx, uint32_t tlen) {
  uint32_t st[8];
  sha256_init(st);
  uint32_t w[16];
  const uint32_t mlen = hlen + tlen;
  const uint32_t nblk = (mlen + 9 + 63) / 64;
  // Interior blocks (all 64 bytes inside the text) are read as 17 aligned dwords and shifted into place.  One wavefront per
  // SIMD is all this kernel gets at batch sizes of a few thousand (15 hashes per proof), so nothing else hides the load latency:
  // the dwords of block k + 1 are requested BEFORE block k is compressed.
  const uint32_t first_fast = (hlen + 63) / 64, last_fast = mlen / 64;            // fast blocks: [first_fast, last_fast)
  const uintptr_t a0 = (uintptr_t)(tx + ((size_t)first_fast * 64 - hlen));
  const uint32_t sh = (uint32_t)(a0 & 3) * 8;
  const uint32_t *q = (const uint32_t *)(a0 & ~(uintptr_t)3);                       // dword holding the first byte of block first_fast
  uint32_t nx[17];
  if (first_fast < last_fast) {
#pragma unroll
    for (int i = 0; i < 17; i++) nx[i] = q[i];
  }
  for (uint32_t blk = 0; blk < nblk; blk++) {
    const uint32_t p0 = blk * 64;
    if (blk >= first_fast && blk < last_fast) {
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32((uint32_t)((((uint64_t)nx[i + 1] << 32) | nx[i]) >> sh));
      if (blk + 1 < last_fast) {
        const uint32_t *qn = q + (size_t)(blk + 1 - first_fast) * 16;
#pragma unroll
        for (int i = 0; i < 17; i++) nx[i] = qn[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = rp_msg_word(hdr_be, hlen, tx, mlen, p0 + 4 * i);
      if (blk == nblk - 1) { w[14] = 0; w[15] = mlen * 8; }     // mlen < 2^29 bytes
    }
    sha256_compress(st, w);
  }
  fe v; sha256_digest_to_limbs(st, v.v);
  fe t; const uint32_t br = raw_sub(t, v, fr_modulus());
#pragma unroll
  for (int i = 0; i < 8; i++) v.v[i] = br ? v.v[i] : t.v[i];
  return v;
}

// The same hash by a PAIR of wavefronts (workgroup of 128 threads, 64 hashes): threads 0..63 PRODUCE — they fetch the message words of
// block k + 1, expand its schedule and leave W[t] + K[t] in LDS — while threads 64..127 CONSUME block k (the 64 dependent rounds).  A lane
// of one wavefront issues ~1700 dependent-ish instructions per block at one wavefront's rate; split this way the critical path is the
// ~900 instructions of the rounds, and the two wavefronts sit on different SIMDs.  `lds`: 2 x 64 x 64 + 1 words.  Every thread of the
// workgroup must call this (inactive lanes pass active = false); the result is valid in the consumer threads only.
static constexpr uint32_t RP_HASH_PC_LDS_WORDS = 2 * 64 * 64 + 1;
BPPP_DI fe rp_hash_to_fr_pc(bool active, const uint32_t *hdr_be, uint32_t hlen, const uint8_t *tx, uint32_t tlen, uint32_t *lds) {
  const uint32_t lane = threadIdx.x & 63u;
  const bool producer = threadIdx.x < 64u;
  const uint32_t mlen = hlen + tlen;
  const uint32_t nblk = active ? (mlen + 9 + 63) / 64 : 0u;
  uint32_t *maxp = lds + 2 * 64 * 64;
  if (threadIdx.x == 0) *maxp = 0;
  __syncthreads();
  if (producer) atomicMax(maxp, nblk);
  __syncthreads();
  const uint32_t maxblk = *maxp;
  uint32_t st[8];
  sha256_init(st);
  // producer state: as in rp_hash_to_fr, interior blocks come as 17 aligned dwords requested one block ahead
  const uint32_t first_fast = (hlen + 63) / 64, last_fast = mlen / 64;
  const uintptr_t a0 = (uintptr_t)(tx + ((size_t)first_fast * 64 - hlen));
  const uint32_t sh = (uint32_t)(a0 & 3) * 8;
  const uint32_t *q = (const uint32_t *)(a0 & ~(uintptr_t)3);
  uint32_t nx[17];
  if (producer && nblk && first_fast < last_fast) {
#pragma unroll
    for (int i = 0; i < 17; i++) nx[i] = q[i];
  }
  for (uint32_t it = 0; it <= maxblk; it++) {
    if (producer) {
      if (it < nblk) {
        const uint32_t blk = it, p0 = blk * 64;
        uint32_t w[16];
        if (blk >= first_fast && blk < last_fast) {
#pragma unroll
          for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32((uint32_t)((((uint64_t)nx[i + 1] << 32) | nx[i]) >> sh));
          if (blk + 1 < last_fast) {
            const uint32_t *qn = q + (size_t)(blk + 1 - first_fast) * 16;
#pragma unroll
            for (int i = 0; i < 17; i++) nx[i] = qn[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; i++) w[i] = rp_msg_word(hdr_be, hlen, tx, mlen, p0 + 4 * i);
          if (blk == nblk - 1) { w[14] = 0; w[15] = mlen * 8; }
        }
        uint32_t *dst = lds + (size_t)(it & 1u) * 64 * 64 + lane;
        sha256_schedule_wk(w, [&](int t, uint32_t v) { dst[t * 64] = v; });
      }
    } else if (it >= 1 && it - 1 < nblk) {
      const uint32_t *src = lds + (size_t)((it - 1) & 1u) * 64 * 64 + lane;
      uint32_t wk[64];                                   // all 64 LDS reads are issued before the first round needs one
#pragma unroll
      for (int t = 0; t < 64; t++) wk[t] = src[t * 64];
      sha256_rounds_wk(st, [&](int t) { return wk[t]; });
    }
    __syncthreads();
  }
  fe v; sha256_digest_to_limbs(st, v.v);
  fe t; const uint32_t br = raw_sub(t, v, fr_modulus());
#pragma unroll
  for (int i = 0; i < 8; i++) v.v[i] = br ? v.v[i] : t.v[i];
  return v;
}

}  // namespace bppp
