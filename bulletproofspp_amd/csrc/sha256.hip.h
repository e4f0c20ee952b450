// sha256.hip.h — SHA-256 (FIPS 180-4) for the Fiat-Shamir transcript, on the device and on the host.
//
// The reference's oracle is `hash = decode . fromStrict . SHA.hash` (app/Main.hs:64-65, cryptohash-sha256) over
// `show n <> show (length ps) <> foldMap (coords . toA) ps` (shaOracle, app/Main.hs:75-80): every oracle call hashes the WHOLE
// transcript again, newest commitments first (src/ZKP.hs:96-101).  A 64by64 verification makes 15 such hashes of ~10-13 KB each
// (~2800 compression-function calls per proof): with the group arithmetic on the GPU that hashing is what is left, so it runs
// there too — one lane per (proof, oracle output), the transcript text staged once per proof in HBM (csrc/rp.hip).
//
// One compression function, `sha256_compress`, compiled for both sides (the host side serves the prover, whose oracle calls
// are sequential).  The digest -> field decode of Binary (Prime p) (src/Encoding.hs:75-79: four big-endian 64-bit words,
// least-significant first) is `sha256_digest_to_limbs`.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BPPP_HD __host__ __device__ __forceinline__
#else
#define BPPP_HD inline
#endif

namespace bppp {

BPPP_HD uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// three-input bit operations: one v_bitop3_b32 each on gfx950 (truth table from src0 = 0xF0, src1 = 0xCC, src2 = 0xAA); the
// compiler keeps two-input xors otherwise, and the XOR chains are a quarter of a compression's instructions
#if defined(__HIP_DEVICE_COMPILE__)
#define SHA_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define SHA_CH(e, f, g) __builtin_amdgcn_bitop3_b32((e), (f), (g), 0xCA)
#define SHA_MAJ(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xE8)
#else
#define SHA_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#define SHA_CH(e, f, g) (((e) & (f)) ^ (~(e) & (g)))
#define SHA_MAJ(a, b, c) (((a) & (b)) ^ ((a) & (c)) ^ ((b) & (c)))
#endif

BPPP_HD void sha256_init(uint32_t h[8]) {
  h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
  h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}

// one 64-byte block, given as 16 big-endian words; the round constants are literals so that neither side needs a table in memory
BPPP_HD void sha256_compress(uint32_t h[8], uint32_t w[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#define SHA_RND(i, K)                                                                                            \
  {                                                                                                              \
    uint32_t wi;                                                                                                 \
    if ((i) < 16) wi = w[(i)];                                                                                   \
    else {                                                                                                       \
      const uint32_t w15 = w[((i) + 1) & 15], w2 = w[((i) + 14) & 15];                                           \
      const uint32_t s0 = SHA_XOR3(sha_rotr(w15, 7), sha_rotr(w15, 18), (w15 >> 3));                             \
      const uint32_t s1 = SHA_XOR3(sha_rotr(w2, 17), sha_rotr(w2, 19), (w2 >> 10));                              \
      wi = w[(i) & 15] = w[(i) & 15] + s0 + w[((i) + 9) & 15] + s1;                                              \
    }                                                                                                            \
    const uint32_t t1 = hh + SHA_XOR3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25)) + SHA_CH(e, f, g) + (K) + wi; \
    const uint32_t t2 = SHA_XOR3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22)) + SHA_MAJ(a, b, c);          \
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;                                          \
  }
  SHA_RND(0, 0x428a2f98u) SHA_RND(1, 0x71374491u) SHA_RND(2, 0xb5c0fbcfu) SHA_RND(3, 0xe9b5dba5u)
  SHA_RND(4, 0x3956c25bu) SHA_RND(5, 0x59f111f1u) SHA_RND(6, 0x923f82a4u) SHA_RND(7, 0xab1c5ed5u)
  SHA_RND(8, 0xd807aa98u) SHA_RND(9, 0x12835b01u) SHA_RND(10, 0x243185beu) SHA_RND(11, 0x550c7dc3u)
  SHA_RND(12, 0x72be5d74u) SHA_RND(13, 0x80deb1feu) SHA_RND(14, 0x9bdc06a7u) SHA_RND(15, 0xc19bf174u)
  SHA_RND(16, 0xe49b69c1u) SHA_RND(17, 0xefbe4786u) SHA_RND(18, 0x0fc19dc6u) SHA_RND(19, 0x240ca1ccu)
  SHA_RND(20, 0x2de92c6fu) SHA_RND(21, 0x4a7484aau) SHA_RND(22, 0x5cb0a9dcu) SHA_RND(23, 0x76f988dau)
  SHA_RND(24, 0x983e5152u) SHA_RND(25, 0xa831c66du) SHA_RND(26, 0xb00327c8u) SHA_RND(27, 0xbf597fc7u)
  SHA_RND(28, 0xc6e00bf3u) SHA_RND(29, 0xd5a79147u) SHA_RND(30, 0x06ca6351u) SHA_RND(31, 0x14292967u)
  SHA_RND(32, 0x27b70a85u) SHA_RND(33, 0x2e1b2138u) SHA_RND(34, 0x4d2c6dfcu) SHA_RND(35, 0x53380d13u)
  SHA_RND(36, 0x650a7354u) SHA_RND(37, 0x766a0abbu) SHA_RND(38, 0x81c2c92eu) SHA_RND(39, 0x92722c85u)
  SHA_RND(40, 0xa2bfe8a1u) SHA_RND(41, 0xa81a664bu) SHA_RND(42, 0xc24b8b70u) SHA_RND(43, 0xc76c51a3u)
  SHA_RND(44, 0xd192e819u) SHA_RND(45, 0xd6990624u) SHA_RND(46, 0xf40e3585u) SHA_RND(47, 0x106aa070u)
  SHA_RND(48, 0x19a4c116u) SHA_RND(49, 0x1e376c08u) SHA_RND(50, 0x2748774cu) SHA_RND(51, 0x34b0bcb5u)
  SHA_RND(52, 0x391c0cb3u) SHA_RND(53, 0x4ed8aa4au) SHA_RND(54, 0x5b9cca4fu) SHA_RND(55, 0x682e6ff3u)
  SHA_RND(56, 0x748f82eeu) SHA_RND(57, 0x78a5636fu) SHA_RND(58, 0x84c87814u) SHA_RND(59, 0x8cc70208u)
  SHA_RND(60, 0x90befffau) SHA_RND(61, 0xa4506cebu) SHA_RND(62, 0xbef9a3f7u) SHA_RND(63, 0xc67178f2u)
#undef SHA_RND
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// The same compression split in two for a PAIR of wavefronts (csrc/rphash.hip.h): the message schedule of a block has no dependency on
// the chaining state, so one wavefront expands W[0..63] (+ K) of block k + 1 while the other runs the 64 rounds of block k.
BPPP_HD uint32_t sha256_k(int i) {
  const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  return K[i];
}
// W[t] + K[t] for t = 0 .. 63, handed to put(t, value); w[16] is clobbered
template <class Put> BPPP_HD void sha256_schedule_wk(uint32_t w[16], Put put) {
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) wi = w[i];
    else {
      const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const uint32_t s0 = SHA_XOR3(sha_rotr(w15, 7), sha_rotr(w15, 18), (w15 >> 3));
      const uint32_t s1 = SHA_XOR3(sha_rotr(w2, 17), sha_rotr(w2, 19), (w2 >> 10));
      wi = w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    put(i, wi + sha256_k(i));
  }
}
// the 64 rounds over get(t) = W[t] + K[t]; h += the result
template <class Get> BPPP_HD void sha256_rounds_wk(uint32_t h[8], Get get) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    const uint32_t t1 = hh + SHA_XOR3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25)) + SHA_CH(e, f, g) + get(i);
    const uint32_t t2 = SHA_XOR3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22)) + SHA_MAJ(a, b, c);
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// `decode` of the 32 digest bytes through Binary (Prime p) (src/Encoding.hs:75-79): limb i (64 bits, i = 0 least significant)
// is the big-endian word at bytes 8i .. 8i+7, i.e. (h[2i] << 32) | h[2i+1].  Output: 8 little-endian 32-bit limbs of the
// 256-bit integer, NOT yet reduced (toP reduces; the caller subtracts the modulus once — the value is < 2^256 < 2 m).
BPPP_HD void sha256_digest_to_limbs(const uint32_t h[8], uint32_t v[8]) {
  for (int i = 0; i < 4; i++) { v[2 * i] = h[2 * i + 1]; v[2 * i + 1] = h[2 * i]; }
}

// ---- host side: the SHA extensions of the host CPU when it has them (every x86-64 server part since Zen / Ice Lake): ~2 GB/s per
// core against ~0.3 GB/s for the portable rounds above — the oracle of small batches runs on the host (csrc/rp.hip, csrc/rpprove_dev.hip)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
}  // namespace bppp
#include <immintrin.h>
namespace bppp {
#define BPPP_SHA_NI 1
__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_shani(uint32_t state[8], const uint8_t *data, size_t nblk) {
  alignas(16) static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
  __m128i t = _mm_loadu_si128((const __m128i *)&state[0]), s1 = _mm_loadu_si128((const __m128i *)&state[4]);
  t = _mm_shuffle_epi32(t, 0xB1); s1 = _mm_shuffle_epi32(s1, 0x1B);
  __m128i s0 = _mm_alignr_epi8(t, s1, 8);                // ABEF
  s1 = _mm_blend_epi16(s1, t, 0xF0);                     // CDGH
  for (; nblk; nblk--, data += 64) {
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int g = 0; g < 16; g++) {                       // four rounds per step
      if (g < 4) m[g] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(data + 16 * g)), bswap);
      __m128i wk = _mm_add_epi32(m[g & 3], _mm_load_si128((const __m128i *)&K[4 * g]));
      s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
      if (g >= 3 && g <= 14) {                           // next schedule words: W[t] = sigma1(W[t-2]) + W[t-7] + sigma0(W[t-15]) + W[t-16]
        const __m128i w7 = _mm_alignr_epi8(m[g & 3], m[(g - 1) & 3], 4);
        m[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(m[(g + 1) & 3], w7), m[g & 3]);
      }
      wk = _mm_shuffle_epi32(wk, 0x0E);
      s0 = _mm_sha256rnds2_epu32(s0, s1, wk);
      if (g >= 1 && g <= 12) m[(g - 1) & 3] = _mm_sha256msg1_epu32(m[(g - 1) & 3], m[g & 3]);
    }
    s0 = _mm_add_epi32(s0, save0); s1 = _mm_add_epi32(s1, save1);
  }
  t = _mm_shuffle_epi32(s0, 0x1B); s1 = _mm_shuffle_epi32(s1, 0xB1);
  s0 = _mm_blend_epi16(t, s1, 0xF0);                     // DCBA
  s1 = _mm_alignr_epi8(s1, t, 8);                        // HGFE
  _mm_storeu_si128((__m128i *)&state[0], s0); _mm_storeu_si128((__m128i *)&state[4], s1);
}
inline bool sha256_have_shani() { static const bool have = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1"); return have; }
#endif

// ---- host-side streaming interface (prover transcripts, rho derivation on the host when needed)
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len;
  Sha256() { reset(); }
  void reset() { sha256_init(h); len = 0; }
  void block(const uint8_t *p) {
#if defined(BPPP_SHA_NI)
    if (sha256_have_shani()) { sha256_blocks_shani(h, p, 1); return; }
#endif
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    sha256_compress(h, w);
  }
  void update(const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    size_t fill = (size_t)(len & 63);
    len += n;
    if (fill) {
      size_t take = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, take); p += take; n -= take; fill += take;
      if (fill < 64) return;
      block(buf);
    }
#if defined(BPPP_SHA_NI)
    if (n >= 64 && sha256_have_shani()) { const size_t nb = n / 64; sha256_blocks_shani(h, p, nb); p += 64 * nb; n -= 64 * nb; }
#endif
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) memcpy(buf, p, n);
  }
  void finish(uint32_t out_h[8]) {
    uint64_t bits = len * 8;
    uint8_t pad[72] = {0x80};
    size_t fill = (size_t)(len & 63), padlen = (fill < 56 ? 56 - fill : 120 - fill);
    update(pad, padlen);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) out_h[i] = h[i];
  }
};

}  // namespace bppp
