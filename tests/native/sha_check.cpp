// SHA-256 on the host: the SHA-extension path against the portable rounds and the FIPS 180-4 vectors.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "sha256.hip.h"
using namespace bppp;
static void portable(const uint8_t *p, size_t n, uint32_t out[8]) {       // the portable compression only
  uint32_t h[8]; sha256_init(h);
  std::vector<uint8_t> m(p, p + n);
  m.push_back(0x80); while (m.size() % 64 != 56) m.push_back(0);
  for (int i = 0; i < 8; i++) m.push_back((uint8_t)(((uint64_t)n * 8) >> (56 - 8 * i)));
  for (size_t o = 0; o < m.size(); o += 64) {
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)m[o + 4 * i] << 24) | ((uint32_t)m[o + 4 * i + 1] << 16) | ((uint32_t)m[o + 4 * i + 2] << 8) | m[o + 4 * i + 3];
    sha256_compress(h, w);
  }
  memcpy(out, h, 32);
}
int main() {
  int bad = 0;
  { Sha256 h; h.update("abc", 3); uint32_t d[8]; h.finish(d);
    const uint32_t want[8] = {0xba7816bfu, 0x8f01cfeau, 0x414140deu, 0x5dae2223u, 0xb00361a3u, 0x96177a9cu, 0xb410ff61u, 0xf20015adu};
    if (memcmp(d, want, 32)) { printf("abc mismatch\n"); bad++; } }
  std::mt19937_64 g(3);
  for (int it = 0; it < 2000; it++) {
    const size_t n = it < 300 ? (size_t)it : (size_t)(g() % 20000);
    std::vector<uint8_t> m(n);
    for (auto &b : m) b = (uint8_t)g();
    Sha256 h;                                     // streamed in random pieces
    size_t o = 0;
    while (o < n) { size_t k = 1 + g() % 200; if (k > n - o) k = n - o; h.update(m.data() + o, k); o += k; }
    uint32_t d[8], w[8];
    h.finish(d);
    portable(m.data(), n, w);
    if (memcmp(d, w, 32)) { if (bad < 5) printf("mismatch at length %zu\n", n); bad++; }
  }
  // the compression split into its two halves (message schedule + K / the 64 rounds), as the wavefront pairs of the device use it
  for (int it = 0; it < 2000; it++) {
    uint32_t w[16], w2[16], h1[8], h2[8], wk[64];
    for (int i = 0; i < 16; i++) w2[i] = w[i] = (uint32_t)g();
    for (int i = 0; i < 8; i++) h2[i] = h1[i] = (uint32_t)g();
    sha256_compress(h1, w);
    sha256_schedule_wk(w2, [&](int t, uint32_t v) { wk[t] = v; });
    sha256_rounds_wk(h2, [&](int t) { return wk[t]; });
    if (memcmp(h1, h2, 32)) { if (bad < 5) printf("split compression mismatch\n"); bad++; }
  }
#if defined(BPPP_SHA_NI)
  printf("bad %d; sha extensions %s\n", bad, sha256_have_shani() ? "used" : "absent");
#else
  printf("bad %d; sha extensions not compiled\n", bad);
#endif
  return bad != 0;
}
