// Host-side helpers of the window combine and the small-batch verifier:
//   from_limbs26 (csrc/hostmath.hpp) — a lazily reduced 10 x 26-bit device value -> canonical Fq — against the multiply-add definition
//   sum n[i] 2^(26 i) mod p evaluated by Horner's rule with the library's own field multiply (the routine it replaced);
//   HostPool (csrc/hostpool.hpp) — every item exactly once, on more than one thread, across many calls.
#include <atomic>
#include <cstdio>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <vector>
#include "hostmath.hpp"
#include "hostpool.hpp"
using namespace bppp_host;
static U256 horner(const uint32_t *n) {
  const Mod &M = FQ();
  U256 acc = U256::zero(), radix = U256::from_u64(1ull << 26);
  for (int i = 9; i >= 0; i--) acc = madd(mmul(acc, radix, M), U256::from_u64(n[i]), M);
  return acc;
}
int main() {
  int bad = 0;
  std::mt19937_64 g(11);
  for (int it = 0; it < 400000; it++) {
    uint32_t n[10];
    for (int i = 0; i < 10; i++) {
      const uint64_t r = g();
      n[i] = (it & 3) == 0 ? 0xFFFFFFFFu : (it & 3) == 1 ? (uint32_t)r : (uint32_t)(r & 0x3FFFFFF) + ((it & 4) ? 0u : (uint32_t)(r >> 60));
    }
    if (it == 1) { for (int i = 0; i < 10; i++) n[i] = 0x3FFFFFF; n[9] = 0x3FFFFF; }                       // 2^256 - 1
    if (it == 2) { for (int i = 0; i < 10; i++) n[i] = 0x3FFFFFF; n[0] = 0x3FFFC2F; n[1] = 0x3FFFFBF; n[9] = 0x3FFFFF; }   // p itself -> 0
    if (it == 3) for (int i = 0; i < 10; i++) n[i] = 0;
    const U256 a = horner(n), b = from_limbs26(n);
    if (cmp(a, b) != 0 || cmp(b, FQ().m) >= 0) { if (bad < 5) printf("from_limbs26 mismatch at case %d\n", it); bad++; }
  }
  { uint32_t n[10]; for (int i = 0; i < 10; i++) n[i] = 0x3FFFFFF; n[0] = 0x3FFFC2F; n[1] = 0x3FFFFBF; n[9] = 0x3FFFFF;
    if (!from_limbs26(n).is_zero()) { printf("p does not reduce to 0\n"); bad++; } }
  // ---- the pool
  {
    bppp::HostPool pool(5);
    std::mutex m;
    std::set<std::thread::id> ids;
    for (int call = 0; call < 300; call++) {
      const size_t count = call % 37;                        // includes 0 and 1 (run inline)
      std::vector<std::atomic<int>> hit(count ? count : 1);
      for (auto &h : hit) h.store(0);
      const std::function<void(size_t)> f = [&](size_t i) {
        hit[i].fetch_add(1);
        if (call == 36) { volatile unsigned spin = 0; for (int k = 0; k < 200000; k++) spin = spin + (unsigned)k; }   // long enough for the workers to join in
        if (call == 36) { std::lock_guard<std::mutex> lk(m); ids.insert(std::this_thread::get_id()); }
      };
      pool.run(count, f);
      for (size_t i = 0; i < count; i++) if (hit[i].load() != 1) { if (bad < 5) printf("pool: item %zu of call %d ran %d times\n", i, call, hit[i].load()); bad++; }
    }
    if (std::thread::hardware_concurrency() > 1 && ids.size() < 2) { printf("pool: all items of a 36-item call ran on one thread\n"); bad++; }
  }
  { bppp::HostPool none(0); int s = 0; none.run(7, [&](size_t i) { s += (int)i; }); if (s != 21) { printf("pool without workers\n"); bad++; } }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad ? 1 : 0;
}
