#include <cstdio>
#include <cstdint>
#include <random>
#include <chrono>
#include "hostmath.hpp"
using namespace bppp_host;
int main() {
  std::mt19937_64 g(1);
  const Mod &M = FR();
  long bad = 0; const int N = 200000;
  std::vector<U256> xs;
  for (int i = 0; i < N; i++) {
    U256 x;
    int kind = i % 8;
    for (int k = 0; k < 4; k++) x.w[k] = g();
    if (kind == 1) { x.w[3] = 0; x.w[2] = 0; }          // small
    if (kind == 2) { x.w[3] = 0; x.w[2] = 0; x.w[1] = 0; }
    if (kind == 3) { x = M.m; x.w[0] -= 1 + (g() & 0xffff); }   // near n
    if (kind == 4) { memset(x.w, 0, 32); x.w[(g() % 4)] = 1ull << (g() % 64); }
    if (kind == 5) { x = M.m; for (int k = 3; k >= 0; k--) { x.w[k] >>= 1; if (k) x.w[k] |= M.m.w[k] << 63 ? 0 : 0; } x.w[0] += g() & 0xff; }
    if (cmp(x, M.m) >= 0) { x.w[3] >>= 1; }
    xs.push_back(x);
  }
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::pair<SInt,SInt>> A(N), B(N);
  for (int i = 0; i < N; i++) A[i] = rational_reduce_scalar_plain(xs[i]);
  auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) B[i] = rational_reduce_scalar(xs[i]);
  auto t2 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) {
    if (memcmp(A[i].first.m, B[i].first.m, 40) || memcmp(A[i].second.m, B[i].second.m, 40) || A[i].first.neg != B[i].first.neg || A[i].second.neg != B[i].second.neg) { if (bad < 5) printf("mismatch at %d kind %d\n", i, i % 8); bad++; }
  }
  printf("bad %ld; plain %.2f us, fast %.2f us per call\n", bad, std::chrono::duration<double>(t1 - t0).count() / N * 1e6, std::chrono::duration<double>(t2 - t1).count() / N * 1e6);
  return bad != 0;
}
