// fqmul_variants.hip — the forms of the Fq multiplication (csrc/fq26.hip.h, benchmarks/fqmul_variants.h) against each other: same results on
// random and worst-case-magnitude inputs, multiplications per second at 8 wavefronts per SIMD (the bppp_test_mulmod_rate shape: four
// independent chains per lane).   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o benchmarks/fqmul_variants benchmarks/fqmul_variants.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "fqmul_variants.h"
using namespace bppp;

template <int V> BPPP_DI fq mulv(const fq &a, const fq &b) {
  if (V == 1) return fq_mul(a, b);
  if (V == 2) return fq_mul_v2(a, b);
  if (V == 3) return fq_mul_v3(a, b);
  return fq_mul_cols(a, b);
}
template <int V> __global__ void __launch_bounds__(256) k_rate(const uint32_t *__restrict__ seed, int iters, uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  fq a[4], b;
  for (int k = 0; k < 4; k++) a[k] = fq_from_fe(fe_load(seed + (size_t)((t + 17 * k) & 1023) * 8));
  b = fq_from_fe(fe_load(seed + (size_t)((t * 7 + 3) & 1023) * 8));
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = mulv<V>(a[k], b);
  }
  fq r = fq_add(fq_add(a[0], a[1]), fq_add(a[2], a[3]));
  fe_store(out + (size_t)t * 8, fq_to_fe(r));
}
// worst-case magnitudes: (8 a) * (-7 b)
template <int V> __global__ void k_mag(const uint32_t *__restrict__ seed, uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const fq x = fq_from_fe(fe_load(seed + (size_t)(t & 1023) * 8)), y = fq_from_fe(fe_load(seed + (size_t)((t * 5 + 1) & 1023) * 8));
  fe_store(out + (size_t)t * 8, fq_to_fe(mulv<V>(fq_mul_int(x, 8), fq_neg<7>(fq_mul_int(y, 7)))));
}
template <int V> double run(const uint32_t *seed, uint32_t *out, std::vector<uint32_t> &res, std::vector<uint32_t> &mag) {
  const int blocks = 256 * 4 * 8 / 4 * 2, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_rate<V><<<blocks, 256>>>(seed, 8, out);
  hipEventRecord(e0);
  k_rate<V><<<blocks, 256>>>(seed, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  res.resize((size_t)blocks * 256 * 8); hipMemcpy(res.data(), out, res.size() * 4, hipMemcpyDeviceToHost);
  k_mag<V><<<64, 256>>>(seed, out);
  mag.resize(64 * 256 * 8); hipMemcpy(mag.data(), out, mag.size() * 4, hipMemcpyDeviceToHost);
  return (double)blocks * 256.0 * 4.0 * iters / (ms * 1e-3);
}
int main() {
  uint32_t *seed, *out;
  hipMalloc(&seed, 1024 * 32); hipMalloc(&out, (size_t)16384 * 256 * 32);
  std::vector<uint32_t> h(1024 * 8);
  uint64_t z = 0x9E3779B97F4A7C15ull;
  for (auto &w : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; w = (uint32_t)(z >> 16); }
  for (int i = 0; i < 1024; i++) h[8 * i + 7] &= 0x7FFFFFFFu;
  for (int k = 0; k < 8; k++) { h[k] = k < 7 ? 0xFFFFFFFFu : 0x7FFFFFFFu; h[8 + k] = 0; }     // edge operands
  h[16] = 0xFFFFFC2Eu; h[17] = 0xFFFFFFFEu; for (int k = 2; k < 8; k++) h[16 + k] = 0xFFFFFFFFu;   // p - 1
  hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> r0, r1, r2, m0, m1, m2, r3, m3;
  const double g0 = run<0>(seed, out, r0, m0), g1 = run<1>(seed, out, r1, m1), g2 = run<2>(seed, out, r2, m2), g3 = run<3>(seed, out, r3, m3);
  printf("fq_mul_cols (rounds 1-3) %.1f G mulmod/s\n", g0 / 1e9);
  printf("fq_mul (two chains)      %.1f G mulmod/s  results %s, magnitude-8 %s\n", g1 / 1e9, r1 == r0 ? "equal" : "DIFFER", m1 == m0 ? "equal" : "DIFFER");
  printf("v2 (register R0, R1)     %.1f G mulmod/s  results %s, magnitude-8 %s\n", g2 / 1e9, r2 == r0 ? "equal" : "DIFFER", m2 == m0 ? "equal" : "DIFFER");
  printf("v3 (H chain + free low)   %.1f G mulmod/s  results %s, magnitude-8 %s\n", g3 / 1e9, r3 == r0 ? "equal" : "DIFFER", m3 == m0 ? "equal" : "DIFFER");
  return (r3 == r0 && m3 == m0 && r1 == r0 && r2 == r0 && m1 == m0 && m2 == m0) ? 0 : 1;
}
