// gather_locality.hip — how fast does gfx950 serve random 64-byte gathers (one comb-table entry per lane) as a function of the table size and of
// whether the 64 lanes of a wavefront gather from ONE 256-KB table row (same term, same window: lane = instance) or from 64 different rows
// (lane = term, the mapping of k_comb_msm)?   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather benchmarks/gather_locality.hip && /tmp/gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// rows = table bytes / 256 KB; entries of 64 B, 4096 per row.  mode 0: every lane its own row; mode 1: one row per wavefront and step
template <int INFLIGHT>
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ tab, uint64_t rows, int mode, int steps, int filler, uint32_t *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0, f = lane;
  for (int s = 0; s < steps; s += INFLIGHT) {
    uint4 v[INFLIGHT][4];
#pragma unroll
    for (int j = 0; j < INFLIGHT; j++) {
      const uint64_t hw = mix(((uint64_t)wave << 24) ^ (uint64_t)(s + j)), hl = mix(hw ^ ((uint64_t)lane << 48) ^ 0x9e37u);
      const uint64_t row = (mode ? hw : hl) % rows, ent = (hl >> 20) & 4095u;
      const uint4 *p = tab + (row * 4096 + ent) * 4;
      v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2]; v[j][3] = p[3];
    }
    for (int k = 0; k < filler; k++) f = f * 1664525u + 1013904223u;       // VALU work between gathers (dependent chain: ~2 instructions per iteration)
#pragma unroll
    for (int j = 0; j < INFLIGHT; j++) acc += v[j][0].x ^ v[j][1].y ^ v[j][2].z ^ v[j][3].w;
  }
  if (acc + f == 0x12345u) out[0] = acc;
}
int main(int argc, char **argv) {
  const double gbs[] = {1.0, 8.0, 41.0, 146.0};
  uint32_t *out; CK(hipMalloc(&out, 64));
  for (double gb : gbs) {
    const uint64_t rows = (uint64_t)(gb * (1ull << 30)) >> 18;
    uint4 *tab;
    if (hipMalloc(&tab, rows << 18) != hipSuccess) { printf("%.0f GB: allocation failed\n", gb); continue; }
    CK(hipMemset(tab, 1, rows << 18));
    for (int filler : {0, 1400}) for (int mode = 0; mode < 2; mode++) {
      const int steps = filler ? 64 : 256, blocks = 256 * 8;               // 2 waves per SIMD with the filler loop (as k_comb_msm<2>), 8192 wavefronts
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      k_gather<1><<<blocks, 256>>>(tab, rows, mode, 8, filler, out);
      CK(hipEventRecord(a));
      k_gather<1><<<blocks, 256>>>(tab, rows, mode, steps, filler, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double g = (double)blocks * 256 * steps;
      printf("table %5.0f GB  %s  filler %4d: %7.3f ms  %6.2f G gathers/s  %7.1f GB/s\n", gb, mode ? "one row per wavefront " : "one row per lane      ", filler, ms, g / ms / 1e6, g * 64 / ms / 1e6);
      fflush(stdout);
    }
    CK(hipFree(tab));
  }
  return 0;
}
