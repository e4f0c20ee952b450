// valu_microbench.hip — measures per-instruction VALU throughput on gfx950 for the integer ops a
// 256-bit modular multiply can be built from ("this number sets the real
// ceiling for everything").  Build: hipcc -O3 --offload-arch=gfx950 -o valu_microbench valu_microbench.hip
// Prints, per op: cycles per wave-instruction per SIMD at full occupancy (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 4096
#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u;
  uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 3;
  uint32_t d0 = a, d1 = b, d2 = a + 7, d3 = b + 9;
  double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5;
  for (int i = 0; i < ITER; i++) {
    if (OP == 0) { REP16(asm volatile("v_mad_u64_u32 %0, s[6:7], %4, %5, %0\n v_mad_u64_u32 %1, s[6:7], %4, %5, %1\n v_mad_u64_u32 %2, s[6:7], %4, %5, %2\n v_mad_u64_u32 %3, s[6:7], %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "s6", "s7");) }
    if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 2) { REP16(asm volatile("v_mul_hi_u32 %0, %4, %0\n v_mul_hi_u32 %1, %4, %1\n v_mul_hi_u32 %2, %4, %2\n v_mul_hi_u32 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 3) { REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    if (OP == 4) { REP16(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f2), "v"(f3));) }
    if (OP == 5) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c1));) }
    if (OP == 6) { REP16(asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 7) { REP16(asm volatile("v_add_co_u32 %0, vcc, %4, %0\n v_addc_co_u32 %1, vcc, %4, %1, vcc\n v_addc_co_u32 %2, vcc, %4, %2, vcc\n v_addc_co_u32 %3, vcc, %4, %3, vcc" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a) : "vcc");) }
    if (OP == 8) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 9) { REP16(asm volatile("v_mul_u32_u24 %0, %4, %0\n v_mul_hi_u32_u24 %1, %4, %1\n v_mul_u32_u24 %2, %4, %2\n v_mul_hi_u32_u24 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 10) { REP16(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));) }
    if (OP == 11) { REP16(asm volatile("v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %4, %5, %1\n v_mad_i32_i24 %2, %4, %5, %2\n v_mad_i32_i24 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    if (OP == 12) { REP16(asm volatile("v_add3_u32 %0, %4, %5, %0\n v_add3_u32 %1, %4, %5, %1\n v_add3_u32 %2, %4, %5, %2\n v_add3_u32 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    if (OP == 14) { REP16(asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 7\n v_alignbit_b32 %3, %3, %3, 7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 15) { REP16(asm volatile("v_bitop3_b32 %0, %4, %5, %0 bitop3:0x96\n v_bitop3_b32 %1, %4, %5, %1 bitop3:0x96\n v_bitop3_b32 %2, %4, %5, %2 bitop3:0x96\n v_bitop3_b32 %3, %4, %5, %3 bitop3:0x96" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    if (OP == 16) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 25, %4\n v_lshl_or_b32 %1, %1, 25, %4\n v_lshl_or_b32 %2, %2, 25, %4\n v_lshl_or_b32 %3, %3, 25, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 17) { REP16(asm volatile("v_lshrrev_b32 %0, 7, %0\n v_lshrrev_b32 %1, 7, %1\n v_lshrrev_b32 %2, 7, %2\n v_lshrrev_b32 %3, 7, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 18) { REP16(asm volatile("v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n v_xor_b32 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
    if (OP == 19) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    if (OP == 13) { REP16(asm volatile("v_mad_u64_u32 %0, s[6:7], %4, %5, %1\n v_mad_u64_u32 %1, s[6:7], %4, %5, %2\n v_mad_u64_u32 %2, s[6:7], %4, %5, %3\n v_mad_u64_u32 %3, s[6:7], %4, %5, %0" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "s6", "s7");) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 + c1 + c2 + c3) + d0 + d1 + d2 + d3 + (uint32_t)(f0 + f1 + f2 + f3);
}

template <int OP> void run(const char *name, int waves_per_simd) {
  int cus = 256;
  int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, 2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_wave = (double)ITER * 16 * 4;
  // per SIMD: waves_per_simd waves each issuing insts_per_wave; time ms
  double cyc = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 8}) {
    run<0>("v_mad_u64_u32 (4 chains)", w);
    run<13>("v_mad_u64_u32 (rotating)", w);
    run<1>("v_mul_lo_u32", w);
    run<2>("v_mul_hi_u32", w);
    run<3>("v_mad_u32_u24", w);
    run<11>("v_mad_i32_i24", w);
    run<9>("v_mul(_hi)_u32_u24", w);
    run<4>("v_fma_f64", w);
    run<5>("v_lshl_add_u64", w);
    run<10>("v_lshrrev_b64", w);
    run<6>("v_add_u32", w);
    run<12>("v_add3_u32", w);
    run<7>("v_add_co/addc_co chain", w);
    run<8>("v_mov_b32", w);
    run<14>("v_alignbit_b32", w);
    run<15>("v_bitop3_b32", w);
    run<16>("v_lshl_or_b32", w);
    run<17>("v_lshrrev_b32", w);
    run<18>("v_xor_b32", w);
    run<19>("v_perm_b32", w);
  }
  return 0;
}
