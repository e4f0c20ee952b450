// candidate forms of the Fq multiplication measured against each other by benchmarks/fqmul_variants.hip
//   fq_mul_cols  (csrc/fq26.hip.h)  rounds 1-3: plain C columns, the compiler's order
//   fq_mul       (csrc/fq26.hip.h)  round 4 production: two pinned carry chains, carries in the products' addends, R0 / R1 in scalar registers
//   fq_mul_v2    (here)             the round 1-3 columns with only R0 / R1 moved to registers
#pragma once
#include "../bulletproofspp_amd/csrc/fq26.hip.h"
namespace bppp {
BPPP_DI fq fq_mul_v2(const fq &a, const fq &b) {
  const uint32_t R0 = fq_sreg(FQ_R0), R1 = fq_sreg(FQ_R1);
  uint64_t d = 0;
  FQ_COL(d, 9)
  const uint32_t t9 = (uint32_t)d & FQ_M26; d >>= 26;
  uint32_t u[9];
#pragma unroll
  for (int k = 10; k <= 18; k++) {
    FQ_COL(d, k)
    u[k - 10] = (uint32_t)d & FQ_M26; d >>= 26;
  }
  const uint64_t u9 = d;
  fq r;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    FQ_COL(c, k)
    c += (uint64_t)u[k] * R0;
    if (k) c += (uint64_t)u[k - 1] * R1;
    r.n[k] = (uint32_t)c & FQ_M26; c >>= 26;
  }
  fq_mul_tail(r, c, t9, u9, u[8]);
  return r;
}
// v3: H chain pinned as in production; the 45 low products as free, unpinned column sums (they fill the slots between H's dependent
// products), then a pinned fold chain  c = u_k R0 + carry; c += u_(k-1) R1; c += P_k  (one explicit 64-bit add per low column)
BPPP_DI fq fq_mul_v3(const fq &a, const fq &b) {
  const uint32_t R0 = fq_sreg(FQ_R0), R1 = fq_sreg(FQ_R1);
  uint64_t P[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { uint64_t c = 0; FQ_COL(c, k) P[k] = c; }
  uint64_t d = 0;
#pragma unroll
  for (int i = 0; i <= 9; i++) d = fq_madc(a.n[i], b.n[9 - i], d);
  const uint32_t t9 = (uint32_t)d & FQ_M26; d >>= 26;
  uint32_t u[9];
#pragma unroll
  for (int k = 10; k <= 18; k++) {
#pragma unroll
    for (int i = k - 9; i <= 9; i++) d = fq_madc(a.n[i], b.n[k - i], d);
    u[k - 10] = (uint32_t)d & FQ_M26; d >>= 26;
  }
  const uint64_t u9 = d;
  fq r;
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    c = fq_madc(u[k], R0, c);
    if (k) c = fq_madc(u[k - 1], R1, c);
    c += P[k];
    r.n[k] = (uint32_t)c & FQ_M26; c >>= 26;
  }
  fq_mul_tail(r, c, t9, u9, u[8]);
  return r;
}
}  // namespace bppp
