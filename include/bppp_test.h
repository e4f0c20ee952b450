/*
 * bppp_test.h — test-only hooks exported by libbppp_hip_test.so (a separate library; the product library does not contain them) so the parity tests can exercise the
 * device field and group arithmetic directly (the device counterparts of mulField# / addField# /
 * invField#, src/Data/Field/Galois/FastPrime/Internal.hs:909-988, and of nrmlAdd / dbl',
 * src/Commitment.hs:111-144).  Not part of the drop-in boundary.
 */
#ifndef BPPP_TEST_H
#define BPPP_TEST_H
#include "bppp.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
#define BPPP_FE_ADD 0
#define BPPP_FE_SUB 1
#define BPPP_FE_MUL 2
#define BPPP_FE_SQR 3
#define BPPP_FE_INV 4
#define BPPP_FE_NEG 5
/* out[i] = a[i] (op) b[i] in Fq (modulus = 0: production 10x26 limbs; 2: the 8x32 code path) or Fr (modulus = 1: 8x32; 3: production
 * 10x26 limbs of csrc/fr26.hip.h, whose ops 6 and 10-14 drive the lazy limbs to their magnitude bounds); host arrays of n x 4 uint64 */
int bppp_test_fe_op(bppp_ctx *ctx, int op, int modulus, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out);
/* out[i] = p[i] + q[i] (complete group law; op 0: mixed XYZZ+affine, op 1: XYZZ+XYZZ, op 2: 2*p[i]);
 * host arrays of n x 8 uint64 affine points */
int bppp_test_point_op(bppp_ctx *ctx, int op, const uint64_t *p, const uint64_t *q, size_t n, uint64_t *out);
/* Measured ceiling of the field layer: modular multiplications per second of a kernel that does nothing but independent
 * Fq multiplications (10x26-bit limbs) at 8 wavefronts per SIMD.  bench.py quotes the MSM's multiplication rate against it. */
int bppp_test_mulmod_rate(bppp_ctx *ctx, int iters, double *mulmods_per_sec);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
