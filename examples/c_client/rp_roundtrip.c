/* rp_roundtrip.c — the C ABI of libbppp_hip.so used from plain C99, no Python and no HIP headers: what a binding in the reference's
 * own language (the `foreign import ccall` stubs of INTEGRATION.md) would do, end to end.
 *
 *   setup   the schema of examples/64bit (one 64-bit value, base 16, an output; app/Parse.hs:125-172) taken four times, both argument
 *           flavours; the basis = getPoints-style lifts of pseudo-random x (app/Main.hs:68-72) made on the GPU (bppp_lift_x_device)
 *   prove   bppp_rp_prove_batch: proveM (src/RangeProof.hs:93-97), five proofs in lockstep -> the reference's two files per proof
 *   verify  bppp_rp_verify_batch on those files (host buffers in, accept / reject out), then again with one bit of one proof flipped:
 *           the batch must be rejected and exactly that proof reported
 *
 * Build and run (tests/test_gpu_c_client.py does this under -m gpu):
 *   gcc -std=c99 -O2 -Iinclude examples/c_client/rp_roundtrip.c -Lbulletproofspp_amd/lib -lbppp_hip -Wl,-rpath,$PWD/bulletproofspp_amd/lib -o rp_roundtrip
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bppp.h"

static uint64_t sm_state = 0x0123456789ABCDEFull;
static uint64_t splitmix(void) {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static bppp_ctx *ctx = NULL;
#define CHECK(call)                                                                                       \
  do {                                                                                                    \
    int rc_ = (call);                                                                                     \
    if (rc_) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, ctx ? bppp_last_error(ctx) : ""); return 1; } \
  } while (0)
#define EXPECT(cond)                                                                                      \
  do { if (!(cond)) { fprintf(stderr, "%s:%d expectation failed: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

/* npoints points on the curve: lift 3x as many candidates on the device, keep the hits */
static int make_basis(size_t npoints, uint64_t *out_xy) {
  const size_t cand = 3 * npoints + 64;
  uint64_t *xs = (uint64_t *)malloc(cand * 32), *pts = (uint64_t *)malloc(cand * 64);
  void *d_x = NULL, *d_p = NULL;
  size_t i, got = 0;
  if (!xs || !pts) return 1;
  for (i = 0; i < cand * 4; i++) xs[i] = splitmix();
  CHECK(bppp_device_alloc(ctx, cand * 32, &d_x));
  CHECK(bppp_device_alloc(ctx, cand * 64, &d_p));
  CHECK(bppp_upload(ctx, d_x, xs, cand * 32));
  CHECK(bppp_lift_x_device(ctx, d_x, cand, d_p));
  CHECK(bppp_download(ctx, pts, d_p, cand * 64));
  for (i = 0; i < cand && got < npoints; i++) {
    uint64_t any = 0;
    int k;
    for (k = 0; k < 8; k++) any |= pts[8 * i + k];
    if (any) memcpy(out_xy + 8 * got++, pts + 8 * i, 64);
  }
  CHECK(bppp_device_free(ctx, d_x));
  CHECK(bppp_device_free(ctx, d_p));
  free(xs); free(pts);
  return got == npoints ? 0 : 1;
}

static int round_trip(int flavour) {
  enum { NR = 4, B = 5, PLEN = 16 };
  bppp_rp_range ranges[NR];
  bppp_rp_shape shp, shp2;
  bppp_rp *rp = NULL;
  uint64_t *basis, amounts[B][NR][4], types[B][NR][4], blinds[B][NR][4];
  uint8_t prefix[B * PLEN], seed[32], *coms, *proofs;
  uint32_t status[B];
  int accept = -1, b, r, k;
  size_t npoints;
  FILE *ur;

  memset(ranges, 0, sizeof ranges);
  for (r = 0; r < NR; r++) { ranges[r].base = 16; ranges[r].flags = BPPP_RP_OUTPUT; ranges[r].max[1] = 1; }     /* [0, 2^64) */
  CHECK(bppp_rp_shape_of(flavour, 0, ranges, NR, &shp));
  npoints = 2 + shp.lin_len + shp.norm_len;
  basis = (uint64_t *)malloc(npoints * 64);
  EXPECT(basis != NULL);
  EXPECT(make_basis(npoints, basis) == 0);
  CHECK(bppp_rp_create(ctx, flavour, 0, ranges, NR, NULL, 0, basis, npoints, "c client", &rp));
  CHECK(bppp_rp_info(rp, &shp2));
  EXPECT(shp2.norm_len == shp.norm_len && shp2.lin_len == shp.lin_len && shp2.rounds == shp.rounds && shp2.proof_bytes == shp.proof_bytes);

  memset(types, 0, sizeof types);
  memset(amounts, 0, sizeof amounts);
  for (b = 0; b < B; b++)
    for (r = 0; r < NR; r++) {
      amounts[b][r][0] = b == 0 ? (r == 0 ? 0 : r == 1 ? ~0ull : splitmix()) : splitmix();          /* the range's two ends included */
      for (k = 0; k < 4; k++) blinds[b][r][k] = splitmix();
      blinds[b][r][3] >>= 1;                                                                        /* < 2^255 < n */
    }
  for (b = 0; b < B; b++) {                                /* the prover's randomness: a distinct 16-byte prefix per proof */
    char tmp[32];
    snprintf(tmp, sizeof tmp, "c client %07d", b);
    memcpy(prefix + b * PLEN, tmp, PLEN);
  }
  coms = (uint8_t *)calloc(B, shp.coms_bytes);
  proofs = (uint8_t *)calloc(B, shp.proof_bytes);
  EXPECT(coms && proofs);
  CHECK(bppp_rp_prove_batch(rp, B, &amounts[0][0][0], &types[0][0][0], &blinds[0][0][0], prefix, PLEN, coms, proofs));

  ur = fopen("/dev/urandom", "rb");                       /* the verifier's weights need fresh secret randomness */
  EXPECT(ur && fread(seed, 1, 32, ur) == 32);
  fclose(ur);
  CHECK(bppp_rp_verify_batch(rp, B, coms, proofs, seed, &accept, status, NULL, NULL));
  EXPECT(accept == 1);
  for (b = 0; b < B; b++) EXPECT(status[b] == BPPP_RP_VALID);

  proofs[3 * shp.proof_bytes + 7] ^= 1;                    /* one bit of proof 3's first witness scalar */
  CHECK(bppp_rp_verify_batch(rp, B, coms, proofs, seed, &accept, status, NULL, NULL));
  EXPECT(accept == 0);
  for (b = 0; b < B; b++) EXPECT(status[b] == (b == 3 ? BPPP_RP_INVALID : BPPP_RP_VALID));
  proofs[3 * shp.proof_bytes + 7] ^= 1;

  /* an amount outside its range has no witness: the prover must refuse it, not emit a proof */
  amounts[2][1][1] = 1;                                    /* 2^64 + something */
  EXPECT(bppp_rp_prove_batch(rp, B, &amounts[0][0][0], &types[0][0][0], &blinds[0][0][0], prefix, PLEN, coms, proofs) != BPPP_OK);

  printf("flavour %d: %d proofs (%zu + %zu bytes each, %zu rounds) proved, verified, tampering identified, out-of-range refused\n", flavour, (int)B,
         shp.coms_bytes, shp.proof_bytes, shp.rounds);
  bppp_rp_destroy(rp);
  free(basis); free(coms); free(proofs);
  return 0;
}

int main(void) {
  /* 2 G = G + G through the MSM entry point with host buffers: scalars (1, 1) on (G, G) against scalar 2 on G */
  static const uint64_t G[8] = {0x59F2815B16F81798ull, 0x029BFCDB2DCE28D9ull, 0x55A06295CE870B07ull, 0x79BE667EF9DCBBACull,
                                0x9C47D08FFB10D4B8ull, 0xFD17B448A6855419ull, 0x5DA4FBFC0E1108A8ull, 0x483ADA7726A3C465ull};
  uint64_t sc[8] = {1, 0, 0, 0, 1, 0, 0, 0}, two[4] = {2, 0, 0, 0}, pts[16], a[8], c[8];
  CHECK(bppp_ctx_create(0, &ctx));
  memcpy(pts, G, 64); memcpy(pts + 8, G, 64);
  CHECK(bppp_msm(ctx, sc, pts, 2, a));
  CHECK(bppp_msm(ctx, two, G, 1, c));
  EXPECT(memcmp(a, c, 64) == 0 && (a[0] | a[1] | a[2] | a[3]) != 0);
  if (round_trip(0)) return 1;          /* norm-linear argument */
  if (round_trip(1)) return 1;          /* inner-product argument (the CLI's default) */
  bppp_ctx_destroy(ctx);
  printf("c client ok (%s)\n", bppp_version());
  return 0;
}
