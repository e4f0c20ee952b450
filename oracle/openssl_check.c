/*
 * openssl_check.c — independent secp256k1 MSM via OpenSSL libcrypto (TEST INFRASTRUCTURE ONLY).
 * Reads from stdin: n, then n lines "scalar_hex x_hex y_hex" (big-endian hex, point (0,0) = inf);
 * prints "x_hex y_hex" of sum_i scalar_i * P_i (or "0 0" for infinity).
 * Used by tests/test_oracle_openssl.py to pin oracle/bppp_oracle.c's group law.
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/obj_mac.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void) {
  EC_GROUP *g = EC_GROUP_new_by_curve_name(NID_secp256k1);
  BN_CTX *ctx = BN_CTX_new();
  EC_POINT *acc = EC_POINT_new(g), *p = EC_POINT_new(g), *t = EC_POINT_new(g);
  EC_POINT_set_to_infinity(g, acc);
  long n; if (scanf("%ld", &n) != 1) return 2;
  char sh[80], xh[80], yh[80];
  for (long i = 0; i < n; i++) {
    if (scanf("%79s %79s %79s", sh, xh, yh) != 3) return 2;
    BIGNUM *s = NULL, *x = NULL, *y = NULL;
    BN_hex2bn(&s, sh); BN_hex2bn(&x, xh); BN_hex2bn(&y, yh);
    if (BN_is_zero(x) && BN_is_zero(y)) { BN_free(s); BN_free(x); BN_free(y); continue; }
    if (!EC_POINT_set_affine_coordinates(g, p, x, y, ctx)) { fprintf(stderr, "bad point %ld\n", i); return 3; }
    EC_POINT_mul(g, t, NULL, p, s, ctx);
    EC_POINT_add(g, acc, acc, t, ctx);
    BN_free(s); BN_free(x); BN_free(y);
  }
  if (EC_POINT_is_at_infinity(g, acc)) { printf("0 0\n"); return 0; }
  BIGNUM *x = BN_new(), *y = BN_new();
  EC_POINT_get_affine_coordinates(g, acc, x, y, ctx);
  char *xs = BN_bn2hex(x), *ys = BN_bn2hex(y);
  printf("%s %s\n", xs, ys);
  return 0;
}
