// rpprove_host.hpp — host-side pieces shared by the lockstep provers' host routes (csrc/rpprove.hip: typed-reciprocal, csrc/brpprove.hip:
// binary, csrc/ipb_host.hip: the inner-product argument on the host cores): scalar-field shorthands over csrc/hostmath.hpp, the CLI's oracle and
// randomness (shaOracle, hashToScalar: app/Main.hs:64-87; ZKPT, src/ZKP.hs:88-101), batchInverse, and the reference's file encoding
// (src/Encoding.hs:75-86, :130-134).  These routes serve small batches (before a handle has its comb table), digit bases above 256 and the
// BPPP_RP_HOST_ALGEBRA cross-check; the device-resident routes are csrc/rpprove_dev.hip, csrc/brpprove_dev.hip, csrc/ipb.hip.
#pragma once
#include <string.h>
#include <functional>
#include <string>
#include <vector>
#include "rp_internal.hpp"
#include "sha256.hip.h"

namespace bppp_rpp {
using namespace bppp_host;
using bppp::Sha256;
using bppp::sha256_digest_to_limbs;
using bppp_rps::Pos;
using bppp_rps::RangeData;
using bppp_rps::Setup;



inline const Mod &MR() { return FR(); }
inline U256 fa(const U256 &a, const U256 &b) { return madd(a, b, MR()); }
inline U256 fs(const U256 &a, const U256 &b) { return msub(a, b, MR()); }
inline U256 fm(const U256 &a, const U256 &b) { return frmul(a, b); }
inline U256 fneg(const U256 &a) { return mneg(a, MR()); }
inline U256 fdbl(const U256 &a) { return madd(a, a, MR()); }
inline U256 fpow(U256 b, uint64_t e) { U256 r = U256::one(); while (e) { if (e & 1) r = fm(r, b); b = fm(b, b); e >>= 1; } return r; }
inline U256 small(uint64_t v) { return U256::from_u64(v); }

// `show` of a field element: its decimal integer (see sha256_oracle in rangeproof.py / shaOracle, app/Main.hs:75-80)
inline void dec_append(std::string &out, U256 v) {
  char buf[80];
  int n = 0;
  if (v.is_zero()) { out.push_back('0'); return; }
  while (!v.is_zero()) {
    uint64_t rem = 0;
    v = bppp_rps::u_div64(v, 10000000000000000000ull, &rem);
    const bool last = v.is_zero();
    for (int k = 0; k < 19 && (rem || !last); k++) { buf[n++] = (char)('0' + rem % 10); rem /= 10; }
  }
  while (n) out.push_back(buf[--n]);
}
inline void point_text(std::string &out, const uint64_t *xy) { dec_append(out, U256::load(xy)); dec_append(out, U256::load(xy + 4)); }

// digest -> field by Binary (Prime p) (src/Encoding.hs:75-79), toP
inline U256 digest_to_fr(const uint32_t h[8]) {
  uint32_t v[8];
  sha256_digest_to_limbs(h, v);
  U256 r;
  for (int i = 0; i < 4; i++) r.w[i] = ((uint64_t)v[2 * i + 1] << 32) | v[2 * i];
  return bppp_rps::u_mod_n(r);
}

// hashToScalar prefix . show (app/Main.hs:83-87, :189): the prover's randomness, counter from 0 (ZKPT.random, src/ZKP.hs:88-92)
struct Rnd {
  const uint8_t *prefix; size_t plen; uint64_t n = 0;
  U256 next() {
    Sha256 h;
    h.update(prefix, plen);
    const std::string c = std::to_string(n++);
    h.update(c.data(), c.size());
    uint32_t d[8];
    h.finish(d);
    return digest_to_fr(d);
  }
};

struct RPW { U256 sc; std::vector<U256> lin, nrm; };

struct PState {
  std::vector<U256> v, ty, bl;                 // inputs (amount, type, blinding) as field elements
  std::vector<U256> d, mi, pv;                 // per norm position: digit (type for typing), inline multiplicity, ps (amount | 1)
  std::vector<U256> ms_shared;                 // linLen - 6 shared multiplicities, bases in sorted order
  RPW dm, m, r, blw;
  std::vector<U256> u, vv, rr, cc;             // Phase2 (TypedReciprocal.hs:180-181)
  U256 e, x, r0, q, xp, r1, t, e_inv, r0_inv, q0, q0_inv, r1_inv;
  std::vector<U256> shared_cs;
  U256 ns_sc, ns_ty, ns_bl;                    // sum_i inputCoeff_i * (v, ty, bl)_i
  std::vector<std::string> groups;             // transcript text, one string per oracle call, oldest first
  size_t npoints = 0;
  Rnd rnd;
  std::string err;
  std::vector<uint32_t> tmp_ds, tmp_cnt, tmp_ms, tmp_msc;   // make_witness scratch
};

// shaOracle (app/Main.hs:75-80) over ZKPT's transcript (src/ZKP.hs:96-101): the new commitments go IN FRONT; output n hashes
// tag <> show n <> show (length ps) <> text of the whole transcript, newest call first
inline void oracle(const std::string &tag, PState &ps, const uint64_t *const *pts, size_t npts, int count, U256 *out) {
  std::string g;
  g.reserve(npts * 160);
  for (size_t i = 0; i < npts; i++) point_text(g, pts[i]);
  ps.groups.push_back(std::move(g));
  ps.npoints += npts;
  for (int n = 1; n <= count; n++) {
    Sha256 h;
    const std::string hdr = tag + std::to_string(n) + std::to_string(ps.npoints);
    h.update(hdr.data(), hdr.size());
    for (size_t k = ps.groups.size(); k-- > 0;) h.update(ps.groups[k].data(), ps.groups[k].size());
    uint32_t d[8];
    h.finish(d);
    out[n - 1] = digest_to_fr(d);
  }
}


// a^(n-2) with the dedicated multiply; 0 -> 0
inline U256 finv(const U256 &a) {
  U256 e; sub_raw(e, MR().m, U256::from_u64(2));
  U256 acc = U256::one(), base = a;
  for (int i = 0; i < 256; i++) { if (e.bit(i)) acc = fm(acc, base); base = fm(base, base); }
  return acc;
}
// batchInverse (src/Data/Field/BatchInverse.hs:18-39): Montgomery's trick, 0 -> 0
inline void batch_inv(std::vector<U256> &v) {
  const size_t n = v.size();
  if (!n) return;
  std::vector<U256> pre(n);
  U256 acc = U256::one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!v[i].is_zero()) acc = fm(acc, v[i]); }
  U256 y = finv(acc);
  for (size_t i = n; i-- > 0;) {
    if (v[i].is_zero()) continue;
    const U256 inv = fm(y, pre[i]);
    y = fm(y, v[i]);
    v[i] = inv;
  }
}

inline void put_field(uint8_t *dst, const U256 &v) {      // Binary (Prime p) put (Encoding.hs:81-86)
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) dst[8 * i + k] = (uint8_t)(v.w[i] >> (56 - 8 * k));
}
// encodeCommitments (Encoding.hs:130-134): packed sign bits (y > p - y), then the x coordinates
inline void encode_points(uint8_t *dst, const uint64_t *const *pts, size_t n) {
  const size_t ns = (n + 7) / 8;
  memset(dst, 0, ns);
  for (size_t i = 0; i < n; i++) {
    const U256 y = U256::load(pts[i] + 4), ny = mneg(y, FQ());
    if (cmp(y, ny) > 0) dst[i >> 3] |= (uint8_t)(1u << (i & 7));
    put_field(dst + ns + 32 * i, U256::load(pts[i]));
  }
}


}  // namespace bppp_rpp

namespace bppp {
// csrc/rpprove.hip
int rpp_build_fixed_table(bppp_rp *rp);
int rpp_ensure_pwork(bppp_rp *rp, size_t bytes);       // the [3][64][15] fixed-base table of (g, H0, H1) for input commitments before a comb table exists
// csrc/ipb_host.hip: proveBPM of the inner-product flavour with its field algebra on the host cores (the cross-check of csrc/ipb.hip)
int ip_argument_lockstep(bppp_rp *rp, size_t B, size_t k, const uint64_t *psv_in, const uint64_t *rr, const uint64_t *nrm, const uint64_t *lc_in, const uint64_t *lx_in,
                         const std::function<bppp_rpp::PState &(size_t)> &tr_of, uint64_t *resp, uint64_t *wn, uint64_t *wl);
// csrc/brpprove.hip: RangeProof.Binary, host-algebra route and the wrapper of the device-resident one
int prove_batch_binary(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *blinds, const uint8_t *rand_prefix, size_t prefix_len, uint8_t *coms_files,
                       uint8_t *proof_files);
int prove_batch_binary_dev(bppp_rp *rp, size_t batch, const uint64_t *amounts, const uint64_t *blinds, const uint8_t *rand_prefix, size_t prefix_len, uint8_t *coms_files,
                           uint8_t *proof_files, size_t index_base = 0);
}  // namespace bppp

extern "C" {
int bppp_nlb_create(bppp_ctx *ctx, size_t batch, const uint64_t *s, const uint64_t g_xy[8], const uint64_t *q, const uint64_t *norm_x, const uint64_t *norm_g_xy,
                    size_t nlen, const uint64_t *lin_c, const uint64_t *lin_x, const uint64_t *lin_h_xy, size_t llen, bppp_nlb **out);
int bppp_basis_create_device(bppp_ctx *ctx, const void *d_points_xy, size_t n, int window_bits, size_t batch_hint, bppp_basis **out);
int bppp_msm_basis(bppp_basis *basis, const void *d_scalars, size_t n_terms, size_t batch, uint64_t *out_xy);
}
