// rp_internal.hpp — declarations shared by the two halves of the range-proof layer (csrc/rp.hip: setup + batch verifier,
// csrc/rpprove.hip: batch prover).
#pragma once
#include <string>
#include <thread>
#include <vector>
#include "ctx.hpp"
#include "hostmath.hpp"
#include "rpsetup.hpp"
#include "hostpool.hpp"

struct bppp_trrp;
struct bppp_nlb;
struct bppp_basis;

namespace bppp {

struct RpDims {
  uint32_t nr, k, fn, fl;            // ranges (input commitments), rounds, final witness lengths
  uint32_t nrp, nch;                 // range-proof commitments ahead of the inputs (4: blCom rCom dmCom mCom; Binary 2: blCom dCom) and
                                     // oracle outputs before the argument's rounds (7: e x r0 q x' r1 t; Binary 4: q x r t)
  uint32_t coms_bytes, proof_bytes;  // per-proof file sizes
  uint32_t text_stride;              // bytes reserved per proof for the transcript text (multiple of 16)
};
__host__ __device__ inline uint32_t rp_npts(const RpDims &D) { return D.nrp + D.nr + 2 * D.k; }

static constexpr int RP_HDR_MAX = 64;
struct HashPlan { uint32_t hdr_be[RP_HDR_MAX / 4]; uint32_t hlen, start_pt, out_slot; };   // header as big-endian words, zero-padded
// header bytes -> big-endian words
inline void rp_pack_header(const std::string &h, uint32_t be[RP_HDR_MAX / 4]) {
  for (int i = 0; i < RP_HDR_MAX / 4; i++) be[i] = 0;
  for (size_t i = 0; i < h.size() && i < (size_t)RP_HDR_MAX; i++) be[i >> 2] |= (uint32_t)(uint8_t)h[i] << (24 - 8 * (i & 3));
}   // out_slot: index into ch[7] (< 7) or 7 + index into es[k]


// f(lo, hi) on disjoint ranges covering [0, n), one host thread each (at most 16: the GPU box's CPU share per GPU)
template <class F> static void rp_parallel(size_t n, F f) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 4);
  if (nt <= 1) { f((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++) th.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

int msm_run(bppp_ctx *, const void *, const void *, size_t, size_t, int, int, uint64_t *);
// bppp_{nl,ip}_verify_batch_device with the validation of untrusted inputs optional (csrc/nlbatch.hip)
int nl_verify_batch_run(bppp_ctx *, size_t, size_t, size_t, size_t, size_t, size_t, size_t, const void *, const void *, const void *, const void *, const void *, const void *,
                        const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, uint64_t *, bool);
int ip_verify_batch_run(bppp_ctx *, size_t, size_t, size_t, size_t, size_t, size_t, size_t, const void *, const void *, const void *, const void *, const void *, const void *,
                        const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, uint64_t *, bool);

}  // namespace bppp

extern "C" {
int bppp_trrp_create(bppp_ctx *ctx, int flavour, int has_types, size_t nlen, size_t llen, size_t nranges, const uint32_t *pos_kind, const uint32_t *pos_range,
                     const uint32_t *pos_slot, const uint32_t *pos_sym, const uint64_t *pos_coeff, const uint64_t *range_min, const uint32_t *range_assumed,
                     size_t nsyms, const uint64_t *syms, const uint32_t *cs_slot, const uint32_t *cs_sym, size_t npub, const uint32_t *pub_is_out,
                     const uint64_t *pub_amount, const uint32_t *pub_sym, bppp_trrp **out);
void bppp_trrp_destroy(bppp_trrp *t);
int bppp_trrp_public_device(bppp_trrp *t, size_t batch, const void *d_challenges, void *d_q, void *d_sp, void *d_pub_norm, void *d_pub_lin_c, void *d_init_scalars);
}


namespace bppp { struct CombTable; void comb_destroy(CombTable *); }
// Tuning knobs of one handle (bppp_rp_set_option, include/bppp.h).  The BPPP_RP_* / BPPP_NLB_* environment variables are read ONCE,
// when the handle is created, as the initial values; no entry point reads the environment per call.
struct RpOptions {
  size_t comb_min = 1024;             // BPPP_RP_COMB_MIN: first batch size (or cumulative proofs) that builds the comb table
  size_t comb_budget = (size_t)32 << 30;   // BPPP_RP_COMB_GB: table budget; also capped by a share of the free HBM (rp_ensure_comb)
  int comb_bits = 0;                  // BPPP_RP_COMB_BITS: force a window width (0 = widest that fits the budget)
  bool no_comb = false;               // BPPP_RP_NO_COMB
  size_t split_min = 4096;            // BPPP_RP_SPLIT_MIN: smallest batch run as two half-batches in flight
  size_t split_min_binary = 1024;     // BPPP_RP_SPLIT_MIN_BINARY: the same for RangeProof.Binary handles (rows of thousands of terms: fewer proofs fill the chip)
  bool no_split = false;              // BPPP_RP_NO_SPLIT
  size_t host_oracle_verify = 8;      // BPPP_RP_HOST_ORACLE_MAX: largest batch whose transcript hashing runs on the host (verifier)
  size_t host_oracle_prove = 64;      //                          ... (prover)
  size_t hash_fork_max = 64;          // BPPP_RP_HASH_FORK_MAX: largest batch whose two hashing halves run side by side on two streams (verifier)
  bool fold_points = false;           // BPPP_NLB_FOLD_POINTS: point-folding argument although a table exists
  bool host_algebra = false;          // BPPP_RP_HOST_ALGEBRA: field algebra and hashing of the prover on the host
  bool timing = false;                // BPPP_RP_TIMING: phase times on stderr
  void from_env();
};
struct bppp_brp_tabs;      // csrc/rp.hip: device tables of a RangeProof.Binary setup
struct bppp_rp {
  bppp_ctx *ctx = nullptr;
  RpOptions opt;
  bppp_brp_tabs *btabs = nullptr;
  bppp_rps::Setup st;
  bppp_trrp *tabs = nullptr;
  std::string tag;
  std::vector<uint64_t> h_g, h_G, h_H;          // the basis on the host (prover: commit inputs, argument start)
  // the registered basis, affine, resident in HBM in commitRPW's term order (src/RangeProof/Internal.hs:45-50):
  // [g (1) | H (llen) | G (nlen)]
  uint32_t *d_basis = nullptr;
  const uint32_t *d_g() const { return d_basis; }
  const uint32_t *d_H() const { return d_basis + 16; }
  const uint32_t *d_G() const { return d_basis + 16 * (1 + st.llen); }
  bppp::HashPlan *d_plan = nullptr;
  uint32_t nhash = 0;
  bppp::RpDims D{};
  // prover side (csrc/rpprove.hip): fixed-base window table of g, H[0], H[1] for the input commitments, prover workspace
  uint32_t *d_fixed = nullptr;
  bppp_basis *commit_basis = nullptr;           // [g | H | G] registered with its fixed-base table: the range-proof commitments
  void *pwork = nullptr; size_t pwork_bytes = 0;
  void *awork = nullptr; size_t awork_bytes = 0;   // grow-only workspace of the device-resident inner-product argument (csrc/ipb.hip)
  void *hpin = nullptr; size_t hpin_bytes = 0;     // pinned host staging of the batch prover
  // the creation arguments, kept so that a second handle on its OWN context (stream, workspaces) can be made: a large prove batch
  // runs as two half-batches in flight, the host shares of one under the kernels of the other (csrc/rpprove.hip)
  std::vector<bppp_rp_range> c_ranges; std::vector<bppp_rp_public> c_pubs; std::vector<uint64_t> c_points; int c_has_types = 0;
  int c_conserve = 0; uint64_t c_net_public[4] = {0, 0, 0, 0};      // bppp_rp_create_binary's arguments, for the twin handle
  bppp_rp *twin = nullptr; bppp_ctx *twin_ctx = nullptr; bool is_twin = false;
  // fixed-base comb over [g | H | G] (csrc/comb.hip): the range-proof commitments and the argument's round commitments of large
  // batches; the twin handle uses its parent's table
  bppp::CombTable *comb = nullptr; bool comb_owned = false, comb_failed = false; size_t proved_total = 0;
  uint32_t *d_comb_out = nullptr; size_t comb_out_rows = 0;   // fixed-base tables of the argument's first round (csrc/nlb.hip)
  // grow-only verifier workspace and the staging buffer of the host-buffer entry point
  bppp::HostPool *pool = nullptr;                // workers of the host oracle (batches of 2 .. host_oracle_verify proofs), made on first use
  hipEvent_t slice_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // one per upload slice of bppp_rp_verify_batch
  const uint8_t *host_coms = nullptr, *host_proofs = nullptr;   // set by bppp_rp_verify_batch around its call of the device entry point: files still on the host
  uint64_t *hstage = nullptr; size_t hstage_bytes = 0;   // pinned, grow-only: the host oracle's downloads and uploads (a pageable target makes every async copy a blocking one)
  uint32_t *hflag = nullptr;                     // pinned: the verifier's "some proof did not decode" word, copied out while the batch is still in flight
  void *work = nullptr; size_t work_bytes = 0;
  void *stage = nullptr; size_t stage_bytes = 0;
};

int rp_ensure_twin(bppp_rp *rp);      // csrc/rp.hip
int rp_ensure_comb(bppp_rp *rp);      // csrc/rpprove.hip
