// rpsetup.hpp — host-side structure of a typed reciprocal range-proof setup (RangeProof.TypedReciprocal of the reference).
//
// What `setup` fixes once per schema (src/RangeProof/TypedReciprocal.hs:332-359): the ranges and their digit coefficients
// (makeRangeData :103-120), which bases are shared, the norm / linear vector lengths, the number of argument rounds
// (optimalWitnessSize, src/Bulletproof/NormArgument.hs:165-178) and the Phase1 layout of the norm vector
// (makePhase1s / makePhase1sVer :133-169).  Used by csrc/rp.hip for both halves: the batch verifier uploads the layout as the
// tables of csrc/trrp.hip, the batch prover walks it with the private digits.  Plain integers are 256-bit (amounts, range bounds);
// everything that is a field element is reduced mod n where the reference's `fromInteger` would.
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "hostmath.hpp"

namespace bppp_rps {
using bppp_host::U256;
using bppp_host::u128;

// ---- small unsigned 256-bit integer helpers (range arithmetic is on integers, not field elements)
inline U256 u_add(const U256 &a, const U256 &b) { U256 r; bppp_host::add_raw(r, a, b); return r; }
inline U256 u_sub(const U256 &a, const U256 &b) { U256 r; bppp_host::sub_raw(r, a, b); return r; }
inline bool u_lt(const U256 &a, const U256 &b) { return bppp_host::cmp(a, b) < 0; }
inline U256 u_mul64(const U256 &a, uint64_t m, bool *overflow = nullptr) {
  U256 r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.w[i] * m; r.w[i] = (uint64_t)c; c >>= 64; }
  if (overflow && (uint64_t)c) *overflow = true;
  return r;
}
inline U256 u_div64(const U256 &a, uint64_t d, uint64_t *rem = nullptr) {
  U256 q; u128 r = 0;
  for (int i = 3; i >= 0; i--) { r = (r << 64) | a.w[i]; q.w[i] = (uint64_t)(r / d); r %= d; }
  if (rem) *rem = (uint64_t)r;
  return q;
}
// Range bounds and amounts are INTEGERS that may be negative (examples/rec_test has "min": -20; the reference's Integer): two's complement
// over 256 bits, |x| < 2^255.  Differences (widths, value - min) come out right from the wrapping subtraction; order and the image in
// the scalar field need the sign.
inline bool s_neg(const U256 &a) { return (a.w[3] >> 63) != 0; }
inline bool s_lt(const U256 &a, const U256 &b) { return s_neg(a) != s_neg(b) ? s_neg(a) : bppp_host::cmp(a, b) < 0; }
inline U256 u_mod_n(const U256 &a);
inline U256 s_mod_n(const U256 &a) {      // fromInteger into the scalar field for a signed 256-bit integer
  if (!s_neg(a)) return u_mod_n(a);
  U256 m = u_mod_n(u_sub(U256::zero(), a)), r;
  if (m.is_zero()) return m;
  bppp_host::sub_raw(r, bppp_host::FR().m, m);
  return r;
}
inline U256 u_mod_n(const U256 &a) {      // fromInteger into the scalar field for a < 2^256 < 2n
  U256 r = a;
  if (bppp_host::cmp(r, bppp_host::FR().m) >= 0) bppp_host::sub_raw(r, r, bppp_host::FR().m);
  return r;
}

// ---- RangeData (TypedReciprocal.hs:85-101) and makeRangeData (:103-120)
struct RangeData {
  uint32_t base = 0;
  U256 lo, hi;
  bool shared = false, output = false, assumed = false, has_bit = false;
  std::vector<U256> coeffs;        // baseCoeffs as INTEGERS (empty when assumed)
};

// integerLog (src/Utils.hs:78-79): number of times n can be divided by b before it drops below b
inline int integer_log(uint64_t b, U256 n) {
  int r = 0;
  const U256 bb = U256::from_u64(b);
  while (!u_lt(n, bb)) { n = u_div64(n, b); r++; }
  return r;
}

inline bool make_range_data(uint32_t base, const U256 &lo, const U256 &hi, bool shared, bool output, bool assumed, RangeData &out, std::string &err) {
  if (!(s_lt(lo, hi)) || base < 2) { err = "invalid range (need max > min and base > 1)"; return false; }
  const U256 w = u_sub(hi, lo);
  if (bppp_host::cmp(w, bppp_host::FR().m) >= 0) { err = "range wider than the scalar field"; return false; }
  const uint64_t b = base;
  const U256 wm1 = u_sub(w, U256::one());
  const int n1 = integer_log(b, wm1);
  uint64_t rem = 0;
  u_div64(wm1, b - 1, &rem);
  const bool has_bit = rem != 0;
  std::vector<U256> pw(n1 + 1);                      // b^0 .. b^n1  (b^n1 <= w - 1: no overflow)
  pw[0] = U256::one();
  for (int i = 1; i <= n1; i++) pw[i] = u_mul64(pw[i - 1], b);
  std::vector<U256> bs;
  const U256 top = u_sub(w, pw[n1]);                 // w - b^n1 >= 1
  if (!has_bit) bs.push_back(u_div64(top, b - 1));
  else if (u_lt(top, pw[n1])) bs.push_back(top);     // w < 2 b^n1
  else {
    // bn1 = 1 + w div (2 (b - 1)) - (b^n1 - 1) div (b - 1)
    U256 bn1 = u_sub(u_add(U256::one(), u_div64(w, 2 * (b - 1))), u_div64(u_sub(pw[n1], U256::one()), b - 1));
    bool ovf = false;
    U256 first = u_sub(u_sub(w, u_mul64(bn1, b - 1, &ovf)), pw[n1]);
    if (ovf) { err = "range arithmetic overflow"; return false; }
    bs.push_back(first); bs.push_back(bn1);
  }
  for (int i = 1; i <= n1; i++) bs.push_back(pw[n1 - i]);
  for (const U256 &c : bs) if (c.is_zero()) { err = "degenerate range: a digit coefficient is zero"; return false; }
  out.base = base; out.lo = lo; out.hi = hi; out.shared = shared; out.output = output; out.assumed = assumed; out.has_bit = has_bit;
  out.coeffs = assumed ? std::vector<U256>() : bs;
  return true;
}

// digits (TypedReciprocal.hs:125-127): greedy mixed-radix digits of n (already shifted by the range minimum); the first digit
// is binary when has_bit.  d = min (radix - 1) (n div coeff): the largest d <= radix - 1 with d * coeff <= n, by bisection.
// Each digit is the largest d <= radix - 1 with d * coeff <= n (the reference subtracts repeatedly; same digits).  Amounts are
// nearly always below 2^64, where that is one machine division; wider remainders bisect with a 256 x 64-bit product.
inline void digits_into(const RangeData &rd, U256 n, std::vector<uint32_t> &out) {
  out.clear();
  for (size_t i = 0; i < rd.coeffs.size(); i++) {
    const uint32_t radix = (rd.has_bit && i == 0) ? 2u : rd.base;
    const U256 &cf = rd.coeffs[i];
    if (!(n.w[1] | n.w[2] | n.w[3])) {
      uint32_t d;
      if (cf.w[1] | cf.w[2] | cf.w[3]) d = 0;
      else if (cf.w[0] == 0) d = radix - 1;
      else { const uint64_t q = n.w[0] / cf.w[0]; d = q < radix - 1 ? (uint32_t)q : radix - 1; }
      n.w[0] -= (uint64_t)d * cf.w[0];
      out.push_back(d);
      continue;
    }
    uint32_t lo = 0, hi = radix - 1;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      bool ovf = false;
      const U256 prod = u_mul64(cf, mid, &ovf);
      if (!ovf && !u_lt(n, prod)) lo = mid; else hi = mid - 1;
    }
    if (lo) n = u_sub(n, u_mul64(cf, lo));
    out.push_back(lo);
  }
}
inline std::vector<uint32_t> digits(const RangeData &rd, const U256 &n) {
  std::vector<uint32_t> out;
  digits_into(rd, n, out);
  return out;
}

// ---- round counts (src/Bulletproof.hs:300-304; NormArgument.hs:165-178), NL flavour
inline size_t round_reduce(size_t n) { return n / 2 + n % 2; }
inline void number_rounds_reduce(size_t n, size_t &rounds, size_t &fin) {
  rounds = 0;
  while (n >= 5) { n = round_reduce(n); rounds++; }
  fin = n;
}
inline void optimal_witness_size_nl(size_t n_len, size_t l_len, size_t &rounds, size_t &fn, size_t &fl) {
  size_t nR, n1, lR, l1;
  number_rounds_reduce(n_len, nR, n1);
  number_rounds_reduce(l_len, lR, l1);
  size_t r = std::max(nR, lR);
  for (size_t i = nR; i < r; i++) n1 = round_reduce(n1);
  for (size_t i = lR; i < r; i++) l1 = round_reduce(l1);
  if (n1 + l1 > 5) { rounds = r + 1; fn = round_reduce(n1); fl = round_reduce(l1); }
  else { rounds = r; fn = n1; fl = l1; }
}

// IP flavour (src/Bulletproof/InnerProductArgument.hs:253-267): the norm vector is paired up first (nLen padded to even, halved) and reduced
// with numberRoundsReduce' (src/Bulletproof.hs:307-308: one more round while more than 2 pairs are left); fn counts SCALARS (2 per pair)
inline void optimal_witness_size_ip(size_t n_len, size_t l_len, size_t &rounds, size_t &fn, size_t &fl) {
  size_t nR, n1, lR, l1;
  number_rounds_reduce((n_len + (n_len % 2)) / 2, nR, n1);
  if (n1 > 2) { nR++; n1 = round_reduce(n1); }
  number_rounds_reduce(l_len, lR, l1);
  size_t r = std::max(nR, lR);
  for (size_t i = nR; i < r; i++) n1 = round_reduce(n1);
  for (size_t i = lR; i < r; i++) l1 = round_reduce(l1);
  if (2 * n1 + l1 > 5) { rounds = r + 1; fn = 2 * round_reduce(n1); fl = round_reduce(l1); }
  else { rounds = r; fn = 2 * n1; fl = l1; }
}

// ---- Phase1 layout of the norm vector (one record per position; private fields live with the witness)
enum : uint32_t { POS_TYPING = 0, POS_INLINE = 1, POS_SHARED = 2, POS_F_IO = 1u << 8, POS_F_IA = 1u << 9, POS_NO_SYM = 0xFFFFFFFFu };
struct Pos {
  uint32_t kind;        // POS_* | flags
  uint32_t range;       // index of the range
  uint32_t radix;       // the digit's base (2 for the leading bit), 0 for typing
  uint32_t digit_index; // which digit of the range (inline / shared)
  U256 coeff;           // baseCoeff mod n (0 beyond the coefficient list)
  uint32_t sym_small;   // inline: the symbol s of this slot (1 for the bit, 1..b-1 otherwise), 0 = none
};

struct PublicVT { bool is_output; U256 type, amount; };   // (isOutput, type, amount), field elements

struct Setup {
  int flavour = 0;                      // 0 = Bulletproof.NormArgument (NL), 1 = Bulletproof.InnerProductArgument (IP)
  int kind = 0;                         // 0 = RangeProof.TypedReciprocal, 1 = RangeProof.Binary (make_setup_binary below)
  bool conserve = false;                // Binary: the amounts (and the net public amount) must balance (Binary.hs:137)
  U256 net_public;                      // Binary: netPublic as an element of the scalar field
  size_t nlive = 0;                     // Binary: norm positions of the ranges that are not assumed (the rest of nlen stays zero)
  bool has_types = false;
  std::vector<RangeData> rds;
  std::vector<PublicVT> pubs;
  std::vector<uint32_t> m_bases, sorted_bases;
  size_t nlen = 0, llen = 0, rounds = 0, fn = 0, fl = 0;
  std::vector<Pos> pos;                 // nlen records, types first when has_types
  std::vector<size_t> first_pos;        // per range: index in `pos` of its first digit record
  int slot_of(uint32_t base) const {
    auto it = std::lower_bound(sorted_bases.begin(), sorted_bases.end(), base);
    return (it != sorted_bases.end() && *it == base) ? (int)(it - sorted_bases.begin()) : -1;
  }
};

// setup (TypedReciprocal.hs:332-359) + the verifier's Phase1 list (makePhase1sVer :163-169)
inline bool make_setup(bool has_types, const std::vector<RangeData> &rds, const std::vector<PublicVT> &pubs, Setup &st, std::string &err, int flavour = 0) {
  st = Setup();
  st.flavour = flavour;
  st.has_types = has_types; st.rds = rds; st.pubs = pubs;
  bool any_has_bit = false, any_shared_has_bit = false;
  std::vector<uint32_t> mb, sb;
  for (const RangeData &rd : rds) {
    if (rd.assumed) continue;
    any_has_bit |= rd.has_bit; any_shared_has_bit |= rd.has_bit && rd.shared;
    sb.push_back(rd.base);
    if (rd.shared) mb.push_back(rd.base);
  }
  if (any_shared_has_bit) mb.push_back(2);
  if (any_has_bit) sb.push_back(2);
  std::sort(mb.begin(), mb.end()); mb.erase(std::unique(mb.begin(), mb.end()), mb.end());
  std::sort(sb.begin(), sb.end()); sb.erase(std::unique(sb.begin(), sb.end()), sb.end());
  st.m_bases = mb; st.sorted_bases = sb;
  st.nlen = 0;
  for (const RangeData &rd : rds) st.nlen += rd.coeffs.size() + (has_types ? 1 : 0);
  st.llen = 6;
  for (uint32_t b : mb) st.llen += b - 1;
  if (!st.nlen) { err = "empty norm vector"; return false; }
  // untrusted sizes: a base of 2^32 - 1 asks for four billion linear entries (shared) or reciprocal symbols (inline) — refuse before anything
  // is allocated (the argument's kernels index vectors below 2^24 entries, csrc/nlb.hip)
  if (st.nlen >= ((size_t)1 << 24) || st.llen >= ((size_t)1 << 24)) { err = "setup too large: norm / linear vector of 2^24 entries or more"; return false; }
  // An inline range whose reciprocal symbols ([1 | has_bit] ++ [1 .. base - 1]) outnumber its digits: makePhase1s pads bs, ds, ms, ns to the
  // LONGEST (TypedReciprocal.hs:150-153) while setup counts one norm position per digit (:346), so the reference's own norm vector is longer
  // than its basis gs — the extra entries are committed to the identity (dotWith's padding, Commitment.hs:423-424), bound by nothing, and the
  // final witness no longer has the length optimalWitnessSize nrmLen promises decodeProof'.  Not a layout a sound proof can use: refused.
  for (const RangeData &rd : rds)
    if (!rd.assumed && !rd.shared && (rd.has_bit ? 1 : 0) + (size_t)(rd.base - 1) > rd.coeffs.size()) {
      err = "an inline range has more reciprocal symbols than digits (base - 1 > number of digits): unsupported layout"; return false;
    }
  if (flavour) optimal_witness_size_ip(st.nlen, st.llen, st.rounds, st.fn, st.fl);
  else optimal_witness_size_nl(st.nlen, st.llen, st.rounds, st.fn, st.fl);
  // Phase1 records
  if (has_types)
    for (size_t i = 0; i < rds.size(); i++)
      st.pos.push_back(Pos{POS_TYPING | (rds[i].output ? POS_F_IO : 0u) | (rds[i].assumed ? POS_F_IA : 0u), (uint32_t)i, 0, 0, U256::zero(), 0});
  st.first_pos.assign(rds.size(), 0);
  for (size_t i = 0; i < rds.size(); i++) {
    const RangeData &rd = rds[i];
    st.first_pos[i] = st.pos.size();
    if (rd.assumed) continue;
    const size_t nb = rd.coeffs.size();
    if (rd.shared) {
      for (size_t j = 0; j < nb; j++)
        st.pos.push_back(Pos{POS_SHARED, (uint32_t)i, (rd.has_bit && j == 0) ? 2u : rd.base, (uint32_t)j, u_mod_n(rd.coeffs[j]), 0});
    } else {
      // inline: zipWith over bs, ds, ms, ns padded to the longest (:150-153); ns = [1 | has_bit] ++ [1 .. b-1]
      const size_t nsym = (rd.has_bit ? 1 : 0) + (rd.base - 1);
      const size_t ln = std::max(nb, nsym);
      for (size_t j = 0; j < ln; j++) {
        uint32_t s = 0;
        if (j < nsym) s = rd.has_bit ? (j == 0 ? 1u : (uint32_t)j) : (uint32_t)(j + 1);
        st.pos.push_back(Pos{POS_INLINE, (uint32_t)i, (rd.has_bit && j == 0) ? 2u : rd.base, (uint32_t)j, j < nb ? u_mod_n(rd.coeffs[j]) : U256::zero(), s});
      }
    }
  }
  if (st.pos.size() != st.nlen) { err = "internal: Phase1 layout and norm length disagree"; return false; }
  for (const Pos &p : st.pos)
    if ((p.kind & 0xFFu) != POS_TYPING && st.slot_of(p.radix) < 0) { err = "internal: digit base missing from the base map"; return false; }
  return true;
}

// ---- RangeProof.Binary (src/RangeProof/Binary.hs)
// makeRangeData (:48-54): coefficients b_n : 2^(n1-1) ... 1 with n1 = integerLog 2 (max - min - 1), b_n = (max - min) - 2^n1
inline bool make_range_data_binary(const U256 &lo, const U256 &hi, bool output, bool assumed, RangeData &out, std::string &err) {
  if (!s_lt(lo, hi)) { err = "invalid range (need max > min)"; return false; }
  const U256 w = u_sub(hi, lo);
  if (bppp_host::cmp(w, bppp_host::FR().m) >= 0) { err = "range wider than the scalar field"; return false; }
  const int n1 = integer_log(2, u_sub(w, U256::one()));
  U256 p2 = U256::one();
  for (int i = 0; i < n1; i++) p2 = u_add(p2, p2);                    // 2^n1 <= w - 1
  out = RangeData();
  out.base = 2; out.lo = lo; out.hi = hi; out.output = output; out.assumed = assumed;
  out.coeffs.push_back(u_sub(w, p2));
  for (int i = 1; i <= n1; i++) { U256 c = U256::one(); for (int k = 0; k < n1 - i; k++) c = u_add(c, c); out.coeffs.push_back(c); }
  return true;
}
// makeDigits (:56-69): the top digit takes b_n, the rest are the n1 bits of what is left.  The reference takes the top coefficient only
// when nAdj > b_n; for a power-of-two width (b_n = 2^n1) and nAdj = b_n its low part would need n1 + 1 bits — that one value takes the
// top digit here (the deviation bulletproofspp_amd/rangeproof_binary.py documents).  n = the value minus the range minimum.
inline void digits_binary_into(const RangeData &rd, const U256 &n, std::vector<uint32_t> &out) {
  out.clear();
  if (rd.assumed) return;
  const int n1 = (int)rd.coeffs.size() - 1;
  const U256 &bn = rd.coeffs[0];
  bool over = false;                                                   // n >> n1 != 0
  for (int b = n1; b < 256; b++) over |= n.bit(b);
  const bool top = u_lt(bn, n) || over;
  const U256 rest = top ? u_sub(n, bn) : n;
  out.push_back(top ? 1u : 0u);
  for (int i = 0; i < n1; i++) out.push_back(rest.bit(n1 - 1 - i) ? 1u : 0u);
}

// setupBRP (:143-156): nrmLen = sum of ALL ranges' coefficient counts (assumed ones included: their positions stay zero), linLen = 2
// (the blinding generators h0, h1), rounds = optimalWitnessSize nrmLen 2 on both sides (the reference's prover uses integerLog 2 nrmLen - 1,
// which agrees wherever its own proofs verify: SURVEY.md App. D-1).  `pos` holds one record per LIVE position, in order.
inline bool make_setup_binary(bool conserve, const std::vector<RangeData> &rds, const U256 &net_public_mod_n, int flavour, Setup &st, std::string &err) {
  st = Setup();
  st.kind = 1; st.flavour = flavour; st.conserve = conserve; st.net_public = net_public_mod_n; st.rds = rds;
  st.nlen = 0; st.llen = 2;
  st.first_pos.assign(rds.size(), 0);
  for (size_t i = 0; i < rds.size(); i++) {
    st.nlen += rds[i].coeffs.size();
    st.first_pos[i] = st.pos.size();
    if (rds[i].assumed) continue;
    for (size_t j = 0; j < rds[i].coeffs.size(); j++) st.pos.push_back(Pos{POS_SHARED, (uint32_t)i, 2u, (uint32_t)j, u_mod_n(rds[i].coeffs[j]), 0});
  }
  st.nlive = st.pos.size();
  if (!st.nlen) { err = "empty norm vector"; return false; }
  if (st.nlen >= ((size_t)1 << 24)) { err = "setup too large: norm vector of 2^24 entries or more"; return false; }
  if (flavour) optimal_witness_size_ip(st.nlen, st.llen, st.rounds, st.fn, st.fl);
  else optimal_witness_size_nl(st.nlen, st.llen, st.rounds, st.fn, st.fl);
  return true;
}

}  // namespace bppp_rps
