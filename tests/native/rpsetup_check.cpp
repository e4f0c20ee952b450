// The host-side setup code that parses UNTRUSTED range descriptions (csrc/rpsetup.hpp: make_range_data, make_setup, make_range_data_binary,
// make_setup_binary, digits_into, digits_binary_into, the round-count rules) under AddressSanitizer / UndefinedBehaviorSanitizer:
//   * every range list of the reference's examples (given on stdin by tests/test_host_sanitizers.py: one line per schema) must set up, with the
//     shapes of SURVEY.md App. B when the caller passes them;
//   * a few hundred mutated ranges — min >= max, base 0 / 1 / 2 / 2^32 - 1, negative minima, the whole 256-bit width, zero ranges, hundreds of
//     ranges — must be either accepted or refused with a message, never crash, overflow a buffer or hit undefined behaviour;
//   * accepted ranges: the digits of the minimum, the maximum - 1 and random members recombine to the value (sum d_i coeff_i = value - min).
// stdin: lines  "<name> <typed> <flavour> <nlen> <llen> <rounds> <nranges> { <base> <min> <max> <shared> <output> <assumed> }*"  (decimal, min may be negative)
#include <cstdio>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include "rpsetup.hpp"
using namespace bppp_rps;
using bppp_host::U256;

static U256 parse_dec(const std::string &s) {          // signed decimal -> 256-bit two's complement
  bool neg = !s.empty() && s[0] == '-';
  U256 v = U256::zero();
  for (size_t i = neg ? 1 : 0; i < s.size(); i++) v = u_add(u_mul64(v, 10), U256::from_u64((uint64_t)(s[i] - '0')));
  return neg ? u_sub(U256::zero(), v) : v;
}
static int bad = 0;
#define CHECK(c, ...) do { if (!(c)) { if (bad < 10) { printf(__VA_ARGS__); printf("\n"); } bad++; } } while (0)

static void check_digits(const RangeData &rd, const U256 &value_minus_min, bool binary) {
  std::vector<uint32_t> ds;
  if (binary) digits_binary_into(rd, value_minus_min, ds); else digits_into(rd, value_minus_min, ds);
  if (rd.assumed) { CHECK(ds.empty() || !binary, "assumed range has digits"); return; }
  CHECK(ds.size() == rd.coeffs.size(), "digit count %zu != coefficient count %zu", ds.size(), rd.coeffs.size());
  U256 acc = U256::zero();
  for (size_t i = 0; i < ds.size() && i < rd.coeffs.size(); i++) {
    const uint32_t radix = binary ? 2u : ((rd.has_bit && i == 0) ? 2u : rd.base);
    CHECK(ds[i] < radix, "digit %u out of its radix %u", ds[i], radix);
    acc = u_add(acc, u_mul64(rd.coeffs[i], ds[i]));
  }
  CHECK(bppp_host::cmp(acc, value_minus_min) == 0, "digits do not recombine (base %u)", rd.base);
}

int main() {
  std::mt19937_64 g(2024);
  std::string line;
  int nschemas = 0;
  while (std::getline(std::cin, line)) {
    if (line.empty()) continue;
    std::istringstream in(line);
    std::string name; int typed, flavour; size_t nlen, llen, rounds, nr;
    in >> name >> typed >> flavour >> nlen >> llen >> rounds >> nr;
    std::vector<RangeData> rds(nr), brds;
    std::string err;
    bool ok = true;
    for (size_t i = 0; i < nr; i++) {
      uint32_t base; std::string lo, hi; int sh, out, as;
      in >> base >> lo >> hi >> sh >> out >> as;
      ok = ok && make_range_data(base, parse_dec(lo), parse_dec(hi), sh != 0, out != 0, as != 0, rds[i], err);
      RangeData b;
      if (make_range_data_binary(parse_dec(lo), parse_dec(hi), out != 0, as != 0, b, err)) brds.push_back(b);
    }
    CHECK(ok, "%s: a range of a reference example was refused: %s", name.c_str(), err.c_str());
    if (!ok) continue;
    Setup st;
    std::vector<PublicVT> pubs;
    if (typed) pubs.push_back(PublicVT{false, U256::from_u64(15), U256::from_u64(1)});
    if (name != "bin_test") {
      CHECK(make_setup(typed != 0, rds, pubs, st, err, flavour), "%s: make_setup refused: %s", name.c_str(), err.c_str());
      if (nlen) CHECK(st.nlen == nlen && st.llen == llen && st.rounds == rounds, "%s: shape (%zu, %zu, %zu), expected (%zu, %zu, %zu)", name.c_str(), st.nlen, st.llen, st.rounds, nlen, llen, rounds);
      for (const RangeData &rd : rds) {
        const U256 w = u_sub(rd.hi, rd.lo);
        check_digits(rd, U256::zero(), false); check_digits(rd, u_sub(w, U256::one()), false);
        for (int k = 0; k < 50; k++) { U256 v; for (int q = 0; q < 4; q++) v.w[q] = g(); while (!u_lt(v, w)) { for (int q = 3; q >= 0; q--) if (v.w[q]) { v.w[q] >>= 1; break; } } check_digits(rd, v, false); }
      }
    }
    if (brds.size() == nr) {
      Setup sb;
      CHECK(make_setup_binary(true, brds, U256::zero(), flavour, sb, err), "%s: make_setup_binary refused: %s", name.c_str(), err.c_str());
      if (name == "bin_test" && nlen) CHECK(sb.nlen == nlen && sb.rounds == rounds, "bin_test: shape (%zu, %zu), expected (%zu, %zu)", sb.nlen, sb.rounds, nlen, rounds);
      for (const RangeData &rd : brds) {
        const U256 w = u_sub(rd.hi, rd.lo);
        check_digits(rd, U256::zero(), true); check_digits(rd, u_sub(w, U256::one()), true);
        for (int k = 0; k < 50; k++) { U256 v; for (int q = 0; q < 4; q++) v.w[q] = g(); while (!u_lt(v, w)) { for (int q = 3; q >= 0; q--) if (v.w[q]) { v.w[q] >>= 1; break; } } check_digits(rd, v, true); }
      }
    }
    nschemas++;
  }
  // ---- mutated ranges: accepted or refused, never a crash
  const uint32_t bases[] = {0u, 1u, 2u, 3u, 4u, 7u, 16u, 255u, 256u, 257u, 65536u, 0x7FFFFFFFu, 0xFFFFFFFFu};
  int accepted = 0, refused = 0;
  for (int it = 0; it < 600; it++) {
    U256 lo, hi;
    for (int q = 0; q < 4; q++) { lo.w[q] = g(); hi.w[q] = g(); }
    switch (it % 10) {
      case 0: hi = lo; break;                                                       // empty
      case 1: lo = U256::zero(); hi = U256::one(); break;                          // one value
      case 2: lo = u_sub(U256::zero(), U256::from_u64(g() % 1000)); hi = U256::from_u64(1 + g() % 100000); break;   // negative minimum
      case 3: lo = U256::zero(); for (int q = 0; q < 4; q++) hi.w[q] = ~0ull; hi.w[3] >>= 1; break;                  // 2^255 - 1 wide
      case 4: lo.w[3] |= 1ull << 63; hi.w[3] &= ~(1ull << 63); break;              // negative to positive, huge
      case 5: lo = U256::from_u64(g() % 50); hi = u_add(lo, U256::from_u64(1 + g() % 70000)); break;
      case 6: lo = U256::zero(); hi = U256::zero(); hi.w[it % 4] = 1ull << (g() % 64); break;       // a power of two
      case 7: std::swap(lo, hi); break;
      case 8: lo = U256::zero(); hi = bppp_host::FR().m; break;                    // exactly the field order wide
      default: lo.w[3] = hi.w[3] = 0; lo.w[2] = hi.w[2] = 0; break;
    }
    const uint32_t base = bases[(it / 10) % (sizeof bases / sizeof bases[0])];
    RangeData rd; std::string err;
    for (int flags = 0; flags < 4; flags++) {
      const bool shared = flags & 1, assumed = flags & 2;
      if (make_range_data(base, lo, hi, shared, true, assumed, rd, err)) {
        accepted++;
        const U256 w = u_sub(hi, lo);
        if (!assumed && rd.coeffs.size() < 300 && base <= 65536) { check_digits(rd, U256::zero(), false); check_digits(rd, u_sub(w, U256::one()), false); }
        Setup st; std::vector<RangeData> one(1 + it % 3, rd);
        if (!make_setup(it & 1, one, std::vector<PublicVT>(), st, err, (it >> 1) & 1)) refused++;
        else CHECK(st.pos.size() == st.nlen && st.rounds < 64, "accepted setup inconsistent");
      } else { refused++; CHECK(!err.empty(), "refused without a message"); }
    }
    RangeData b; std::string e2;
    if (make_range_data_binary(lo, hi, true, it & 1, b, e2)) {
      accepted++;
      if (!(it & 1)) { const U256 w = u_sub(hi, lo); check_digits(b, U256::zero(), true); check_digits(b, u_sub(w, U256::one()), true); }
      Setup sb; std::vector<RangeData> many(1 + it % 5, b);
      CHECK(make_setup_binary(it & 2, many, U256::from_u64(g()), it & 1, sb, e2) || !e2.empty(), "binary setup refused without a message");
    } else refused++;
  }
  { Setup st; std::string err; CHECK(!make_setup(false, std::vector<RangeData>(), std::vector<PublicVT>(), st, err, 0) && !err.empty(), "an empty range list set up"); }
  { Setup st; std::string err; CHECK(!make_setup_binary(true, std::vector<RangeData>(), U256::zero(), 0, st, err) && !err.empty(), "an empty binary range list set up"); }
  // the inline layout the reference mis-sizes (TypedReciprocal.hs:145-153 pads to base - 1 symbols, :346 counts digits only): refused with a message
  { RangeData rd; std::string err; Setup st;
    CHECK(make_range_data(16, U256::zero(), U256::from_u64(100), false, true, false, rd, err) && rd.coeffs.size() == 3, "range 16 [0, 100)");
    CHECK(!make_setup(false, std::vector<RangeData>(1, rd), std::vector<PublicVT>(), st, err, 0) && err.find("unsupported layout") != std::string::npos, "padded inline layout accepted"); }
  // round-count rules on every length up to 5000 (no overflow, final lengths as the rules promise)
  for (size_t n = 1; n <= 5000; n++) {
    size_t r, fn, fl;
    optimal_witness_size_nl(n, 1 + n % 300, r, fn, fl); CHECK(fn + fl <= 5 && fn >= 1 && r < 20, "nl rounds at %zu", n);
    optimal_witness_size_ip(n, 1 + n % 300, r, fn, fl); CHECK(fn + fl <= 5 && r < 20, "ip rounds at %zu", n);
  }
  printf("schemas %d, mutated accepted %d refused %d: %s\n", nschemas, accepted, refused, bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
