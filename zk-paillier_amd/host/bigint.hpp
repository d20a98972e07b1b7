// bigint.hpp — minimal host-side arbitrary-precision unsigned integer for the C++ mirror of the
// reference API (zkproofs.hpp).  It only carries values to and from the C ABI and does the O(1)
// per-proof bookkeeping (n = p*q, range/3, sampling, decimal parsing); every modular
// exponentiation goes to the GPU through libzkp_hip.so.  Not a GMP replacement, not optimised.
#pragma once
#include <algorithm>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace zkproofs {

class BigInt {
 public:
  std::vector<uint32_t> l;   // little-endian limbs, no trailing zero limbs (zero = empty)

  BigInt() {}
  BigInt(uint64_t v) { while (v) { l.push_back((uint32_t)v); v >>= 32; } }
  static BigInt zero() { return BigInt(); }
  static BigInt one() { return BigInt(1); }

  void trim() { while (!l.empty() && l.back() == 0) l.pop_back(); }
  bool is_zero() const { return l.empty(); }
  bool is_odd() const { return !l.empty() && (l[0] & 1); }
  size_t bit_length() const { return l.empty() ? 0 : 32 * (l.size() - 1) + (32 - __builtin_clz(l.back())); }
  bool bit(size_t i) const { return i / 32 < l.size() && ((l[i / 32] >> (i % 32)) & 1); }

  static int cmp(const BigInt& a, const BigInt& b) {
    if (a.l.size() != b.l.size()) return a.l.size() < b.l.size() ? -1 : 1;
    for (size_t i = a.l.size(); i-- > 0;) if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
    return 0;
  }
  friend bool operator==(const BigInt& a, const BigInt& b) { return cmp(a, b) == 0; }
  friend bool operator!=(const BigInt& a, const BigInt& b) { return cmp(a, b) != 0; }
  friend bool operator<(const BigInt& a, const BigInt& b) { return cmp(a, b) < 0; }
  friend bool operator>(const BigInt& a, const BigInt& b) { return cmp(a, b) > 0; }
  friend bool operator<=(const BigInt& a, const BigInt& b) { return cmp(a, b) <= 0; }
  friend bool operator>=(const BigInt& a, const BigInt& b) { return cmp(a, b) >= 0; }

  friend BigInt operator+(const BigInt& a, const BigInt& b) {
    BigInt r; r.l.resize(std::max(a.l.size(), b.l.size()) + 1);
    uint64_t c = 0;
    for (size_t i = 0; i < r.l.size(); i++) {
      c += (uint64_t)(i < a.l.size() ? a.l[i] : 0) + (i < b.l.size() ? b.l[i] : 0);
      r.l[i] = (uint32_t)c; c >>= 32;
    }
    r.trim(); return r;
  }
  // a - b, requires a >= b
  friend BigInt operator-(const BigInt& a, const BigInt& b) {
    if (a < b) throw std::domain_error("BigInt: negative result");
    BigInt r; r.l.resize(a.l.size());
    int64_t br = 0;
    for (size_t i = 0; i < a.l.size(); i++) {
      int64_t d = (int64_t)a.l[i] - (i < b.l.size() ? b.l[i] : 0) - br;
      br = d < 0; r.l[i] = (uint32_t)d;
    }
    r.trim(); return r;
  }
  friend BigInt operator*(const BigInt& a, const BigInt& b) {
    BigInt r; if (a.is_zero() || b.is_zero()) return r;
    r.l.assign(a.l.size() + b.l.size(), 0);
    for (size_t i = 0; i < a.l.size(); i++) {
      uint64_t c = 0;
      for (size_t j = 0; j < b.l.size(); j++) { c += (uint64_t)a.l[i] * b.l[j] + r.l[i + j]; r.l[i + j] = (uint32_t)c; c >>= 32; }
      r.l[i + b.l.size()] = (uint32_t)c;
    }
    r.trim(); return r;
  }
  BigInt shl(size_t k) const {
    if (is_zero()) return *this;
    BigInt r; r.l.assign(l.size() + k / 32 + 1, 0);
    for (size_t i = 0; i < l.size(); i++) {
      uint64_t v = (uint64_t)l[i] << (k % 32);
      r.l[i + k / 32] |= (uint32_t)v; r.l[i + k / 32 + 1] |= (uint32_t)(v >> 32);
    }
    r.trim(); return r;
  }
  static BigInt pow2(size_t k) { return one().shl(k); }
  // (quotient, remainder), shift-subtract: fine for a handful of calls per proof
  static std::pair<BigInt, BigInt> divmod(const BigInt& a, const BigInt& m) {
    if (m.is_zero()) throw std::domain_error("BigInt: division by zero");
    BigInt q, r;
    q.l.assign(a.l.size(), 0);
    for (size_t i = a.bit_length(); i-- > 0;) {
      r = r.shl(1);
      if (a.bit(i)) { if (r.l.empty()) r.l.push_back(1); else r.l[0] |= 1; }
      if (r >= m) { r = r - m; q.l[i / 32] |= 1u << (i % 32); }
    }
    q.trim(); return {q, r};
  }
  friend BigInt operator%(const BigInt& a, const BigInt& m) { return divmod(a, m).second; }
  BigInt div_floor(const BigInt& d) const { return divmod(*this, d).first; }

  static BigInt gcd(BigInt a, BigInt b) { while (!b.is_zero()) { BigInt t = a % b; a = b; b = t; } return a; }
  // a^-1 mod m (throws if not invertible): extended Euclid with coefficients kept in [0, m)
  static BigInt mod_inv(const BigInt& a, const BigInt& m) {
    BigInt r0 = m, r1 = a % m, t0 = zero(), t1 = one();
    while (!r1.is_zero()) {
      auto qr = divmod(r0, r1);
      BigInt qt = (qr.first * t1) % m;
      BigInt t2 = (t0 + m - qt) % m;
      r0 = r1; r1 = qr.second; t0 = t1; t1 = t2;
    }
    if (r0 != one()) throw std::domain_error("BigInt: not invertible");
    return t0;
  }

  // ---- conversions
  static BigInt from_str_radix10(const std::string& s) {
    BigInt r;
    for (char ch : s) {
      if (ch < '0' || ch > '9') throw std::invalid_argument("BigInt: bad decimal digit");
      uint64_t c = (uint64_t)(ch - '0');
      for (auto& w : r.l) { c += (uint64_t)w * 10; w = (uint32_t)c; c >>= 32; }
      if (c) r.l.push_back((uint32_t)c);
    }
    return r;
  }
  // reference byte form: minimal big-endian, zero -> one 00 byte ([upstream] curv BigInt::to_bytes)
  std::vector<uint8_t> to_bytes() const {
    if (is_zero()) return {0};
    size_t n = (bit_length() + 7) / 8;
    std::vector<uint8_t> b(n);
    for (size_t i = 0; i < n; i++) b[n - 1 - i] = (uint8_t)(l[i / 4] >> (8 * (i % 4)));
    return b;
  }
  static BigInt from_bytes(const uint8_t* p, size_t n) {
    BigInt r; r.l.assign((n + 3) / 4, 0);
    for (size_t i = 0; i < n; i++) r.l[i / 4] |= (uint32_t)p[n - 1 - i] << (8 * (i % 4));
    r.trim(); return r;
  }
  static BigInt from_bytes(const std::vector<uint8_t>& b) { return from_bytes(b.data(), b.size()); }
  // fixed-width limb export for the C ABI; throws if the value does not fit (non-canonical operand)
  void to_limbs(uint32_t* out, size_t n) const {
    if (l.size() > n) throw std::length_error("BigInt: operand wider than the fixed ABI width");
    std::fill(out, out + n, 0u);
    std::copy(l.begin(), l.end(), out);
  }
  static BigInt from_limbs(const uint32_t* p, size_t n) { BigInt r; r.l.assign(p, p + n); r.trim(); return r; }
  std::string to_hex() const {
    if (is_zero()) return "0";
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = l.size(); i-- > 0;) for (int k = 28; k >= 0; k -= 4) s.push_back(d[(l[i] >> k) & 15]);
    return s.substr(s.find_first_not_of('0'));
  }

  // ---- sampling ([upstream] curv Samplable: OS randomness)
  static BigInt sample(size_t bits) {
    static std::random_device rd;
    BigInt r; r.l.resize((bits + 31) / 32);
    for (auto& w : r.l) w = rd();
    if (bits % 32) r.l.back() &= (1u << (bits % 32)) - 1;
    r.trim(); return r;
  }
  static BigInt sample_below(const BigInt& upper) {
    if (upper.is_zero()) throw std::domain_error("BigInt: sample_below(0)");
    for (;;) { BigInt c = sample(upper.bit_length()); if (c < upper) return c; }
  }
  static BigInt sample_range(const BigInt& lo, const BigInt& hi) { return lo + sample_below(hi - lo); }
};

}  // namespace zkproofs
