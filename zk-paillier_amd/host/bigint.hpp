// bigint.hpp — host-side arbitrary-precision SIGNED integer for the C++ mirror of the reference API (zkproofs.hpp).
// It carries values to and from the C ABI and does the per-proof bookkeeping the reference does in curv::BigInt around its
// modular exponentiations (n = p*q, range/3, sampling, comparisons, the `%` of over-wide or negative fields of a received proof);
// every modular exponentiation goes to the GPU through libzkp_hip.so.  Operator semantics follow curv::BigInt over GMP, because a
// deserialised proof may hold any integer:
//   a % m          truncated remainder, sign of the dividend (Rust `%`: mpz_tdiv_r)       [upstream, recalled]
//   modulus(m)     floored, result in [0, |m|) (BigInt::modulus / mod_mul: mpz_mod)        [upstream, recalled]
//   div_floor(d)   floored quotient (mpz_fdiv_q)
//   to_bytes()     big-endian MAGNITUDE, zero -> one 00 byte (mpz_export ignores the sign) [upstream, recalled]
// Division is Knuth's algorithm D on 32-bit limbs; sampling draws from a per-thread ChaCha20 stream keyed from the OS.
#pragma once
#include <sys/random.h>
#include <sys/types.h>
#include <unistd.h>

#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace zkproofs {

namespace detail {
// ChaCha20 block function (RFC 8439) as a DRBG: the reference samples through rand's OsRng / thread_rng; one OS read per 32-bit word
// costs more than the GPU step at 4096 proofs x 128 rows x 136 words.  Key and nonce come from the kernel (getrandom(2); it blocks until
// the pool is initialised and cannot return a deterministic stream the way std::random_device may); the stream is re-keyed
//   * in a child after fork(): a forked process must not replay the witness randomness and nonces of its parent — a pthread_atfork
//     handler bumps a process-wide generation number in the child, and every word drawn compares it with the generator's own (one
//     relaxed load; the words still buffered at the fork are dropped, not handed out twice);
//   * after RESEED_BLOCKS blocks (64 MiB of output), so that a state captured once does not give away the stream forever;
// and the state is wiped when the thread's generator is destroyed.
inline std::atomic<uint32_t>& fork_generation() {
  static std::atomic<uint32_t> gen{0};
  static const int registered = pthread_atfork(nullptr, nullptr, [] { fork_generation().fetch_add(1, std::memory_order_relaxed); });
  (void)registered;
  return gen;
}
struct ChaChaRng {
  static constexpr uint32_t RESEED_BLOCKS = 1u << 20;
  uint32_t st[16], buf[16];
  int have = 0;
  uint32_t blocks_left = 0;
  uint32_t generation = 0;
  ChaChaRng() { reseed(); }
  ~ChaChaRng() { wipe(st, sizeof st); wipe(buf, sizeof buf); }
  static void wipe(void* p, size_t n) { volatile unsigned char* v = (volatile unsigned char*)p; while (n--) *v++ = 0; }
  static void os_random(void* out, size_t n) {
    unsigned char* p = (unsigned char*)out;
    while (n) {
      const ssize_t got = getrandom(p, n, 0);
      if (got < 0) { if (errno == EINTR) continue; throw std::runtime_error("getrandom failed: no entropy source for witness randomness"); }
      p += got; n -= (size_t)got;
    }
  }
  void reseed() {
    static const uint32_t sigma[4] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    std::memcpy(st, sigma, 16);
    os_random(st + 4, 48);             // 256-bit key, counter, nonce
    st[12] = 0; st[13] = 0;            // 64-bit block counter
    have = 0;
    blocks_left = RESEED_BLOCKS;
    generation = fork_generation().load(std::memory_order_relaxed);
  }
  static uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  void block() {
    if (!blocks_left) reseed();
    blocks_left--;
    uint32_t x[16];
    std::memcpy(x, st, 64);
#define ZKP_QR(a, b, c, d) x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12); x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
    for (int r = 0; r < 10; r++) {
      ZKP_QR(0, 4, 8, 12) ZKP_QR(1, 5, 9, 13) ZKP_QR(2, 6, 10, 14) ZKP_QR(3, 7, 11, 15)
      ZKP_QR(0, 5, 10, 15) ZKP_QR(1, 6, 11, 12) ZKP_QR(2, 7, 8, 13) ZKP_QR(3, 4, 9, 14)
    }
#undef ZKP_QR
    for (int i = 0; i < 16; i++) buf[i] = x[i] + st[i];
    if (++st[12] == 0) ++st[13];
    have = 16;
  }
  uint32_t next() {
    if (generation != fork_generation().load(std::memory_order_relaxed)) reseed();      // this is a forked child: new key, buffered words dropped
    if (!have) block();
    return buf[--have];
  }
  static ChaChaRng& local() { static thread_local ChaChaRng r; return r; }
};
}  // namespace detail

class BigInt {
 public:
  std::vector<uint32_t> l;   // magnitude: little-endian limbs, no trailing zero limbs (zero = empty)
  bool neg = false;          // sign (zero is never negative)

  BigInt() {}
  BigInt(uint64_t v) { while (v) { l.push_back((uint32_t)v); v >>= 32; } }
  static BigInt from_i64(int64_t v) { BigInt r(v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v); r.neg = v < 0; return r; }
  static BigInt zero() { return BigInt(); }
  static BigInt one() { return BigInt(1); }

  void trim() { while (!l.empty() && l.back() == 0) l.pop_back(); if (l.empty()) neg = false; }
  bool is_zero() const { return l.empty(); }
  bool is_negative() const { return neg; }
  bool is_odd() const { return !l.empty() && (l[0] & 1); }
  size_t bit_length() const { return l.empty() ? 0 : 32 * (l.size() - 1) + (32 - __builtin_clz(l.back())); }   // of the magnitude
  bool bit(size_t i) const { return i / 32 < l.size() && ((l[i / 32] >> (i % 32)) & 1); }
  BigInt abs() const { BigInt r = *this; r.neg = false; return r; }
  BigInt operator-() const { BigInt r = *this; if (!r.is_zero()) r.neg = !neg; return r; }

  static int cmp_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
  }
  static int cmp(const BigInt& a, const BigInt& b) {
    if (a.neg != b.neg) return a.neg ? -1 : 1;
    const int c = cmp_mag(a.l, b.l);
    return a.neg ? -c : c;
  }
  friend bool operator==(const BigInt& a, const BigInt& b) { return cmp(a, b) == 0; }
  friend bool operator!=(const BigInt& a, const BigInt& b) { return cmp(a, b) != 0; }
  friend bool operator<(const BigInt& a, const BigInt& b) { return cmp(a, b) < 0; }
  friend bool operator>(const BigInt& a, const BigInt& b) { return cmp(a, b) > 0; }
  friend bool operator<=(const BigInt& a, const BigInt& b) { return cmp(a, b) <= 0; }
  friend bool operator>=(const BigInt& a, const BigInt& b) { return cmp(a, b) >= 0; }

 private:
  static std::vector<uint32_t> add_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    std::vector<uint32_t> r(std::max(a.size(), b.size()) + 1);
    uint64_t c = 0;
    for (size_t i = 0; i < r.size(); i++) {
      c += (uint64_t)(i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0);
      r[i] = (uint32_t)c; c >>= 32;
    }
    return r;
  }
  static std::vector<uint32_t> sub_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {   // |a| >= |b|
    std::vector<uint32_t> r(a.size());
    int64_t br = 0;
    for (size_t i = 0; i < a.size(); i++) {
      int64_t d = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - br;
      br = d < 0; r[i] = (uint32_t)d;
    }
    return r;
  }
  static BigInt add_signed(const BigInt& a, const BigInt& b, bool bneg) {
    BigInt r;
    if (a.neg == bneg) { r.l = add_mag(a.l, b.l); r.neg = a.neg; }
    else {
      const int c = cmp_mag(a.l, b.l);
      if (c >= 0) { r.l = sub_mag(a.l, b.l); r.neg = a.neg; } else { r.l = sub_mag(b.l, a.l); r.neg = bneg; }
    }
    r.trim(); return r;
  }

 public:
  friend BigInt operator+(const BigInt& a, const BigInt& b) { return add_signed(a, b, b.neg); }
  friend BigInt operator-(const BigInt& a, const BigInt& b) { return add_signed(a, b, !b.neg); }
  friend BigInt operator*(const BigInt& a, const BigInt& b) {
    BigInt r; if (a.is_zero() || b.is_zero()) return r;
    r.l.assign(a.l.size() + b.l.size(), 0);
    for (size_t i = 0; i < a.l.size(); i++) {
      uint64_t c = 0;
      const uint64_t ai = a.l[i];
      for (size_t j = 0; j < b.l.size(); j++) { c += ai * b.l[j] + r.l[i + j]; r.l[i + j] = (uint32_t)c; c >>= 32; }
      r.l[i + b.l.size()] = (uint32_t)c;
    }
    r.neg = a.neg != b.neg;
    r.trim(); return r;
  }
  BigInt shl(size_t k) const {
    if (is_zero()) return *this;
    BigInt r; r.l.assign(l.size() + k / 32 + 1, 0);
    for (size_t i = 0; i < l.size(); i++) {
      uint64_t v = (uint64_t)l[i] << (k % 32);
      r.l[i + k / 32] |= (uint32_t)v; r.l[i + k / 32 + 1] |= (uint32_t)(v >> 32);
    }
    r.neg = neg;
    r.trim(); return r;
  }
  static BigInt pow2(size_t k) { return one().shl(k); }

  // |a| = q |m| + r, 0 <= r < |m| (magnitudes only): Knuth, TAOCP vol. 2, 4.3.1 algorithm D
  static void divmod_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& m, std::vector<uint32_t>& q, std::vector<uint32_t>& r) {
    if (m.empty()) throw std::domain_error("BigInt: division by zero");
    q.clear(); r.clear();
    if (cmp_mag(a, m) < 0) { r = a; return; }
    const size_t n = m.size(), mm = a.size() - n;
    if (n == 1) {
      q.assign(a.size(), 0);
      uint64_t rem = 0;
      for (size_t i = a.size(); i-- > 0;) { const uint64_t cur = (rem << 32) | a[i]; q[i] = (uint32_t)(cur / m[0]); rem = cur % m[0]; }
      if (rem) r.push_back((uint32_t)rem);
      while (!q.empty() && q.back() == 0) q.pop_back();
      return;
    }
    const int s = __builtin_clz(m.back());
    std::vector<uint32_t> v(n), u(a.size() + 1);
    for (size_t i = n; i-- > 0;) v[i] = (m[i] << s) | (s && i ? m[i - 1] >> (32 - s) : 0);
    u[a.size()] = s ? a.back() >> (32 - s) : 0;
    for (size_t i = a.size(); i-- > 0;) u[i] = (a[i] << s) | (s && i ? a[i - 1] >> (32 - s) : 0);
    q.assign(mm + 1, 0);
    for (size_t j = mm + 1; j-- > 0;) {
      const uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
      uint64_t qh = num / v[n - 1], rh = num % v[n - 1];
      while (qh >> 32 || qh * v[n - 2] > ((rh << 32) | u[j + n - 2])) { qh--; rh += v[n - 1]; if (rh >> 32) break; }
      int64_t borrow = 0; uint64_t carry = 0;
      for (size_t i = 0; i < n; i++) {
        const uint64_t p = qh * v[i] + carry;
        carry = p >> 32;
        const int64_t t = (int64_t)u[i + j] - borrow - (int64_t)(p & 0xFFFFFFFFu);
        u[i + j] = (uint32_t)t; borrow = t < 0;
      }
      const int64_t t = (int64_t)u[j + n] - borrow - (int64_t)carry;
      u[j + n] = (uint32_t)t;
      if (t < 0) {          // qh was one too large: add the divisor back
        qh--;
        uint64_t c = 0;
        for (size_t i = 0; i < n; i++) { c += (uint64_t)u[i + j] + v[i]; u[i + j] = (uint32_t)c; c >>= 32; }
        u[j + n] += (uint32_t)c;
      }
      q[j] = (uint32_t)qh;
    }
    r.resize(n);
    for (size_t i = 0; i < n; i++) r[i] = (u[i] >> s) | (s ? (uint32_t)((uint64_t)u[i + 1] << (32 - s)) : 0);
    while (!r.empty() && r.back() == 0) r.pop_back();
    while (!q.empty() && q.back() == 0) q.pop_back();
  }
  // (truncated quotient, truncated remainder): a = q m + r, r has the sign of a
  static std::pair<BigInt, BigInt> divmod(const BigInt& a, const BigInt& m) {
    BigInt q, r;
    divmod_mag(a.l, m.l, q.l, r.l);
    q.neg = !q.l.empty() && (a.neg != m.neg);
    r.neg = !r.l.empty() && a.neg;
    return {q, r};
  }
  friend BigInt operator%(const BigInt& a, const BigInt& m) { return divmod(a, m).second; }       // Rust `%`
  BigInt modulus(const BigInt& m) const {                                                            // BigInt::modulus: in [0, |m|)
    BigInt r = divmod(*this, m).second;
    return r.neg ? r + m.abs() : r;
  }
  BigInt div_floor(const BigInt& d) const {
    auto qr = divmod(*this, d);
    if (!qr.second.is_zero() && (qr.second.neg != d.neg)) return qr.first - one();
    return qr.first;
  }

  static BigInt gcd(BigInt a, BigInt b) { a.neg = b.neg = false; while (!b.is_zero()) { BigInt t = a % b; a = b; b = t; } return a; }
  // a^-1 mod m (throws if not invertible): extended Euclid with coefficients kept in [0, m)
  static BigInt mod_inv(const BigInt& a, const BigInt& m) {
    BigInt r0 = m, r1 = a.modulus(m), t0 = zero(), t1 = one();
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
  // BigInt::from_str_radix(s, 10) as serialize.rs:28 uses it: an optional leading '-', then digits
  static BigInt from_str_radix10(const std::string& s) {
    BigInt r;
    size_t i = 0;
    const bool minus = !s.empty() && s[0] == '-';
    if (minus) i = 1;
    if (i == s.size()) throw std::invalid_argument("BigInt: empty decimal string");
    while (i < s.size()) {
      uint64_t chunk = 0, scale = 1;
      for (int k = 0; k < 9 && i < s.size(); k++, i++) {
        if (s[i] < '0' || s[i] > '9') throw std::invalid_argument("BigInt: bad decimal digit");
        chunk = chunk * 10 + (uint64_t)(s[i] - '0'); scale *= 10;
      }
      uint64_t c = chunk;
      for (auto& w : r.l) { c += (uint64_t)w * scale; w = (uint32_t)c; c >>= 32; }
      if (c) r.l.push_back((uint32_t)c);
    }
    r.trim();
    r.neg = minus && !r.is_zero();
    return r;
  }
  std::string to_str_radix10() const {
    if (is_zero()) return "0";
    std::vector<uint32_t> v = l;
    std::string s;
    while (!v.empty()) {
      uint64_t rem = 0;
      for (size_t i = v.size(); i-- > 0;) { const uint64_t cur = (rem << 32) | v[i]; v[i] = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u; }
      while (!v.empty() && v.back() == 0) v.pop_back();
      for (int k = 0; k < 9 && (rem || !v.empty()); k++) { s.push_back((char)('0' + rem % 10)); rem /= 10; }
    }
    if (neg) s.push_back('-');
    std::reverse(s.begin(), s.end());
    return s;
  }
  // reference byte form: minimal big-endian magnitude, zero -> one 00 byte ([upstream] curv BigInt::to_bytes)
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
  // fixed-width limb export for the C ABI; throws if the value is negative or does not fit (non-canonical operand)
  bool fits_limbs(size_t n) const { return !neg && l.size() <= n; }
  void to_limbs(uint32_t* out, size_t n) const {
    if (neg) throw std::length_error("BigInt: negative operand at the fixed-width ABI");
    if (l.size() > n) throw std::length_error("BigInt: operand wider than the fixed ABI width");
    std::copy(l.begin(), l.end(), out);
    std::fill(out + l.size(), out + n, 0u);
  }
  static BigInt from_limbs(const uint32_t* p, size_t n) { while (n && p[n - 1] == 0) n--; BigInt r; r.l.assign(p, p + n); return r; }
  std::string to_hex() const {
    if (is_zero()) return "0";
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = l.size(); i-- > 0;) for (int k = 28; k >= 0; k -= 4) s.push_back(d[(l[i] >> k) & 15]);
    s = s.substr(s.find_first_not_of('0'));
    return neg ? "-" + s : s;
  }

  // ---- sampling ([upstream] curv Samplable: OS randomness; here a ChaCha20 stream keyed from the OS, one per thread)
  static BigInt sample(size_t bits) {
    detail::ChaChaRng& g = detail::ChaChaRng::local();
    BigInt r; r.l.resize((bits + 31) / 32);
    for (auto& w : r.l) w = g.next();
    if (bits % 32) r.l.back() &= (1u << (bits % 32)) - 1;
    r.trim(); return r;
  }
  static bool coin() { return detail::ChaChaRng::local().next() & 1; }
  static BigInt sample_below(const BigInt& upper) {
    if (upper.is_zero() || upper.neg) throw std::domain_error("BigInt: sample_below needs a positive bound");
    for (;;) { BigInt c = sample(upper.bit_length()); if (c < upper) return c; }
  }
  static BigInt sample_range(const BigInt& lo, const BigInt& hi) { return lo + sample_below(hi - lo); }
};

}  // namespace zkproofs
