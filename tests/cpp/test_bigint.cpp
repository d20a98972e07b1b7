// TEST INFRASTRUCTURE: host/bigint.hpp against GMP (the library the reference's BigInt wraps) on random signed operands of mixed sizes:
// + - * , truncated % (mpz_tdiv_r), modulus (mpz_mod), div_floor (mpz_fdiv_q), gcd, mod_inv, decimal text both ways, to_bytes.
//   g++ -O2 -std=c++17 -I/opt/conda/include tests/cpp/test_bigint.cpp -L/opt/conda/lib -lgmp -Wl,-rpath,/opt/conda/lib -o build/test_bigint
#include <gmp.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/wait.h>
#include <unistd.h>
#include "../../zk-paillier_amd/host/bigint.hpp"
using zkproofs::BigInt;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

static BigInt random_int(int max_limbs) {
  BigInt r;
  const int n = rnd() % (max_limbs + 1);
  r.l.resize(n);
  const int style = rnd() % 5;
  for (auto& w : r.l) w = style == 0 ? 0xFFFFFFFFu : style == 1 ? (rnd() & 1 ? 0 : 0xFFFFFFFFu) : style == 2 ? 0x80000000u : rnd();
  r.trim();
  if (!r.is_zero() && (rnd() & 1)) r.neg = true;
  return r;
}
static void to_mpz(mpz_t z, const BigInt& a) {
  mpz_import(z, a.l.size(), -1, 4, 0, 0, a.l.data());
  if (a.neg) mpz_neg(z, z);
}
static bool same(const BigInt& a, const mpz_t z) {
  mpz_t t; mpz_init(t); to_mpz(t, a);
  const bool ok = mpz_cmp(t, z) == 0 && (a.l.empty() ? !a.neg : a.l.back() != 0);
  mpz_clear(t);
  return ok;
}
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL %s (iteration %d)\n", what, it); return 1; } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
  mpz_t x, y, z, w;
  mpz_inits(x, y, z, w, NULL);
  for (int it = 0; it < iters; it++) {
    const BigInt a = random_int(it % 3 == 0 ? 270 : 40), b = random_int(it % 5 == 0 ? 140 : 20);
    to_mpz(x, a); to_mpz(y, b);
    mpz_add(z, x, y); CHECK(same(a + b, z), "add");
    mpz_sub(z, x, y); CHECK(same(a - b, z), "sub");
    mpz_mul(z, x, y); CHECK(same(a * b, z), "mul");
    CHECK((mpz_cmp(x, y) < 0) == (a < b) && (mpz_cmp(x, y) == 0) == (a == b) && (mpz_cmp(x, y) > 0) == (a > b), "cmp");
    if (!b.is_zero()) {
      mpz_tdiv_r(z, x, y); CHECK(same(a % b, z), "truncated remainder");
      mpz_mod(z, x, y); CHECK(same(a.modulus(b), z), "floored modulus");
      mpz_fdiv_q(z, x, y); CHECK(same(a.div_floor(b), z), "div_floor");
      mpz_tdiv_qr(z, w, x, y);
      auto qr = BigInt::divmod(a, b);
      CHECK(same(qr.first, z) && same(qr.second, w), "divmod");
    }
    mpz_gcd(z, x, y); CHECK(same(BigInt::gcd(a, b), z), "gcd");
    {
      const std::string s = a.to_str_radix10();
      char* g = mpz_get_str(nullptr, 10, x);
      CHECK(s == g, "to decimal");
      free(g);
      CHECK(BigInt::from_str_radix10(s) == a, "from decimal");
    }
    {
      const auto by = a.to_bytes();
      size_t cnt = 0;
      uint8_t buf[4 * 280];
      mpz_export(buf, &cnt, 1, 1, 0, 0, x);
      if (cnt == 0) { buf[0] = 0; cnt = 1; }
      CHECK(by.size() == cnt && std::memcmp(by.data(), buf, cnt) == 0, "to_bytes");
      CHECK(BigInt::from_bytes(by) == a.abs(), "from_bytes");
    }
    if (!b.is_zero() && !b.neg && b > BigInt(1)) {
      const bool inv = mpz_invert(z, x, y) != 0;
      bool threw = false; BigInt r;
      try { r = BigInt::mod_inv(a, b); } catch (const std::domain_error&) { threw = true; }
      CHECK(inv == !threw && (!inv || same(r, z)), "mod_inv");
    }
    CHECK(same(a.shl(it % 97), (mpz_mul_2exp(z, x, it % 97), z)), "shl");
  }
  // sampling stays inside its bounds and is not constant
  {
    int it = -1;
    const BigInt lo = BigInt::pow2(200), hi = BigInt::pow2(201);
    BigInt first = BigInt::sample_range(lo, hi); bool differ = false;
    for (int i = 0; i < 200; i++) { BigInt v = BigInt::sample_range(lo, hi); CHECK(v >= lo && v < hi, "sample_range"); differ |= v != first; }
    CHECK(differ, "sampling is constant");
  }
  // a forked child must not replay the parent's stream (round-4 advisor finding): both continue from the same generator state; the child
  // re-keys from the OS because its pid differs, the parent carries on with its own stream
  {
    int it = -2;
    (void)BigInt::sample(256);                                  // the thread's generator exists and has a block in its buffer
    int fd[2];
    CHECK(pipe(fd) == 0, "pipe");
    const pid_t pid = fork();
    if (pid == 0) {
      std::string out;
      for (int i = 0; i < 6; i++) out += BigInt::sample(256).to_str_radix10() + "\n";     // spans the buffered block and the next ones
      (void)!write(fd[1], out.data(), out.size());
      _exit(0);
    }
    close(fd[1]);
    std::string child; char tmp[4096]; ssize_t got;
    while ((got = read(fd[0], tmp, sizeof tmp)) > 0) child.append(tmp, (size_t)got);
    int status = 0; waitpid(pid, &status, 0);
    std::string parent;
    for (int i = 0; i < 6; i++) parent += BigInt::sample(256).to_str_radix10() + "\n";
    // (the words buffered at the fork are dropped by the child, not handed out a second time)
    size_t same_lines = 0, pos_c = 0, pos_p = 0;
    for (int i = 0; i < 6; i++) {
      const size_t ec = child.find('\n', pos_c), ep = parent.find('\n', pos_p);
      CHECK(ec != std::string::npos && ep != std::string::npos, "six samples each");
      same_lines += child.substr(pos_c, ec - pos_c) == parent.substr(pos_p, ep - pos_p);
      pos_c = ec + 1; pos_p = ep + 1;
    }
    CHECK(WIFEXITED(status) && same_lines == 0, "a forked child replays its parent's random stream");
  }
  std::printf("bigint ok: %d iterations\n", iters);
  return 0;
}
