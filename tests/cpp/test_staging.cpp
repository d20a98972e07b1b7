// host/staging.hpp on its own (no GPU, no library): blocks of >= 4 MB come back from the pool with their pages mapped, secret blocks come
// back wiped, the pool stays within its capacity (oldest block out first), capacity 0 switches it off, and it is safe under threads.
#include <cstdio>
#include <thread>
#include <vector>
#include "../../zk-paillier_amd/host/staging.hpp"
using namespace zkproofs;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
  StagingPool& pool = StagingPool::instance();
  pool.set_capacity(size_t(64) << 20);
  const size_t N = (size_t(8) << 20) / 4;                    // 8 MB of words
  uint32_t* first;
  {
    RawBuf<uint32_t> a(N);
    first = a.data();
    CHECK((reinterpret_cast<uintptr_t>(first) & (StagingPool::HUGE - 1)) == 0);
    for (size_t i = 0; i < N; i += 1024) a[i] = 0xabcd0000u + (uint32_t)i;
  }
  CHECK(pool.held() == (size_t(8) << 20));
  {
    RawBuf<uint32_t> b(N - 100);                             // a slightly smaller request takes the same block
    CHECK(b.data() == first);
    CHECK(b[1024] == 0xabcd0000u + 1024);                    // not wiped: it was not secret
    CHECK(pool.held() == 0);
  }
  {
    RawBuf<uint32_t> s(N, true);
    CHECK(s.data() == first);
    for (size_t i = 0; i < N; i++) s[i] = 0x5ec2e7u;
  }
  {
    RawBuf<uint32_t> c(N);
    CHECK(c.data() == first);
    bool zero = true;
    for (size_t i = 0; i < N; i++) zero &= c[i] == 0;
    CHECK(zero);                                             // the secret block was wiped before it was parked
    RawBuf<uint32_t> big(4 * N);                             // no block of that size: fresh memory
    CHECK(big.data() != first);
    CHECK(pool.misses() >= 2);
  }
  CHECK(pool.held() == (size_t(40) << 20));
  {
    RawBuf<uint32_t> tiny(16);                               // small buffers do not go through the pool
    CHECK(tiny.block == 0);
    RawBuf<uint32_t> half(N / 2 + 1);                        // 4 MB + : a block of twice the size is not handed out for it
    CHECK(half.data() != first);
  }
  // capacity: 64 MB holds 8 + 32 + 6 (the `half` block) ...; parking 32 MB more evicts the oldest
  { RawBuf<uint32_t> more(4 * N + 7); (void)more; { RawBuf<uint32_t> more2(4 * N + 9); (void)more2; } }
  CHECK(pool.held() <= pool.capacity());
  pool.set_capacity(0);
  CHECK(pool.held() == 0);
  { RawBuf<uint32_t> d(N); d[0] = 1; }
  CHECK(pool.held() == 0);                                   // switched off: straight back to free()
  pool.set_capacity(size_t(256) << 20);
  std::vector<std::thread> th;
  for (int t = 0; t < 8; t++)
    th.emplace_back([&, t] {
      for (int k = 0; k < 50; k++) {
        RawBuf<uint32_t> x(N + (size_t)((t * 131 + k * 17) % 4096), (k & 1) != 0);
        x[0] = (uint32_t)k; x[N - 1] = (uint32_t)t;
        if (x[0] != (uint32_t)k || x[N - 1] != (uint32_t)t) { std::printf("FAIL thread %d\n", t); fails++; }
      }
    });
  for (auto& x : th) x.join();
  CHECK(pool.held() <= pool.capacity());
  CHECK(pool.hits() > 100);
  // wipe_now: the owner wipes a secret block on its own thread, ahead of a background release (host/zkproofs.hpp wipe_secrets)
  {
    RawBuf<uint32_t> s2(N, true);
    for (size_t i = 0; i < N; i += 512) s2[i] = 0x5ec2e7u;
    s2.wipe_now();
    bool zero = true;
    for (size_t i = 0; i < N; i += 512) zero &= s2[i] == 0;
    CHECK(zero && !s2.secret && s2.pooled());
    RawBuf<uint32_t> tiny2(8, true);
    CHECK(!tiny2.pooled());
  }
  // trim: a long-running service gives the parked blocks back
  CHECK(pool.held() > 0);
  pool.trim();
  CHECK(pool.held() == 0);
  { RawBuf<uint32_t> again(N); again[0] = 3; }
  CHECK(pool.held() == (size_t(8) << 20));                   // ... and the pool refills on demand
  std::printf(fails ? "staging FAILED\n" : "staging ok (hits %zu, misses %zu)\n", pool.hits(), pool.misses());
  return fails != 0;
}
