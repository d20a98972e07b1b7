// Staging memory of the host API (host/zkproofs.hpp): the flat limb arrays a batch call hands to the C ABI and receives from it.
//
// RawBuf: memory that is written in full before it is read (by to_limbs, or by the GPU call): NOT value-initialised — a std::vector would
// memset (and page-fault) 2 GB per 4096-proof call on one thread before the work starts.  Large buffers are 2 MB aligned and marked for
// transparent huge pages: what such a buffer costs is its FIRST TOUCH (a quarter of a million 4 KB faults per gigabyte: the D2H copy of a
// prove call took 125 ms longer into fresh memory than into touched memory, flattening 0.78 GB of received proofs 95 – 113 ms instead
// of 13), and a huge page is one fault per 2 MB where the kernel grants it.
//
// StagingPool: where the kernel does not grant them (THP off, or no free 2 MB frames: seen on boxes of the pool, profiles/r05/host_pipeline),
// the only cure is not to touch fresh memory: blocks of >= 4 MB go back to a process-wide pool instead of to free() and the next batch call
// of a similar size takes them from there, pages mapped.  Bounded ($ZKP_HOST_POOL_MB, default 2048 = the staging of one 4096-proof
// prove_batch; 0 = no pool), oldest block out first; a block flagged `secret` (witnesses, nonces) is wiped before it is parked or freed.
#pragma once
#include <sys/mman.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>
#include <new>

namespace zkproofs {

inline void wipe_bytes(void* p, size_t n) {
  std::memset(p, 0, n);
  __asm__ __volatile__("" : : "r"(p) : "memory");          // (the stores are not dead: the block is read again by whoever takes it next)
}

class StagingPool {
 public:
  static constexpr size_t HUGE = size_t(2) << 20, POOLED_FROM = size_t(4) << 20;
  // (never destroyed: the threads that release a batch's staging off the critical path may still run when main's statics go)
  static StagingPool& instance() { static StagingPool* p = new StagingPool; return *p; }
  static size_t rounded(size_t bytes) { return (bytes + HUGE - 1) & ~(HUGE - 1); }
  // a parked block of at least `bytes` and at most 5/4 of it, or nullptr; *got = its size
  void* take(size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> g(mu_);
    auto best = blocks_.end();
    for (auto it = blocks_.begin(); it != blocks_.end(); ++it)
      if (it->bytes >= bytes && it->bytes <= bytes + bytes / 4 && (best == blocks_.end() || it->bytes < best->bytes)) best = it;
    if (best == blocks_.end()) { misses_++; return nullptr; }
    void* p = best->p;
    *got = best->bytes; held_ -= best->bytes; hits_++;
    blocks_.erase(best);
    return p;
  }
  // parks the block (making room by freeing the oldest ones) or frees it
  void give(void* p, size_t bytes) {
    std::list<Block> out;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (bytes <= cap_) {
        while (held_ + bytes > cap_) { held_ -= blocks_.front().bytes; out.splice(out.end(), blocks_, blocks_.begin()); }
        blocks_.push_back(Block{p, bytes}); held_ += bytes; p = nullptr;
      }
    }
    std::free(p);
    for (Block& b : out) std::free(b.p);
  }
  void set_capacity(size_t bytes) {
    std::list<Block> out;
    {
      std::lock_guard<std::mutex> g(mu_);
      cap_ = bytes;
      while (held_ > cap_) { held_ -= blocks_.front().bytes; out.splice(out.end(), blocks_, blocks_.begin()); }
    }
    for (Block& b : out) std::free(b.p);
  }
  // give everything that is parked back to the OS (a long-running service after a burst of large batches; the pool refills on demand)
  void trim() {
    std::list<Block> out;
    {
      std::lock_guard<std::mutex> g(mu_);
      out.swap(blocks_);
      held_ = 0;
    }
    for (Block& b : out) std::free(b.p);
  }
  size_t capacity() { std::lock_guard<std::mutex> g(mu_); return cap_; }
  size_t held() { std::lock_guard<std::mutex> g(mu_); return held_; }
  size_t hits() { std::lock_guard<std::mutex> g(mu_); return hits_; }
  size_t misses() { std::lock_guard<std::mutex> g(mu_); return misses_; }

 private:
  struct Block { void* p; size_t bytes; };
  StagingPool() {
    const char* e = std::getenv("ZKP_HOST_POOL_MB");
    cap_ = (e ? (size_t)std::max(0ll, std::atoll(e)) : size_t(2048)) << 20;
  }
  std::mutex mu_;
  std::list<Block> blocks_;                                  // oldest first
  size_t cap_ = 0, held_ = 0, hits_ = 0, misses_ = 0;
};

template <class T> struct RawBuf {
  T* p = nullptr;
  size_t block = 0;                                          // bytes of a pooled block, 0 for a small malloc'ed one
  size_t bytes = 0;
  bool secret = false;
  explicit RawBuf(size_t n, bool secret_ = false) : bytes(std::max<size_t>(n, 1) * sizeof(T)), secret(secret_) {
    if (bytes >= StagingPool::POOLED_FROM) {
      block = StagingPool::rounded(bytes);
      size_t got = 0;
      p = static_cast<T*>(StagingPool::instance().take(block, &got));
      if (p) block = got;
      else {
        p = static_cast<T*>(std::aligned_alloc(StagingPool::HUGE, block));
        if (p) (void)madvise(p, block, MADV_HUGEPAGE);
      }
    } else p = static_cast<T*>(std::malloc(bytes));
    if (!p) throw std::bad_alloc();
  }
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  // a secret block is wiped where its owner decides (wipe_now: on the calling thread, before the block is handed to a background
  // release) or, at the latest, here
  void wipe_now() {
    if (secret && p) wipe_bytes(p, bytes);
    secret = false;
  }
  bool pooled() const { return block != 0; }
  ~RawBuf() {
    if (secret) wipe_bytes(p, bytes);
    if (block) StagingPool::instance().give(p, block);
    else std::free(p);
  }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

}  // namespace zkproofs
