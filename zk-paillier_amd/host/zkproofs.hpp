// zkproofs.hpp — C++ host mirror of the reference's proof API for the hot path, on top of the C
// ABI of include/zkp_hip.h.  (The reference is Rust; there is no Rust toolchain in this image, so
// the host side above the C ABI is C++ — same type and method names, argument meaning and error
// behaviour as the reference, so that tests read like the reference's own tests.)
//
//   reference                                                      here
//   zkproofs::RangeProofNi::{prove,verify,verify_self}              RangeProofNi::{prove,verify,verify_self}   (+ *_batch)
//   zkproofs::{MulProof, CorrectMessageProof}::{prove,verify}, BigInt::mod_inv    MulProof, CorrectMessageProof, mod_inv_batch
//   zkproofs::RangeProof::{verifier_commit,verify_commit,generate_encrypted_pairs,generate_proof,verifier_output}   RangeProof::*
//     src/zkproofs/range_proof_ni.rs:36-128
//   zkproofs::NiCorrectKeyProof::{proof,verify}                     NiCorrectKeyProof::{proof,verify}
//     src/zkproofs/correct_key_ni.rs:36-100
//   zkproofs::{CompositeDLogProof,DLogStatement}::{prove,verify}    same
//     src/zkproofs/wi_dlog_proof.rs:33-91
//   zkproofs::{CorrectKey,Challenge,VerificationAid,CorrectKeyProof} same (interactive, correct_key.rs:28-183)
//   zkproofs::{VerlinProof,VerlinStatement,VerlinWitness}, gen_phi   same (verlin_proof.rs:35-165)
//   zkproofs::{ZeroProof,ZeroStatement,ZeroWitness}                 same (zero_enc_proof.rs:26-95)
//   zkproofs::{CiphertextProof,CiphertextStatement,CiphertextWitness} same (correct_ciphertext.rs:23-98)
//   paillier::{Keypair,EncryptionKey,DecryptionKey,Paillier}        same names, only what the path needs
//   zkproofs::IncorrectProof (errors.rs:5-13)                       Result<> with is_ok()/is_err()/expect()
//
// Rust panics (assert_eq!, index out of bounds, .expect on Err) become C++ exceptions (Panic).
// Every modular exponentiation runs on the GPU; this layer only samples, hashes small transcripts
// (NiCorrectKeyProof::proof's MGF), converts BigInt <-> limbs and flattens proofs into batches.
#pragma once
#include <sys/mman.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/zkp_hip.h"
#include "bigint.hpp"
#include "staging.hpp"

namespace zkproofs {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
struct IncorrectProof {};   // src/zkproofs/errors.rs:5-13

// Result<(), IncorrectProof>.  A batch call returns one Result per proof; a proof on which the reference would have
// PANICKED (index out of bounds, assert) carries that panic and throws it when it is looked at, so that one crafted proof
// cannot take the verdicts of the others with it.
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
class Result {
  enum State { Ok, Err, Panicked, Unsupp } st_;
  std::string what_;
  void look() const { if (st_ == Panicked) throw Panic(what_); if (st_ == Unsupp) throw Unsupported(what_); }
 public:
  explicit Result(bool ok) : st_(ok ? Ok : Err) {}
  static Result panicked(const std::string& what) { Result r(false); r.st_ = Panicked; r.what_ = what; return r; }
  // the ONLY outcome that is not the reference's: a statement this fixed-width engine cannot carry at all (a key that is not a
  // positive odd integer of at most 4096 bits).  Every other input — any size, either sign — gets the reference's verdict or panic.
  static Result unsupported(const std::string& what) { Result r(false); r.st_ = Unsupp; r.what_ = what; return r; }
  bool would_panic() const { return st_ == Panicked; }
  bool is_unsupported() const { return st_ == Unsupp; }
  bool is_ok() const { look(); return st_ == Ok; }
  bool is_err() const { look(); return st_ == Err; }
  void expect(const char* msg) const { look(); if (st_ != Ok) throw Panic(std::string(msg) + ": IncorrectProof"); }
};

// ------------------------------------------------------------------ engine (one ctx per device)
class Engine {
  zkp_ctx* ctx_ = nullptr;
 public:
  explicit Engine(int device = 0) {
    if (zkp_ctx_create(device, &ctx_) != ZKP_OK) throw std::runtime_error("zkp_ctx_create failed: no gfx950 GPU (there is no CPU fallback)");
  }
  ~Engine() { if (ctx_) zkp_ctx_destroy(ctx_); }
  Engine(const Engine&) = delete;
  zkp_ctx* ctx() const { return ctx_; }
  void check(int32_t st, const char* what) const {
    if (st != ZKP_OK) throw std::runtime_error(std::string(what) + " failed (status " + std::to_string(st) + "): " + zkp_last_error_string(ctx_));
  }
  static Engine& instance() { static Engine e(0); return e; }
};

// ------------------------------------------------------------------ host-side parallelism and timing of the batch calls
// The reference spreads a proof's rows over a rayon pool (range_proof.rs:136-187); here the per-proof HOST work of a batch call —
// sampling 4 x 128 values, BigInt <-> limb conversion, rebuilding the proof objects — runs on a few threads (ZKP_HOST_THREADS,
// default min(16, hardware threads)) around the one GPU call.  last_host_timing(): where the time of the most recent
// RangeProofNi::{prove,verify}_batch went (bench.py's host_api leg prints it next to the GPU step).
struct HostTiming { double sample_flatten_ms = 0, gpu_ms = 0, rebuild_ms = 0, general_ms = 0; size_t proofs = 0, general_proofs = 0; unsigned threads = 1; };
inline HostTiming& last_host_timing() { static HostTiming t; return t; }
inline unsigned host_threads() {
  static const unsigned n = [] {
    const char* e = std::getenv("ZKP_HOST_THREADS");
    unsigned v = e ? (unsigned)std::atoi(e) : std::min(16u, std::thread::hardware_concurrency());
    return std::max(1u, std::min(v, 64u));
  }();
  return n;
}
template <class F> inline void parallel_for(size_t count, F body, unsigned max_threads = ~0u) {
  const size_t T = std::min<size_t>(std::min(host_threads(), max_threads), count);
  if (T <= 1) { for (size_t i = 0; i < count; i++) body(i); return; }
  std::exception_ptr first;
  std::mutex mu;
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++)
    th.emplace_back([&, t] {
      try { for (size_t i = count * t / T; i < count * (t + 1) / T; i++) body(i); }
      catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    });
  for (auto& x : th) x.join();
  if (first) std::rethrow_exception(first);
}
// (RawBuf, StagingPool: host/staging.hpp)
struct StopWatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap() { const auto t1 = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; return ms; }
};

// kernel width (bits) for an n of the given size
inline uint32_t width_for(const BigInt& n) {
  const size_t b = n.bit_length();
  if (b <= 1024) return 1024;
  if (b <= 2048) return 2048;
  if (b <= 4096) return 4096;
  throw std::length_error("modulus wider than 4096 bits");
}

// ------------------------------------------------------------------ Paillier key types
struct EncryptionKey {
  BigInt n, nn;
  friend bool operator==(const EncryptionKey& a, const EncryptionKey& b) { return a.n == b.n && a.nn == b.nn; }
};
struct DecryptionKey { BigInt p, q; };
struct Keypair {
  BigInt p, q;
  // [upstream paillier::Keypair::keys]: ek = {n, nn = n^2}, dk = {p, q}
  std::pair<EncryptionKey, DecryptionKey> keys() const { BigInt n = p * q; return {EncryptionKey{n, n * n}, DecryptionKey{p, q}}; }
};

struct Paillier {
  // [upstream kzen-paillier EncryptWithChosenRandomness] on ANY integers m, r (a received proof may hold negative or over-wide
  // fields), exactly as the reference's operators give it:
  //     rn = mod_pow(r, n, nn) in [0, nn);   gm = (m n + 1) % nn  (truncated: negative for m < 0);   c = gm rn % nn
  // With E = Enc(m mod n, r mod n) — floored residues, the canonical operands the kernels take — that is  c = E  for m >= 0 and
  // c = E - nn (or 0 when E = 0) for m < 0: (m n + 1) % nn = -(nn - G) with G = 1 + (m mod n) n, hence c = -((nn - G) rn mod nn).
  // ONE zkp_paillier_enc_batch for the whole list.
  static std::vector<BigInt> encrypt_with_chosen_randomness_batch(const EncryptionKey& ek, const std::vector<std::pair<const BigInt*, const BigInt*>>& mr) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t cnt = mr.size();
    std::vector<uint32_t> n(kw), mm(cnt * kw), rr(cnt * kw), c(cnt * 2 * kw);
    ek.n.to_limbs(n.data(), kw);
    for (size_t i = 0; i < cnt; i++) {
      const BigInt &m = *mr[i].first, &r = *mr[i].second;
      (m.fits_limbs(kw) && m < ek.n ? m : m.modulus(ek.n)).to_limbs(&mm[i * kw], kw);
      (r.fits_limbs(kw) && r < ek.n ? r : r.modulus(ek.n)).to_limbs(&rr[i * kw], kw);
    }
    if (cnt) e.check(zkp_paillier_enc_batch(e.ctx(), nb, cnt, n.data(), 0, mm.data(), rr.data(), c.data(), 0), "zkp_paillier_enc_batch");
    std::vector<BigInt> out;
    out.reserve(cnt);
    for (size_t i = 0; i < cnt; i++) {
      BigInt E = BigInt::from_limbs(&c[i * 2 * kw], 2 * kw);
      out.push_back(mr[i].first->is_negative() && !E.is_zero() ? E - ek.nn : E);
    }
    return out;
  }
  static BigInt encrypt_with_chosen_randomness(const EncryptionKey& ek, const BigInt& m, const BigInt& r) {
    return encrypt_with_chosen_randomness_batch(ek, {{&m, &r}})[0];
  }
};

inline BigInt mod_pow(const BigInt& base, const BigInt& exp, const BigInt& modulus) {
  Engine& e = Engine::instance();
  const size_t b = modulus.bit_length();
  const uint32_t mb = b <= 2048 ? 2048 : b <= 4096 ? 4096 : 8192, L = mb / 32;
  if (exp.bit_length() > mb) throw std::length_error("exponent wider than the modulus width");
  std::vector<uint32_t> bb(L), ee(L), mm(L), out(L);
  base.modulus(modulus).to_limbs(bb.data(), L); exp.to_limbs(ee.data(), L); modulus.to_limbs(mm.data(), L);   // mpz_powm: a negative base gives the residue in [0, m)
  e.check(zkp_modexp_batch(e.ctx(), mb, mb, 1, bb.data(), ee.data(), L, mm.data(), L, out.data(), 0), "zkp_modexp_batch");
  return BigInt::from_limbs(out.data(), L);
}

// batched BigInt::mod_pow over one modulus: out[i] = bases[i]^exps[i or 0] mod modulus (ONE zkp_modexp_batch)
inline std::vector<BigInt> mod_pow_batch(const std::vector<BigInt>& bases, const std::vector<BigInt>& exps, const BigInt& modulus) {
  Engine& e = Engine::instance();
  const size_t b = modulus.bit_length(), cnt = bases.size();
  const uint32_t mb = b <= 2048 ? 2048 : b <= 4096 ? 4096 : 8192, L = mb / 32;
  const bool shared = exps.size() == 1;
  if (!shared && exps.size() != cnt) throw std::invalid_argument("mod_pow_batch: exps must have 1 or bases.size() entries");
  std::vector<uint32_t> bb(cnt * L), ee(exps.size() * L), mm(L), out(cnt * L);
  for (size_t i = 0; i < cnt; i++) bases[i].modulus(modulus).to_limbs(&bb[i * L], L);
  for (size_t i = 0; i < exps.size(); i++) exps[i].to_limbs(&ee[i * L], L);
  modulus.to_limbs(mm.data(), L);
  if (cnt) e.check(zkp_modexp_batch(e.ctx(), mb, mb, cnt, bb.data(), ee.data(), shared ? 0 : L, mm.data(), 0, out.data(), 0), "zkp_modexp_batch");
  std::vector<BigInt> r;
  for (size_t i = 0; i < cnt; i++) r.push_back(BigInt::from_limbs(&out[i * L], L));
  return r;
}

// ------------------------------------------------------------------ host SHA-256 (NiCorrectKeyProof::proof's MGF; the challenge of a RangeProofNi whose transcript holds over-wide values)
namespace detail {
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len = 0; size_t fill = 0;
  static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  Sha256() { static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}; std::memcpy(h, iv, 32); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) { size_t k = std::min(n, 64 - fill); std::memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 64) { block(buf); fill = 0; } }
  }
  void update(const BigInt& v) { auto b = v.to_bytes(); update(b.data(), b.size()); }
  BigInt finish() {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1); pad = 0;
    while (fill != 56) update(&pad, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    uint8_t out[32]; for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
    return BigInt::from_bytes(out, 32);
  }
};
// src/zkproofs/utils.rs:9-22
inline BigInt compute_digest(std::initializer_list<const BigInt*> items) { Sha256 s; for (auto* v : items) s.update(*v); return s.finish(); }
}  // namespace detail

// ------------------------------------------------------------------ RangeProofNi
struct Response {           // src/zkproofs/range_proof.rs:53-78
  enum Kind { Open, Mask } kind = Open;
  BigInt w1, r1, w2, r2;    // Open
  uint8_t j = 0;            // Mask
  BigInt masked_x, masked_r;
};
struct EncryptedPairs { std::vector<BigInt> c1, c2; };   // range_proof.rs:32-39
struct Proof { std::vector<Response> responses; };       // range_proof.rs:80-81

class RangeProofNi {
 public:
  static constexpr size_t SECURITY_PARAMETER = ZKP_SECURITY_PARAMETER;   // range_proof_ni.rs:23
  EncryptionKey ek;
  BigInt range, ciphertext;
  EncryptedPairs encrypted_pairs;
  Proof proof;
  size_t error_factor = SECURITY_PARAMETER;

  struct Statement { BigInt range, ciphertext, secret_x, secret_r; };

  // range_proof_ni.rs:47-82 for many provers sharing one key; randomness sampled as range_proof.rs:133-159.
  // ONE GPU call per batch (round 5; rounds 3 - 4 ran a pipeline of 4 chunks to hide the rebuild of the proof objects behind the next
  // chunk's call — and paid a launch tail and a copy per chunk for it: 2022 ms of GPU calls against 1794 ms for one call at B = 4096,
  // measured on one box).  What made the rebuild expensive was never the copying of 0.8 GB of limbs but the first touch of the heap
  // they go to (3 million small allocations, a quarter of a million page faults): so the proof objects are ALLOCATED AND TOUCHED WHILE
  // THE GPU WORKS (`prealloc`: every BigInt of every proof at its full capacity, on a few helper threads under the blocking call), and
  // after the call `fill` only copies limbs into memory that is already there, on all threads.  ZKP_HOST_PIPELINE=N (N > 1) still cuts
  // the batch into N calls — the next chunk's sampling and the previous chunk's fill then run under a call as well.
  // chunks of a pipelined batch call, the same in prove_batch and verify_batch: ZKP_HOST_PIPELINE = 0 / 1 (one call) or N (up to N
  // chunks of >= 1024 proofs); default `dflt`.  (verify_batch alone also knows ZKP_HOST_PIPELINE=uneven: a quarter, then the rest.)
  static size_t pipeline_chunks(size_t B, size_t dflt) {
    const char* pe = std::getenv("ZKP_HOST_PIPELINE");
    size_t want = dflt;
    if (pe && pe[0] >= '0' && pe[0] <= '9') want = std::max<size_t>(1, (size_t)std::atoi(pe));
    return std::max<size_t>(1, std::min<size_t>(want, B / 1024));
  }
  // The staging buffers of a finished LARGE call (2.2 GB at B = 4096) go back to the pool / the OS on a thread of their own: unmapping
  // them is tens of milliseconds that the caller need not wait for.  Small calls (nothing pooled: a single proof is a batch of one) free
  // theirs inline — no thread per call.  Secret blocks are wiped by the CALLER before this (wipe_secrets below), never only in the
  // background: a process that exits right after the call must not leave witnesses behind.
  template <class Chunks> static void release_later(Chunks&& chunks, bool large) {
    if (!large) { chunks.clear(); return; }
    try { std::thread([held = std::move(chunks)]() mutable { held.clear(); }).detach(); } catch (...) {}     // (no thread: freed here, by `chunks` going out of scope)
  }
  struct ProveChunk {
    size_t lo, hi;
    RawBuf<uint32_t> range, ct, x, r, w1, w2, r1, r2, c1, c2, rw1, rr1, rw2, rr2;
    RawBuf<uint8_t> kind, jj;
    std::vector<uint8_t> status;
    void wipe_secrets() { for (RawBuf<uint32_t>* b : {&x, &r, &w1, &w2, &r1, &r2}) b->wipe_now(); }
    bool large() const { return c1.pooled(); }
    ProveChunk(size_t lo_, size_t hi_, size_t kw, size_t EF)
        : lo(lo_), hi(hi_), range((hi_ - lo_) * kw), ct((hi_ - lo_) * 2 * kw), x((hi_ - lo_) * kw, true), r((hi_ - lo_) * kw, true), w1((hi_ - lo_) * EF * kw, true),
          w2((hi_ - lo_) * EF * kw, true), r1((hi_ - lo_) * EF * kw, true), r2((hi_ - lo_) * EF * kw, true) /* witnesses and nonces: wiped on release */, c1((hi_ - lo_) * EF * 2 * kw), c2((hi_ - lo_) * EF * 2 * kw), rw1((hi_ - lo_) * EF * kw), rr1((hi_ - lo_) * EF * kw),
          rw2((hi_ - lo_) * EF * kw), rr2((hi_ - lo_) * EF * kw), kind((hi_ - lo_) * EF), jj((hi_ - lo_) * EF), status(hi_ - lo_) {}
  };
  static std::vector<RangeProofNi> prove_batch(const EncryptionKey& ek, const std::vector<Statement>& st) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t B = st.size(), EF = SECURITY_PARAMETER;
    RawBuf<uint32_t> n(kw);
    ek.n.to_limbs(n.data(), kw);
    HostTiming& tm = last_host_timing();
    tm = HostTiming(); tm.proofs = B; tm.threads = host_threads();
    const size_t chunks = pipeline_chunks(B, 1);
    std::vector<RangeProofNi> out(B);

    // stage A: sample (range_proof.rs:133-159) and flatten the statements of a chunk
    auto sample_flatten = [&](ProveChunk& c, unsigned max_threads) {
      parallel_for(c.hi - c.lo, [&](size_t k) {
        const size_t b = c.lo + k;
        st[b].range.to_limbs(&c.range[k * kw], kw); st[b].ciphertext.to_limbs(&c.ct[k * 2 * kw], 2 * kw);
        st[b].secret_x.to_limbs(&c.x[k * kw], kw); st[b].secret_r.to_limbs(&c.r[k * kw], kw);
        const BigInt third = st[b].range.div_floor(BigInt(3)), two_thirds = BigInt(2) * third;   // range_proof.rs:133-134
        for (size_t i = 0; i < EF; i++) {
          BigInt a = BigInt::sample_range(third, two_thirds), cc = a - third;                     // :136-141
          if (BigInt::coin()) std::swap(a, cc);                                                    // :144-149
          a.to_limbs(&c.w1[(k * EF + i) * kw], kw); cc.to_limbs(&c.w2[(k * EF + i) * kw], kw);
          BigInt::sample_below(ek.n).to_limbs(&c.r1[(k * EF + i) * kw], kw);                      // :151-159
          BigInt::sample_below(ek.n).to_limbs(&c.r2[(k * EF + i) * kw], kw);
        }
      }, max_threads);
    };
    // stage B: the chunk's 256 Enc per proof, challenges and responses on the GPU (blocking: host buffers)
    auto gpu = [&](ProveChunk& c) {
      zkp_range_ni_proofs p{nb, (uint32_t)EF, c.hi - c.lo, 0, n.data(), c.range.data(), c.ct.data(), c.c1.data(), c.c2.data(), c.kind.data(), c.jj.data(),
                            c.rw1.data(), c.rr1.data(), c.rw2.data(), c.rr2.data()};
      zkp_range_ni_witness w{c.x.data(), c.r.data(), c.w1.data(), c.w2.data(), c.r1.data(), c.r2.data()};
      e.check(zkp_range_ni_prove_batch(e.ctx(), &p, &w, nullptr, nullptr, c.status.data(), 0), "zkp_range_ni_prove_batch");
    };
    // stage C1, under the GPU call: the proof objects of the chunk at their final shape — every BigInt allocated at full capacity and
    // zero-filled (the first touch of its pages).  Which rows will be Open and which Mask is not known yet: every row gets the four
    // fields of an Open response; `fill` moves two of them over for a Mask row and lets the other two go.
    // (more than a few threads only contend for the address space: measured round 4, 16 threads 674 ms against one 377 ms)
    auto prealloc = [&](ProveChunk& c, unsigned max_threads) {
      parallel_for(c.hi - c.lo, [&](size_t k) {
        RangeProofNi& o = out[c.lo + k];
        o.ek = ek; o.range = st[c.lo + k].range; o.ciphertext = st[c.lo + k].ciphertext; o.error_factor = EF;
        o.encrypted_pairs.c1.resize(EF); o.encrypted_pairs.c2.resize(EF); o.proof.responses.resize(EF);
        for (size_t i = 0; i < EF; i++) {
          o.encrypted_pairs.c1[i].l.resize(2 * kw); o.encrypted_pairs.c2[i].l.resize(2 * kw);
          Response& rs = o.proof.responses[i];
          rs.w1.l.resize(kw); rs.r1.l.resize(kw); rs.w2.l.resize(kw); rs.r2.l.resize(kw);
        }
      }, max_threads);
    };
    // stage C2, after the call: limbs into the memory that is already there (no allocation, no page fault: a copy at memory bandwidth)
    auto take = [](BigInt& dst, const uint32_t* p, size_t n) { while (n && p[n - 1] == 0) n--; dst.l.assign(p, p + n); dst.neg = false; };
    auto fill = [&](ProveChunk& c, unsigned max_threads) {
      parallel_for(c.hi - c.lo, [&](size_t k) {
        if (c.status[k] != 0) throw Panic("RangeProofNi::prove: malformed (the reference would panic)");
        RangeProofNi& o = out[c.lo + k];
        for (size_t i = 0; i < EF; i++) {
          const size_t t = k * EF + i;
          take(o.encrypted_pairs.c1[i], &c.c1[t * 2 * kw], 2 * kw);
          take(o.encrypted_pairs.c2[i], &c.c2[t * 2 * kw], 2 * kw);
          Response& rs = o.proof.responses[i];
          take(rs.w1, &c.rw1[t * kw], kw); take(rs.r1, &c.rr1[t * kw], kw);
          if (c.kind[t] == ZKP_RESP_OPEN) {
            rs.kind = Response::Open;
            take(rs.w2, &c.rw2[t * kw], kw); take(rs.r2, &c.rr2[t * kw], kw);
          } else {
            rs.kind = Response::Mask; rs.j = c.jj[t];
            rs.masked_x = std::move(rs.w1); rs.masked_r = std::move(rs.r1);
            rs.w1.l.clear(); rs.r1.l.clear(); rs.w2.l.clear(); rs.r2.l.clear();      // (zero; w2 / r2 keep their 2 x 256 B: a free() here goes to the
                                                                                      //  allocating thread's arena and 16 threads queue for its lock)
          }
        }
      }, max_threads);
    };

    std::vector<std::unique_ptr<ProveChunk>> ch(chunks);
    auto bounds = [&](size_t k) { return std::make_pair(B * k / chunks, B * (k + 1) / chunks); };
    StopWatch sw;
    ch[0].reset(new ProveChunk(bounds(0).first, bounds(0).second, kw, EF));
    sample_flatten(*ch[0], ~0u);
    tm.sample_flatten_ms = sw.lap();
    for (size_t k = 0; k < chunks; k++) {
      std::exception_ptr helper_error;
      std::thread helper([&, k] {
        try {
          const unsigned few = std::max(1u, std::min(4u, host_threads() / 2));
          prealloc(*ch[k], few);                    // the objects this call's outputs will go to
          if (k + 1 < chunks) {                     // next chunk's inputs, on a few threads: the GPU call only waits
            ch[k + 1].reset(new ProveChunk(bounds(k + 1).first, bounds(k + 1).second, kw, EF));
            sample_flatten(*ch[k + 1], std::max(1u, host_threads() / 2));
          }
          if (k > 0) { fill(*ch[k - 1], std::max(1u, host_threads() / 2)); ch[k - 1].reset(); }
        } catch (...) { helper_error = std::current_exception(); }
      });
      StopWatch g;
      try { gpu(*ch[k]); } catch (...) { helper.join(); throw; }
      tm.gpu_ms += g.lap();
      helper.join();
      if (helper_error) std::rethrow_exception(helper_error);
    }
    sw.lap();
    fill(*ch[chunks - 1], ~0u);
    tm.rebuild_ms = sw.lap();
    bool large = false;
    for (auto& c : ch) { c->wipe_secrets(); large = large || c->large(); }      // witnesses and nonces: gone before this call returns
    release_later(std::move(ch), large);
    return out;
  }
  static RangeProofNi prove(const EncryptionKey& ek, const BigInt& range, const BigInt& ciphertext, const BigInt& secret_x, const BigInt& secret_r) {
    return prove_batch(ek, {Statement{range, ciphertext, secret_x, secret_r}})[0];
  }

  // verify_self for many proofs sharing one key (range_proof_ni.rs:109-128).  Every field of a received proof is prover-chosen: any
  // size, either sign (curv::BigInt over GMP; the decimal serde takes a leading '-').  The fixed-width ABI carries kw / 2kw limbs of
  // non-negative values.  So each proof is classified on the host:
  //   * CANONICAL (every field non-negative and within its width, exactly error_factor rows stored): flattened into the SoA batch,
  //     ONE zkp_range_ni_verify_batch for all of them;
  //   * anything else goes through verify_general below: the reference's row logic verbatim on signed host integers, its modular
  //     exponentiations — the Enc of every row that needs one — still batched on the GPU.  Such a proof gets exactly the verdict,
  //     or the panic, the reference reaches for it (tests/golden/signed_cases.json);
  //   * the one thing that has no answer here is a KEY the engine cannot carry (not a positive odd integer of <= 4096 bits):
  //     Result::unsupported.
  static std::vector<Result> verify_batch(const EncryptionKey& ek, const std::vector<const RangeProofNi*>& proofs) {
    const size_t B = proofs.size();
    if (B == 0) return {};
    if (ek.n.is_negative() || !ek.n.is_odd() || ek.n.bit_length() > 4096 || ek.n.bit_length() < 2)
      return std::vector<Result>(B, Result::unsupported("RangeProofNi::verify: the key is not a positive odd integer of at most 4096 bits"));
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    // The batch-wide row count of the single-call path is the one MOST proofs carry (prove always writes SECURITY_PARAMETER, so that is
    // what wins in practice; ties go to it) — not that of proofs[0]: one odd proof at the head of a batch must not push the honest ones
    // off the fast path.  Proofs with another error_factor are verified correctly, by verify_general.
    size_t EF = SECURITY_PARAMETER;
    {
      std::map<size_t, size_t> votes;
      for (const RangeProofNi* p : proofs) votes[p->error_factor]++;
      size_t best = votes.count(EF) ? votes[EF] : 0;
      for (const auto& kv : votes) if (kv.second > best && kv.first > 0 && kv.first <= 4096) { best = kv.second; EF = kv.first; }
    }
    std::vector<size_t> fast, general;
    auto canonical = [&](const RangeProofNi& p) {
      if (p.error_factor != EF || p.proof.responses.size() != EF || p.encrypted_pairs.c1.size() != EF || p.encrypted_pairs.c2.size() != EF) return false;
      if (!p.range.fits_limbs(kw) || !p.ciphertext.fits_limbs(2 * kw)) return false;
      for (size_t i = 0; i < EF; i++) {
        const Response& rs = p.proof.responses[i];
        if (!p.encrypted_pairs.c1[i].fits_limbs(2 * kw) || !p.encrypted_pairs.c2[i].fits_limbs(2 * kw)) return false;
        if (rs.kind == Response::Open ? !(rs.w1.fits_limbs(kw) && rs.r1.fits_limbs(kw) && rs.w2.fits_limbs(kw) && rs.r2.fits_limbs(kw))
                                      : !(rs.masked_x.fits_limbs(kw) && rs.masked_r.fits_limbs(kw))) return false;
      }
      return true;
    };
    HostTiming& tm = last_host_timing();
    tm = HostTiming(); tm.proofs = B; tm.threads = host_threads();
    StopWatch sw;
    std::vector<char> canon(B);
    parallel_for(B, [&](size_t b) { canon[b] = canonical(*proofs[b]); });
    for (size_t b = 0; b < B; b++) (canon[b] ? fast : general).push_back(b);
    tm.general_proofs = general.size();
    std::vector<Result> out(B, Result(false));
    if (!fast.empty()) {
      // ONE GPU call by default.  The batch CAN run as a pipeline of chunks (ZKP_HOST_PIPELINE=N: while the GPU verifies chunk k, a helper
      // thread flattens chunk k + 1; ZKP_HOST_PIPELINE=uneven: a quarter, then the rest; 0 and 1: one call, as in prove_batch), but what a chunk hides is small — flattening 0.78 GB of
      // received proofs takes 13 ms on 16 threads now that the staging memory is huge-page backed (95 ms in round 4: first-touch faults) —
      // and every extra launch has a tail of its own.  B = 4096, one box, round 5: one call 3025 verifies/s, a quarter + the rest 2940,
      // 2 / 4 equal chunks 2938 / 2650.
      const size_t F = fast.size();
      struct VerifyChunk {
        size_t lo, hi;
        RawBuf<uint32_t> range, ct, c1, c2, rw1, rr1, rw2, rr2;
        RawBuf<uint8_t> kind, jj;
        std::vector<uint8_t> verdict;
        VerifyChunk(size_t lo_, size_t hi_, size_t kw, size_t EF)
            : lo(lo_), hi(hi_), range((hi_ - lo_) * kw), ct((hi_ - lo_) * 2 * kw), c1((hi_ - lo_) * EF * 2 * kw), c2((hi_ - lo_) * EF * 2 * kw), rw1((hi_ - lo_) * EF * kw),
              rr1((hi_ - lo_) * EF * kw), rw2((hi_ - lo_) * EF * kw), rr2((hi_ - lo_) * EF * kw), kind((hi_ - lo_) * EF), jj((hi_ - lo_) * EF), verdict(hi_ - lo_) {}
      };
      RawBuf<uint32_t> n(kw);
      ek.n.to_limbs(n.data(), kw);
      const char* pipe_env = std::getenv("ZKP_HOST_PIPELINE");
      const bool uneven = pipe_env && pipe_env[0] == 'u' && F >= 2048;
      const size_t chunks = uneven ? 2 : pipeline_chunks(F, 1);
      auto flatten = [&](VerifyChunk& c, unsigned max_threads) {
        parallel_for(c.hi - c.lo, [&](size_t k) {
          const RangeProofNi& p = *proofs[fast[c.lo + k]];
          p.range.to_limbs(&c.range[k * kw], kw); p.ciphertext.to_limbs(&c.ct[k * 2 * kw], 2 * kw);
          for (size_t i = 0; i < EF; i++) {
            const size_t t = k * EF + i;
            const Response& rs = p.proof.responses[i];
            p.encrypted_pairs.c1[i].to_limbs(&c.c1[t * 2 * kw], 2 * kw); p.encrypted_pairs.c2[i].to_limbs(&c.c2[t * 2 * kw], 2 * kw);
            if (rs.kind == Response::Open) {
              c.kind[t] = ZKP_RESP_OPEN; c.jj[t] = 0;
              rs.w1.to_limbs(&c.rw1[t * kw], kw); rs.r1.to_limbs(&c.rr1[t * kw], kw); rs.w2.to_limbs(&c.rw2[t * kw], kw); rs.r2.to_limbs(&c.rr2[t * kw], kw);
            } else {
              c.kind[t] = ZKP_RESP_MASK; c.jj[t] = rs.j;
              rs.masked_x.to_limbs(&c.rw1[t * kw], kw); rs.masked_r.to_limbs(&c.rr1[t * kw], kw);
              std::fill(&c.rw2[t * kw], &c.rw2[t * kw] + kw, 0u); std::fill(&c.rr2[t * kw], &c.rr2[t * kw] + kw, 0u);
            }
          }
        }, max_threads);
      };
      auto gpu = [&](VerifyChunk& c) {
        zkp_range_ni_proofs p{nb, (uint32_t)EF, c.hi - c.lo, 0, n.data(), c.range.data(), c.ct.data(), c.c1.data(), c.c2.data(), c.kind.data(), c.jj.data(),
                              c.rw1.data(), c.rr1.data(), c.rw2.data(), c.rr2.data()};
        e.check(zkp_range_ni_verify_batch(e.ctx(), &p, c.verdict.data(), 0), "zkp_range_ni_verify_batch");
      };
      std::vector<std::unique_ptr<VerifyChunk>> ch(chunks);
      auto bounds = [&](size_t k) {
        if (uneven) return k == 0 ? std::make_pair(size_t(0), F / 4) : std::make_pair(F / 4, F);
        return std::make_pair(F * k / chunks, F * (k + 1) / chunks);
      };
      ch[0].reset(new VerifyChunk(bounds(0).first, bounds(0).second, kw, EF));
      flatten(*ch[0], ~0u);
      tm.sample_flatten_ms = sw.lap();
      for (size_t k = 0; k < chunks; k++) {
        std::exception_ptr helper_error;
        std::thread helper([&, k] {
          try {
            if (k > 0) ch[k - 1].reset();                 // (the release of a quarter of a gigabyte off the critical path as well)
            if (k + 1 < chunks) {
              ch[k + 1].reset(new VerifyChunk(bounds(k + 1).first, bounds(k + 1).second, kw, EF));
              flatten(*ch[k + 1], std::max(1u, host_threads() / 2));
            }
          } catch (...) { helper_error = std::current_exception(); }
        });
        StopWatch g;
        try { gpu(*ch[k]); } catch (...) { helper.join(); throw; }
        tm.gpu_ms += g.lap();
        helper.join();
        if (helper_error) std::rethrow_exception(helper_error);
        const VerifyChunk& c = *ch[k];
        for (size_t f = c.lo; f < c.hi; f++) {
          const uint8_t v = c.verdict[f - c.lo];
          out[fast[f]] = v == ZKP_VERDICT_MALFORMED ? Result::panicked("RangeProofNi::verify: malformed proof (the reference would panic)") : Result(v == ZKP_VERDICT_ACCEPT);
        }
      }
      bool large = false;
      for (auto& c : ch) large = large || c->c1.pooled();
      release_later(std::move(ch), large);
      sw.lap();
    }
    if (!general.empty()) {
      std::vector<const RangeProofNi*> g;
      for (size_t b : general) g.push_back(proofs[b]);
      sw.lap();
      std::vector<Result> r = verify_general(ek, g);
      for (size_t k = 0; k < general.size(); k++) out[general[k]] = r[k];
      tm.general_ms = sw.lap();
    }
    return out;
  }

  // range_proof_ni.rs:109-128 -> range_proof.rs:254-355 on signed integers of any size (see verify_batch).  Operators as the
  // reference's BigInt has them: `%` truncated (:325,327), div_floor floored (:264), comparisons signed (:300-305,338), equality
  // exact (:293-298,335), to_bytes = magnitude in compute_digest (utils.rs:15-18, over EVERY stored c1 / c2).  Panics of the
  // reference — bits_of_e[i], responses[i], c1[i] / c2[i] past the end (:272-274,293,296,325,327) — are found per row, in the arm
  // that indexes, and win over any `false`.  All Enc of all these proofs: one batch on the GPU.
  // challenges: the verifier's own challenge bytes per proof (interactive RangeProof::verifier_output); null = Fiat-Shamir (above).
  static std::vector<Result> verify_general(const EncryptionKey& ek, const std::vector<const RangeProofNi*>& proofs,
                                            const std::vector<std::vector<uint8_t>>* challenges = nullptr) {
    struct Row { uint8_t what = 0; size_t enc0 = 0; BigInt expect0, expect1; bool flag = true; };   // what: 0 false, 1 open, 2 mask
    using EncList = std::vector<std::pair<const BigInt*, const BigInt*>>;
    struct Plan { std::vector<Row> rows; bool panic = false; EncList encs; size_t base = 0; };     // (enc0 of a row: relative to the plan's own list until the merge below)
    std::vector<Plan> plans(proofs.size());
    // one plan per proof, on the host threads: a transcript hash, a BigInt product and a Knuth division per Mask row — for a batch that an
    // adversarial sender pushed here that is the bulk of the host time
    parallel_for(proofs.size(), [&](size_t b) {
      const RangeProofNi& p = *proofs[b];
      Plan& pl = plans[b];
      EncList& encs = pl.encs;
      std::vector<uint8_t> ebytes;
      if (challenges) ebytes = (*challenges)[b];
      else {
        detail::Sha256 sh;
        sh.update(ek.n);
        for (const BigInt& v : p.encrypted_pairs.c1) sh.update(v);
        for (const BigInt& v : p.encrypted_pairs.c2) sh.update(v);
        ebytes = sh.finish().to_bytes();                               // leading zero bytes of the digest are dropped
      }
      const BigInt third = p.range.div_floor(BigInt(3)), two_thirds = BigInt(2) * third;      // :264-265
      for (size_t i = 0; i < p.error_factor && !pl.panic; i++) {
        if (i >= 8 * ebytes.size() || i >= p.proof.responses.size()) { pl.panic = true; break; }
        const bool ei = (ebytes[i / 8] >> (7 - i % 8)) & 1;
        const Response& rs = p.proof.responses[i];
        Row row;
        if (!ei && rs.kind == Response::Open) {                                              // :277-313
          if (i >= p.encrypted_pairs.c1.size() || i >= p.encrypted_pairs.c2.size()) { pl.panic = true; break; }
          row.what = 1; row.enc0 = encs.size();
          encs.push_back({&rs.w1, &rs.r1}); encs.push_back({&rs.w2, &rs.r2});
          row.expect0 = p.encrypted_pairs.c1[i]; row.expect1 = p.encrypted_pairs.c2[i];
          row.flag = (rs.w2 < third && rs.w1 > third && rs.w1 < two_thirds) || (rs.w1 < third && rs.w2 > third && rs.w2 < two_thirds);   // :300-305
        } else if (ei && rs.kind == Response::Mask) {                                         // :315-343
          const std::vector<BigInt>& cj = rs.j == 1 ? p.encrypted_pairs.c1 : p.encrypted_pairs.c2;   // any j != 1 selects c2, :324-328
          if (i >= cj.size()) { pl.panic = true; break; }
          row.what = 2; row.enc0 = encs.size();
          encs.push_back({&rs.masked_x, &rs.masked_r});
          row.expect0 = (cj[i] * p.ciphertext) % ek.nn;                                       // truncated, sign of the product
          row.flag = !(rs.masked_x < third || rs.masked_x > two_thirds);                      // :338
        }
        pl.rows.push_back(std::move(row));
      }
    });
    EncList encs;
    for (Plan& pl : plans) {
      pl.base = encs.size();
      if (pl.panic) continue;                                        // (a proof that panics needs no Enc: the panic wins over any `false`)
      encs.insert(encs.end(), pl.encs.begin(), pl.encs.end());
    }
    const std::vector<BigInt> E = Paillier::encrypt_with_chosen_randomness_batch(ek, encs);
    std::vector<Result> out;
    for (const Plan& pl : plans) {
      if (pl.panic) { out.push_back(Result::panicked("index out of bounds: the len is less than error_factor")); continue; }
      bool all = true;
      for (const Row& r : pl.rows) {
        bool res = r.what != 0 && r.flag;
        if (r.what == 1) res = res && E[pl.base + r.enc0] == r.expect0 && E[pl.base + r.enc0 + 1] == r.expect1;
        if (r.what == 2) res = res && E[pl.base + r.enc0] == r.expect0;
        all = all && res;
      }
      out.emplace_back(all);
    }
    return out;
  }

  // range_proof_ni.rs:84-107
  Result verify(const EncryptionKey& ek_, const BigInt& ciphertext_) const {
    if (!(ek_ == ek)) throw Panic("assertion failed: `(left == right)` ek");                  // :86
    if (ciphertext_ != ciphertext) throw Panic("assertion failed: `(left == right)` ciphertext");   // :88
    Result r = verify_batch(ek, {this})[0];
    (void)r.is_ok();                          // a single proof panics right here, as the reference does
    return r;
  }
  Result verify_self() const { Result r = verify_batch(ek, {this})[0]; (void)r.is_ok(); return r; }   // :109-128
};


// ------------------------------------------------------------------ interactive RangeProof (src/zkproofs/range_proof.rs:83-355)
struct DataRandomnessPairs { std::vector<BigInt> w1, w2, r1, r2; };   // range_proof.rs:41-47
struct ChallengeBits { std::vector<uint8_t> bytes; };                 // :49-50
struct Commitment { BigInt com; };                                    // :83
struct ChallengeRandomness { BigInt r; };                             // :100

struct RangeProof {
  static constexpr size_t STATISTICAL_ERROR_FACTOR = 40;   // range_proof.rs:30

  // :359-369
  static BigInt compute_digest(const std::vector<uint8_t>& bytes) { detail::Sha256 s; s.update(bytes.data(), bytes.size()); return s.finish(); }
  static BigInt get_paillier_commitment(const EncryptionKey& ek, const BigInt& x, const BigInt& r) { return Paillier::encrypt_with_chosen_randomness(ek, x, r); }

  struct VerifierCommit { Commitment com; ChallengeRandomness r; ChallengeBits e; };
  // :118-126 — e = STATISTICAL_ERROR_FACTOR random bits, com = Enc(SHA256(e), r)
  static VerifierCommit verifier_commit(const EncryptionKey& ek) {
    ChallengeBits e;
    e.bytes.resize(STATISTICAL_ERROR_FACTOR / 8);
    for (auto& b : e.bytes) b = (uint8_t)detail::ChaChaRng::local().next();
    BigInt r = BigInt::sample_below(ek.n);
    return {Commitment{get_paillier_commitment(ek, compute_digest(e.bytes), r)}, ChallengeRandomness{r}, e};
  }
  // :195-208
  static Result verify_commit(const EncryptionKey& ek, const Commitment& com, const ChallengeRandomness& r, const ChallengeBits& e) {
    return Result(com.com == get_paillier_commitment(ek, compute_digest(e.bytes), r.r));
  }

  // :128-193 — the 2 * error_factor encryptions are one launch
  static std::pair<EncryptedPairs, DataRandomnessPairs> generate_encrypted_pairs(const EncryptionKey& ek, const BigInt& range, size_t error_factor) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t EF = error_factor;
    DataRandomnessPairs d;
    const BigInt third = range.div_floor(BigInt(3)), two_thirds = BigInt(2) * third;   // :133-134
    for (size_t i = 0; i < EF; i++) {
      BigInt a = BigInt::sample_range(third, two_thirds), c = a - third;                // :136-141
      if (BigInt::coin()) std::swap(a, c);                                               // :144-149
      d.w1.push_back(a); d.w2.push_back(c);
      d.r1.push_back(BigInt::sample_below(ek.n)); d.r2.push_back(BigInt::sample_below(ek.n));   // :151-159
    }
    std::vector<uint32_t> n(kw), w1(EF * kw), w2(EF * kw), r1(EF * kw), r2(EF * kw), c1(EF * 2 * kw), c2(EF * 2 * kw);
    ek.n.to_limbs(n.data(), kw);
    for (size_t i = 0; i < EF; i++) {
      d.w1[i].to_limbs(&w1[i * kw], kw); d.w2[i].to_limbs(&w2[i * kw], kw); d.r1[i].to_limbs(&r1[i * kw], kw); d.r2[i].to_limbs(&r2[i * kw], kw);
    }
    zkp_range_ni_proofs p{};
    p.n_bits = nb; p.error_factor = (uint32_t)EF; p.batch = 1; p.n = n.data(); p.c1 = c1.data(); p.c2 = c2.data();
    zkp_range_ni_witness w{nullptr, nullptr, w1.data(), w2.data(), r1.data(), r2.data()};
    e.check(zkp_range_generate_encrypted_pairs_batch(e.ctx(), &p, &w, 0), "zkp_range_generate_encrypted_pairs_batch");
    EncryptedPairs ep;
    for (size_t i = 0; i < EF; i++) {
      ep.c1.push_back(BigInt::from_limbs(&c1[i * 2 * kw], 2 * kw)); ep.c2.push_back(BigInt::from_limbs(&c2[i * 2 * kw], 2 * kw));
    }
    return {ep, d};
  }

  static void pack_challenge(const ChallengeBits& e, uint8_t (&eb)[32], uint8_t& elen) {
    if (e.bytes.size() > 32) throw std::invalid_argument("ChallengeBits: the fixed-layout ABI carries at most 256 challenge bits");
    std::memset(eb, 0, 32);
    std::memcpy(eb, e.bytes.data(), e.bytes.size());
    elen = (uint8_t)e.bytes.size();
  }

  // :210-252
  static Proof generate_proof(const EncryptionKey& ek, const BigInt& secret_x, const BigInt& secret_r, const ChallengeBits& ch, const BigInt& range,
                              const DataRandomnessPairs& d, size_t error_factor) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t EF = error_factor;
    if (d.w1.size() < EF || d.w2.size() < EF || d.r1.size() < EF || d.r2.size() < EF) throw Panic("index out of bounds");   // data.w1[i]
    std::vector<uint32_t> n(kw), rg(kw), x(kw), r(kw), w1(EF * kw), w2(EF * kw), r1(EF * kw), r2(EF * kw);
    std::vector<uint32_t> rw1(EF * kw), rr1(EF * kw), rw2(EF * kw), rr2(EF * kw);
    std::vector<uint8_t> kind(EF), jj(EF);
    ek.n.to_limbs(n.data(), kw); range.to_limbs(rg.data(), kw); secret_x.to_limbs(x.data(), kw); secret_r.to_limbs(r.data(), kw);
    for (size_t i = 0; i < EF; i++) {
      d.w1[i].to_limbs(&w1[i * kw], kw); d.w2[i].to_limbs(&w2[i * kw], kw); d.r1[i].to_limbs(&r1[i * kw], kw); d.r2[i].to_limbs(&r2[i * kw], kw);
    }
    uint8_t eb[32], elen, status = 0;
    pack_challenge(ch, eb, elen);
    zkp_range_ni_proofs p{};
    p.n_bits = nb; p.error_factor = (uint32_t)EF; p.batch = 1; p.n = n.data(); p.range = rg.data();
    p.resp_kind = kind.data(); p.resp_j = jj.data(); p.resp_w1 = rw1.data(); p.resp_r1 = rr1.data(); p.resp_w2 = rw2.data(); p.resp_r2 = rr2.data();
    zkp_range_ni_witness w{x.data(), r.data(), w1.data(), w2.data(), r1.data(), r2.data()};
    e.check(zkp_range_generate_proof_batch(e.ctx(), &p, &w, eb, &elen, &status, 0), "zkp_range_generate_proof_batch");
    if (status != 0) throw Panic("RangeProof::generate_proof: malformed (the reference would panic: bits_of_e[i])");
    Proof out;
    for (size_t t = 0; t < EF; t++) {
      Response rs;
      if (kind[t] == ZKP_RESP_OPEN) {
        rs.kind = Response::Open;
        rs.w1 = BigInt::from_limbs(&rw1[t * kw], kw); rs.r1 = BigInt::from_limbs(&rr1[t * kw], kw);
        rs.w2 = BigInt::from_limbs(&rw2[t * kw], kw); rs.r2 = BigInt::from_limbs(&rr2[t * kw], kw);
      } else {
        rs.kind = Response::Mask; rs.j = jj[t];
        rs.masked_x = BigInt::from_limbs(&rw1[t * kw], kw); rs.masked_r = BigInt::from_limbs(&rr1[t * kw], kw);
      }
      out.responses.push_back(std::move(rs));
    }
    return out;
  }

  // :254-355
  static Result verifier_output(const EncryptionKey& ek, const ChallengeBits& ch, const EncryptedPairs& ep, const Proof& proof, const BigInt& range,
                                const BigInt& cipher_x, size_t error_factor) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t EF = error_factor;
    // values the fixed-width call cannot carry (negative, over-wide), short vectors, a challenge of more than 32 bytes: the reference's
    // row logic on signed host integers with the verifier's own challenge (RangeProofNi::verify_general), Enc still on the GPU
    bool canonical = proof.responses.size() >= EF && ep.c1.size() >= EF && ep.c2.size() >= EF && ch.bytes.size() <= 32 && range.fits_limbs(kw) && cipher_x.fits_limbs(2 * kw);
    for (size_t t = 0; canonical && t < EF; t++) {
      const Response& rs = proof.responses[t];
      canonical = ep.c1[t].fits_limbs(2 * kw) && ep.c2[t].fits_limbs(2 * kw) &&
                  (rs.kind == Response::Open ? rs.w1.fits_limbs(kw) && rs.r1.fits_limbs(kw) && rs.w2.fits_limbs(kw) && rs.r2.fits_limbs(kw)
                                             : rs.masked_x.fits_limbs(kw) && rs.masked_r.fits_limbs(kw));
    }
    if (!canonical) {
      RangeProofNi tmp;
      tmp.ek = ek; tmp.range = range; tmp.ciphertext = cipher_x; tmp.encrypted_pairs = ep; tmp.proof = proof; tmp.error_factor = EF;
      const std::vector<std::vector<uint8_t>> chal{ch.bytes};
      Result r = RangeProofNi::verify_general(ek, {&tmp}, &chal)[0];
      (void)r.is_ok();                          // a panic of the reference is thrown here
      return r;
    }
    std::vector<uint32_t> n(kw), rg(kw), ct(2 * kw), c1(EF * 2 * kw), c2(EF * 2 * kw), rw1(EF * kw), rr1(EF * kw), rw2(EF * kw), rr2(EF * kw);
    std::vector<uint8_t> kind(EF), jj(EF);
    ek.n.to_limbs(n.data(), kw); range.to_limbs(rg.data(), kw); cipher_x.to_limbs(ct.data(), 2 * kw);
    for (size_t t = 0; t < EF; t++) {
      ep.c1[t].to_limbs(&c1[t * 2 * kw], 2 * kw); ep.c2[t].to_limbs(&c2[t * 2 * kw], 2 * kw);
      const Response& rs = proof.responses[t];
      if (rs.kind == Response::Open) {
        kind[t] = ZKP_RESP_OPEN;
        rs.w1.to_limbs(&rw1[t * kw], kw); rs.r1.to_limbs(&rr1[t * kw], kw); rs.w2.to_limbs(&rw2[t * kw], kw); rs.r2.to_limbs(&rr2[t * kw], kw);
      } else {
        kind[t] = ZKP_RESP_MASK; jj[t] = rs.j;
        rs.masked_x.to_limbs(&rw1[t * kw], kw); rs.masked_r.to_limbs(&rr1[t * kw], kw);
      }
    }
    uint8_t eb[32], elen, verdict = 0;
    pack_challenge(ch, eb, elen);
    zkp_range_ni_proofs p{nb, (uint32_t)EF, 1, 0, n.data(), rg.data(), ct.data(), c1.data(), c2.data(), kind.data(), jj.data(),
                          rw1.data(), rr1.data(), rw2.data(), rr2.data()};
    e.check(zkp_range_verifier_output_batch(e.ctx(), &p, eb, &elen, &verdict, 0), "zkp_range_verifier_output_batch");
    if (verdict == ZKP_VERDICT_MALFORMED) throw Panic("RangeProof::verifier_output: malformed proof (the reference would panic)");
    return Result(verdict == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ NiCorrectKeyProof
static const uint8_t SALT_STRING[4] = {75, 90, 101, 110};   // correct_key_ni.rs:28

class NiCorrectKeyProof {
 public:
  static constexpr size_t M2 = ZKP_CORRECT_KEY_M2, DIGEST_SIZE = 256;   // correct_key_ni.rs:29-30
  std::vector<BigInt> sigma_vec;

  // correct_key_ni.rs:105-117
  static BigInt mask_generation(size_t out_length, const BigInt& seed) {
    const size_t msklen = out_length / DIGEST_SIZE + 1;
    BigInt acc;
    for (size_t j = 0; j < msklen; j++) { BigInt jj(j); acc = acc + detail::compute_digest({&seed, &jj}).shl(j * DIGEST_SIZE); }
    return acc;
  }
  // correct_key_ni.rs:42-71 — prover side (needs the secret key).  The n-th roots are
  // rho^(n^-1 mod phi) mod n ([upstream] extract_nroot), computed by ONE batched GPU modexp.
  static NiCorrectKeyProof proof(const DecryptionKey& dk, const uint8_t* salt = SALT_STRING, size_t salt_len = 4) {
    Engine& e = Engine::instance();
    const BigInt n = dk.p * dk.q;
    const BigInt phi = (dk.p - BigInt::one()) * (dk.q - BigInt::one());
    const BigInt d = BigInt::mod_inv(n, phi);
    const BigInt salt_bn = [&] { BigInt s = BigInt::from_bytes(salt, salt_len); return detail::compute_digest({&s}); }();
    const uint32_t nb = n.bit_length() <= 2048 ? 2048 : 4096, kw = nb / 32;   // modexp kernel widths
    std::vector<uint32_t> base(M2 * kw), ex(kw), mod(kw), out(M2 * kw);
    for (size_t i = 0; i < M2; i++) {
      BigInt ii(i);
      BigInt seed = detail::compute_digest({&n, &salt_bn, &ii});
      (mask_generation(n.bit_length(), seed) % n).to_limbs(&base[i * kw], kw);
    }
    d.to_limbs(ex.data(), kw); n.to_limbs(mod.data(), kw);
    e.check(zkp_modexp_batch(e.ctx(), nb, nb, M2, base.data(), ex.data(), 0, mod.data(), 0, out.data(), 0), "zkp_modexp_batch");
    NiCorrectKeyProof p;
    for (size_t i = 0; i < M2; i++) p.sigma_vec.push_back(BigInt::from_limbs(&out[i * kw], kw));
    return p;
  }
  // correct_key_ni.rs:73-100 for many (key, proof) pairs in ONE zkp_correct_key_ni_verify_batch (keys of one kernel width per call; the
  // widths are grouped here).  sigma_i is prover-chosen: any size, either sign — the reference only uses it as mod_pow(sigma_i, n, n),
  // which depends on sigma_i mod n (floored: mpz_powm), so a non-canonical root is reduced on the host and gets the reference's verdict;
  // a zero key and fewer than 11 roots are the reference's panics (:82-85, :92), checked in its order; an even key then fails the primorial gcd test whatever the roots are (Err); a key
  // the engine cannot carry (not positive, or wider than 4096 bits) is Result::unsupported.
  static std::vector<Result> verify_batch(const std::vector<std::pair<const EncryptionKey*, const NiCorrectKeyProof*>>& items, const uint8_t* salt = SALT_STRING,
                                          size_t salt_len = 4) {
    Engine& e = Engine::instance();
    std::vector<Result> out(items.size(), Result(false));
    for (uint32_t nb : {1024u, 2048u, 4096u}) {
      const uint32_t kw = nb / 32;
      std::vector<size_t> idx;
      for (size_t k = 0; k < items.size(); k++) {
        const BigInt& n = items[k].first->n;
        // in the reference's order: rho_i = mask_generation(..) % n panics on n = 0 (:82-85); sigma_vec[i] for i < 11 panics on a short vector
        // (:90-93) — both BEFORE anything is compared; only then does an even key fail the primorial gcd test whatever the roots are (:87-88,95)
        if (n.is_zero()) { out[k] = Result::panicked("attempt to calculate the remainder with a divisor of zero"); continue; }
        if (items[k].second->sigma_vec.size() < M2) { out[k] = Result::panicked("index out of bounds: sigma_vec"); continue; }
        if (!n.is_negative() && !n.is_odd()) continue;                    // an even key: Err(IncorrectProof); `out` holds that already
        const bool supported = !n.is_negative() && n.is_odd() && n.bit_length() >= 2 && n.bit_length() <= 4096;
        if (!supported) { if (nb == 1024) out[k] = Result::unsupported("NiCorrectKeyProof::verify: the key is not a positive integer of at most 4096 bits"); continue; }
        if (width_for(n) != nb) continue;
        idx.push_back(k);
      }
      if (idx.empty()) continue;
      std::vector<uint32_t> n(idx.size() * kw), sg(idx.size() * M2 * kw);
      std::vector<uint8_t> v(idx.size(), 9);
      for (size_t q = 0; q < idx.size(); q++) {
        const EncryptionKey& ek = *items[idx[q]].first;
        ek.n.to_limbs(&n[q * kw], kw);
        for (size_t i = 0; i < M2; i++) {
          const BigInt& s = items[idx[q]].second->sigma_vec[i];
          (s.fits_limbs(kw) ? s : s.modulus(ek.n)).to_limbs(&sg[(q * M2 + i) * kw], kw);
        }
      }
      e.check(zkp_correct_key_ni_verify_batch(e.ctx(), nb, idx.size(), n.data(), sg.data(), salt, (uint32_t)salt_len, v.data(), 0), "zkp_correct_key_ni_verify_batch");
      for (size_t q = 0; q < idx.size(); q++) out[idx[q]] = Result(v[q] == ZKP_VERDICT_ACCEPT);
    }
    return out;
  }
  Result verify(const EncryptionKey& ek, const uint8_t* salt = SALT_STRING, size_t salt_len = 4) const {
    Result r = verify_batch({{&ek, this}}, salt, salt_len)[0];
    (void)r.is_ok();                          // a single proof panics right here, as the reference does
    return r;
  }
};

// ------------------------------------------------------------------ interactive CorrectKey (src/zkproofs/correct_key.rs:28-183)
// the same 2048-bit modexp shape x40, composed on the host from batched GPU calls.
struct Challenge { std::vector<BigInt> sn; BigInt e; std::vector<BigInt> z; };   // :29-39
struct VerificationAid { BigInt s_digest; };                                       // :41-45
struct CorrectKeyProof { BigInt s_digest; };                                       // :47-51
enum class CorrectKeyProveError { SniNotCoprimeWithN, ZiNotCoprimeWithN, RniNotCoprimeWithN, EWasntComputedCorrectly };   // :174-183

struct CorrectKey {
  static constexpr size_t STATISTICAL_ERROR_FACTOR = 40;   // :26
  static BigInt digest_chain(const BigInt* first, const std::vector<BigInt>& a, const std::vector<BigInt>* b = nullptr) {
    detail::Sha256 s;
    if (first) s.update(*first);
    for (auto& v : a) s.update(v);
    if (b) for (auto& v : *b) s.update(v);
    return s.finish();
  }
  // :64-102
  static std::pair<Challenge, VerificationAid> challenge(const EncryptionKey& ek) {
    std::vector<BigInt> s, r;
    for (size_t i = 0; i < STATISTICAL_ERROR_FACTOR; i++) { s.push_back(BigInt::sample_below(ek.n)); r.push_back(BigInt::sample_below(ek.n)); }
    return challenge_with(ek, s, r);
  }
  // the same with the values the reference samples (:67-70, :80-83) supplied by the caller (parity tests)
  static std::pair<Challenge, VerificationAid> challenge_with(const EncryptionKey& ek, const std::vector<BigInt>& s, const std::vector<BigInt>& r) {
    if (s.size() != r.size()) throw std::invalid_argument("CorrectKey::challenge_with: |s| != |r|");
    const size_t K = s.size();
    std::vector<BigInt> both = s; both.insert(both.end(), r.begin(), r.end());
    std::vector<BigInt> pw = mod_pow_batch(both, {ek.n}, ek.n);                               // sn, rn  (:73-76, :86-89)
    std::vector<BigInt> sn(pw.begin(), pw.begin() + K), rn(pw.begin() + K, pw.end());
    BigInt e = digest_chain(&ek.n, sn, &rn);                                                  // :91
    std::vector<BigInt> se = mod_pow_batch(s, {e}, ek.n);                                     // s_i^e  (:96)
    std::vector<BigInt> z;
    for (size_t i = 0; i < K; i++) z.push_back((r[i] * se[i]) % ek.n);
    return {Challenge{sn, e, z}, VerificationAid{digest_chain(nullptr, s)}};                  // :100-102
  }
  // :104-162 — returns the proof or the error
  struct ProveResult {
    bool ok; CorrectKeyProof proof; CorrectKeyProveError err;
    bool is_ok() const { return ok; }
    bool is_err() const { return !ok; }
    const CorrectKeyProof& unwrap() const { if (!ok) throw Panic("called `Result::unwrap()` on an `Err` value"); return proof; }
  };
  static ProveResult prove(const DecryptionKey& dk, const Challenge& ch) {
    const BigInt dk_n = dk.q * dk.p, one = BigInt::one();
    auto fail = [](CorrectKeyProveError e) { return ProveResult{false, CorrectKeyProof{}, e}; };
    for (auto& v : ch.sn) if (BigInt::gcd(dk_n, v) != one) return fail(CorrectKeyProveError::SniNotCoprimeWithN);   // :110-116
    for (auto& v : ch.z) if (BigInt::gcd(dk_n, v) != one) return fail(CorrectKeyProveError::ZiNotCoprimeWithN);     // :119-125
    const BigInt phi = (dk.q - one) * (dk.p - one);
    const BigInt phimine = phi - (ch.e % phi);                                                                       // :130
    std::vector<BigInt> zn = mod_pow_batch(ch.z, {dk_n}, dk_n), snphi = mod_pow_batch(ch.sn, {phimine}, dk_n);       // :135-137
    std::vector<BigInt> rn;
    for (size_t i = 0; i < ch.z.size(); i++) rn.push_back((zn[i] * snphi[i]) % dk_n);
    for (auto& v : rn) if (BigInt::gcd(dk_n, v) != one) return fail(CorrectKeyProveError::RniNotCoprimeWithN);       // :143-148
    if (ch.e != digest_chain(&dk_n, ch.sn, &rn)) return fail(CorrectKeyProveError::EWasntComputedCorrectly);         // :151-156
    // s_digest = H(extract_nroot(dk, sn_i)...): sn_i^(n^-1 mod phi) mod n  ([upstream] extract_nroot)              // :159
    std::vector<BigInt> roots = mod_pow_batch(ch.sn, {BigInt::mod_inv(dk_n, phi)}, dk_n);
    return ProveResult{true, CorrectKeyProof{digest_chain(nullptr, roots)}, CorrectKeyProveError::EWasntComputedCorrectly};
  }
  // :164-171
  static Result verify(const CorrectKeyProof& proof, const VerificationAid& va) { return Result(proof.s_digest == va.s_digest); }
};

// ------------------------------------------------------------------ CompositeDLogProof
struct DLogStatement { BigInt N, g, ni; };   // wi_dlog_proof.rs:38-43

class CompositeDLogProof {
 public:
  static constexpr uint32_t Y_BITS = 768;     // y = r + e*s < 2^513 for honest provers
  BigInt x, y;
  // wi_dlog_proof.rs:46-65
  static CompositeDLogProof prove(const DLogStatement& st, const BigInt& secret) {
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.N), kw = nb / 32;
    const BigInt r = BigInt::sample_below(BigInt::pow2(512));     // K + K' + SAMPLE_S = 512, :53-54
    std::vector<uint32_t> N(kw), g(kw), ni(kw), s(8), rr(16), x(kw), y(Y_BITS / 32);
    st.N.to_limbs(N.data(), kw); st.g.to_limbs(g.data(), kw); st.ni.to_limbs(ni.data(), kw); secret.to_limbs(s.data(), 8); r.to_limbs(rr.data(), 16);
    e.check(zkp_dlog_prove_batch(e.ctx(), nb, Y_BITS, 1, N.data(), g.data(), ni.data(), s.data(), rr.data(), x.data(), y.data(), 0), "zkp_dlog_prove_batch");
    return CompositeDLogProof{BigInt::from_limbs(x.data(), kw), BigInt::from_limbs(y.data(), Y_BITS / 32)};
  }
  // base^exp mod N for a non-negative exponent of ANY length (a prover-chosen y is not bounded): exponents wider than the modulus
  // width go through the GPU in chunks, acc = acc^(2^c) * base^chunk, most significant chunk first
  static BigInt mod_pow_wide(const BigInt& base, const BigInt& exp, const BigInt& N) {
    const size_t mb = N.bit_length() <= 2048 ? 2048 : N.bit_length() <= 4096 ? 4096 : 8192, chunk = mb - 32;
    if (exp.bit_length() <= mb) return mod_pow(base, exp, N);
    BigInt acc = BigInt::one();
    for (size_t hi = (exp.bit_length() + chunk - 1) / chunk; hi-- > 0;) {
      BigInt part;
      for (size_t b = 0; b < chunk; b++) if (exp.bit(hi * chunk + b)) part = part + BigInt::pow2(b);
      acc = (mod_pow(acc, BigInt::pow2(chunk), N) * mod_pow(base, part, N)).modulus(N);
    }
    return acc;
  }
  // wi_dlog_proof.rs:67-91.  Honest shapes (x, g, ni < N, 0 <= y < 2^768) go to zkp_dlog_verify_batch.  Anything else a received proof
  // or statement may hold — an over-wide y (y = r + e s is not bounded by the reference), g / ni / x that are negative or >= N — is
  // evaluated as the reference does it: the pre-checks and the hash (over the RAW values' magnitudes) on the host, the two
  // exponentiations on the GPU (mod_pow reduces its base into [0, N), as mpz_powm does), mod_mul and the compare on the host.
  // Not carried: a modulus the engine cannot take (even, or wider than 4096 bits) and a NEGATIVE y ([upstream] mod_pow with a negative
  // exponent: inverse or panic, not recalled with confidence): Result::unsupported.
  Result verify(const DLogStatement& st) const {
    Engine& e = Engine::instance();
    if (st.N.is_negative() || st.N <= BigInt::pow2(128)) throw Panic("assertion failed: statement.N > BigInt::from(2).pow(K as u32)");   // :69
    if (!st.N.is_odd() || st.N.bit_length() > 4096) return Result::unsupported("CompositeDLogProof::verify: the modulus is even or wider than 4096 bits");
    const uint32_t nb = width_for(st.N), kw = nb / 32;
    const bool canonical = !st.g.is_negative() && st.g < st.N && !st.ni.is_negative() && st.ni < st.N && x.fits_limbs(kw) && y.fits_limbs(Y_BITS / 32);
    if (!canonical) {
      if (y.is_negative()) return Result::unsupported("CompositeDLogProof::verify: negative response y");
      if (BigInt::gcd(st.g, st.N) != BigInt::one() || BigInt::gcd(st.ni, st.N) != BigInt::one())
        throw Panic("assertion failed: `(left == right)` gcd(g, N) / gcd(ni, N)");                                      // :72-73
      const BigInt ee = detail::compute_digest({&x, &st.g, &st.N, &st.ni});                                               // :75-80
      const BigInt ni_e = mod_pow(st.ni, ee, st.N), g_y = mod_pow_wide(st.g, y, st.N);                                    // :81-82
      return Result(x == (g_y * ni_e).modulus(st.N));                                                                     // :83-90
    }
    std::vector<uint32_t> N(kw), g(kw), ni(kw), xx(kw), yy(Y_BITS / 32);
    st.N.to_limbs(N.data(), kw); st.g.to_limbs(g.data(), kw); st.ni.to_limbs(ni.data(), kw); x.to_limbs(xx.data(), kw); y.to_limbs(yy.data(), Y_BITS / 32);
    uint8_t v = 9;
    e.check(zkp_dlog_verify_batch(e.ctx(), nb, Y_BITS, 1, N.data(), g.data(), ni.data(), xx.data(), yy.data(), &v, 0), "zkp_dlog_verify_batch");
    if (v == ZKP_VERDICT_MALFORMED) throw Panic("assertion failed in CompositeDLogProof::verify (N > 2^128, gcd(g,N) = gcd(ni,N) = 1)");   // :69,72,73
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ ZeroProof (src/zkproofs/zero_enc_proof.rs:26-95)
struct ZeroWitness { BigInt r; };
struct ZeroStatement { EncryptionKey ek; BigInt c; };
class ZeroProof {
 public:
  BigInt z, a;
  static ZeroProof prove(const ZeroWitness& w, const ZeroStatement& st) {   // :44-64
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32;
    const BigInt r_prime = BigInt::sample_below(st.ek.n);                     // :45
    std::vector<uint32_t> n(kw), c(2 * kw), r(kw), rp(kw), z(2 * kw), a(2 * kw);
    st.ek.n.to_limbs(n.data(), kw); st.c.to_limbs(c.data(), 2 * kw); (w.r % st.ek.nn).to_limbs(r.data(), kw); r_prime.to_limbs(rp.data(), kw);
    e.check(zkp_zero_proof_prove_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), r.data(), rp.data(), z.data(), a.data(), 0), "zkp_zero_proof_prove_batch");
    return ZeroProof{BigInt::from_limbs(z.data(), 2 * kw), BigInt::from_limbs(a.data(), 2 * kw)};
  }
  Result verify(const ZeroStatement& st) const {                             // :66-94
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32;
    std::vector<uint32_t> n(kw), c(2 * kw), zz(2 * kw), aa(2 * kw);
    st.ek.n.to_limbs(n.data(), kw); st.c.to_limbs(c.data(), 2 * kw); z.to_limbs(zz.data(), 2 * kw); a.to_limbs(aa.data(), 2 * kw);
    uint8_t v = 9;
    e.check(zkp_zero_proof_verify_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), zz.data(), aa.data(), &v, 0), "zkp_zero_proof_verify_batch");
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ CiphertextProof (src/zkproofs/correct_ciphertext.rs:23-98)
struct CiphertextWitness { BigInt x, r; };
struct CiphertextStatement { EncryptionKey ek; BigInt c; };
class CiphertextProof {
 public:
  BigInt z1, z2, c_prime;
  static CiphertextProof prove(const CiphertextWitness& w, const CiphertextStatement& st) {   // :42-64
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32, z1w = kw + ZKP_Z1_EXTRA_LIMBS;
    const BigInt x_prime = BigInt::sample_below(st.ek.n), r_prime = BigInt::sample_below(st.ek.n);   // :43-44
    std::vector<uint32_t> n(kw), c(2 * kw), x(kw), r(kw), xp(kw), rp(kw), o1(z1w), o2(2 * kw), oc(2 * kw);
    st.ek.n.to_limbs(n.data(), kw); st.c.to_limbs(c.data(), 2 * kw); w.x.to_limbs(x.data(), kw); w.r.to_limbs(r.data(), kw);
    x_prime.to_limbs(xp.data(), kw); r_prime.to_limbs(rp.data(), kw);
    e.check(zkp_ciphertext_proof_prove_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), x.data(), r.data(), xp.data(), rp.data(), o1.data(), o2.data(), oc.data(), 0),
            "zkp_ciphertext_proof_prove_batch");
    return CiphertextProof{BigInt::from_limbs(o1.data(), z1w), BigInt::from_limbs(o2.data(), 2 * kw), BigInt::from_limbs(oc.data(), 2 * kw)};
  }
  Result verify(const CiphertextStatement& st) const {                                        // :66-97
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32, z1w = kw + ZKP_Z1_EXTRA_LIMBS;
    std::vector<uint32_t> n(kw), c(2 * kw), a1(z1w), a2(2 * kw), ac(2 * kw);
    st.ek.n.to_limbs(n.data(), kw); st.c.to_limbs(c.data(), 2 * kw); z1.to_limbs(a1.data(), z1w); z2.to_limbs(a2.data(), 2 * kw); c_prime.to_limbs(ac.data(), 2 * kw);
    uint8_t v = 9;
    e.check(zkp_ciphertext_proof_verify_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), a1.data(), a2.data(), ac.data(), &v, 0), "zkp_ciphertext_proof_verify_batch");
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ VerlinProof (src/zkproofs/verlin_proof.rs:35-165)
struct VerlinWitness { BigInt x, x_prime, x_double_prime, r_x; };
struct VerlinStatement { EncryptionKey ek; BigInt c, c_prime, phi_x; };

// gen_phi (verlin_proof.rs:138-165) = c^y * c'^y' * Enc(y'', r_y) mod n^2, from three batched GPU calls
inline BigInt gen_phi(const EncryptionKey& ek, const BigInt& c, const BigInt& c_prime, const BigInt& y, const BigInt& y_prime,
                      const BigInt& y_double_prime, const BigInt& r_y) {
  std::vector<BigInt> p = mod_pow_batch({c, c_prime}, {y, y_prime}, ek.nn);
  BigInt e3 = Paillier::encrypt_with_chosen_randomness(ek, y_double_prime % ek.n, r_y);   // (1 + m n) mod n^2 depends on m mod n only; r_y < n for honest callers
  return ((p[0] * p[1]) % ek.nn) * e3 % ek.nn;
}

class VerlinProof {
 public:
  BigInt phi_a, z, z_prime, z_double_prime, r_z;
  static VerlinProof prove(const VerlinWitness& w, const VerlinStatement& st) {   // :60-99
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32, zw = kw + ZKP_Z1_EXTRA_LIMBS;
    const BigInt a = BigInt::sample_below(st.ek.n), ap = BigInt::sample_below(st.ek.n), app = BigInt::sample_below(st.ek.n);   // :61-63
    BigInt r_a = BigInt::sample_below(st.ek.n);
    while (BigInt::gcd(r_a, st.ek.n) != BigInt::one()) r_a = BigInt::sample_below(st.ek.n);                                   // :64-67
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(st.ek.n, kw), c = L1(st.c, 2 * kw), cp = L1(st.c_prime, 2 * kw), phx = L1(st.phi_x, 2 * kw);
    auto x = L1(w.x, kw), xp = L1(w.x_prime, kw), xpp = L1(w.x_double_prime, kw), rx = L1(w.r_x, kw);
    auto va = L1(a, kw), vap = L1(ap, kw), vapp = L1(app, kw), vra = L1(r_a, kw);
    std::vector<uint32_t> pa(2 * kw), z(zw), zp(zw), zpp(zw), rz(2 * kw);
    e.check(zkp_verlin_proof_prove_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), cp.data(), phx.data(), x.data(), xp.data(), xpp.data(), rx.data(),
                                         va.data(), vap.data(), vapp.data(), vra.data(), pa.data(), z.data(), zp.data(), zpp.data(), rz.data(), 0),
            "zkp_verlin_proof_prove_batch");
    return VerlinProof{BigInt::from_limbs(pa.data(), 2 * kw), BigInt::from_limbs(z.data(), zw), BigInt::from_limbs(zp.data(), zw),
                       BigInt::from_limbs(zpp.data(), zw), BigInt::from_limbs(rz.data(), 2 * kw)};
  }
  Result verify(const VerlinStatement& st) const {                                // :101-135
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32, zw = kw + ZKP_Z1_EXTRA_LIMBS;
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(st.ek.n, kw), c = L1(st.c, 2 * kw), cp = L1(st.c_prime, 2 * kw), phx = L1(st.phi_x, 2 * kw), pa = L1(phi_a, 2 * kw);
    auto vz = L1(z, zw), vzp = L1(z_prime, zw), vzpp = L1(z_double_prime, zw), vrz = L1(r_z, 2 * kw);
    uint8_t v = 9;
    e.check(zkp_verlin_proof_verify_batch(e.ctx(), nb, 1, n.data(), 0, c.data(), cp.data(), phx.data(), pa.data(), vz.data(), vzp.data(), vzpp.data(),
                                          vrz.data(), &v, 0), "zkp_verlin_proof_verify_batch");
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ BigInt::mod_inv (batched on the GPU)
// None (std::nullopt-like: empty vector entry flagged) when no inverse exists; throws std::domain_error outside the ABI's domain
struct ModInvResult { bool some; BigInt value; };
inline std::vector<ModInvResult> mod_inv_batch(const std::vector<BigInt>& a, const BigInt& modulus) {
  Engine& e = Engine::instance();
  uint32_t mb = width_for(modulus); if (mb < 2048) mb = 2048;
  const size_t L = mb / 32, cnt = a.size();
  std::vector<uint32_t> av(cnt * L), mv(L), out(cnt * L);
  std::vector<uint8_t> stv(cnt);
  modulus.to_limbs(mv.data(), L);
  for (size_t i = 0; i < cnt; i++) (a[i] % modulus).to_limbs(&av[i * L], L);
  if (cnt) e.check(zkp_modinv_batch(e.ctx(), mb, cnt, av.data(), mv.data(), 0, out.data(), stv.data(), 0), "zkp_modinv_batch");
  std::vector<ModInvResult> r;
  for (size_t i = 0; i < cnt; i++) {
    if (stv[i] == ZKP_INV_DOMAIN) throw std::domain_error("mod_inv: even or trivial modulus");
    r.push_back({stv[i] == ZKP_INV_OK, BigInt::from_limbs(&out[i * L], L)});
  }
  return r;
}

// ------------------------------------------------------------------ MulProof (src/zkproofs/multiplication_proof.rs)
struct MulWitness { BigInt a, b, c, r_a, r_b, r_c; };          // :42-50
struct MulStatement { EncryptionKey ek; BigInt e_a, e_b, e_c; };   // :52-58
inline BigInt sample_paillier_random(const BigInt& modulo) {    // :148-154
  BigInt r = BigInt::sample_below(modulo);
  while (BigInt::gcd(r, modulo) != BigInt::one()) r = BigInt::sample_below(modulo);
  return r;
}
class MulProof {
 public:
  BigInt f, z1, z2, e_d, e_db;
  static MulProof prove(const MulWitness& w, const MulStatement& st) {   // :60-104
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32;
    const BigInt d = BigInt::sample_below(st.ek.n), r_d = sample_paillier_random(st.ek.n);   // :61-62
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(st.ek.n, kw), ea = L1(st.e_a, 2 * kw), eb = L1(st.e_b, 2 * kw), ec = L1(st.e_c, 2 * kw);
    auto a = L1(w.a, kw), b = L1(w.b, kw), ra = L1(w.r_a, kw), rb = L1(w.r_b, kw), rc = L1(w.r_c, kw), vd = L1(d, kw), vrd = L1(r_d, kw);
    std::vector<uint32_t> f(kw), z1(2 * kw), z2(2 * kw), ed(2 * kw), edb(2 * kw);
    uint8_t status = 9;
    e.check(zkp_mul_proof_prove_batch(e.ctx(), nb, 1, n.data(), 0, ea.data(), eb.data(), ec.data(), a.data(), b.data(), ra.data(), rb.data(), rc.data(),
                                      vd.data(), vrd.data(), f.data(), z1.data(), z2.data(), ed.data(), edb.data(), &status, 0), "zkp_mul_proof_prove_batch");
    if (status != 0) throw Panic("called `Option::unwrap()` on a `None` value (mod_inv, multiplication_proof.rs:95)");
    return MulProof{BigInt::from_limbs(f.data(), kw), BigInt::from_limbs(z1.data(), 2 * kw), BigInt::from_limbs(z2.data(), 2 * kw),
                    BigInt::from_limbs(ed.data(), 2 * kw), BigInt::from_limbs(edb.data(), 2 * kw)};
  }
  Result verify(const MulStatement& st) const {                         // :106-146
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(st.ek.n), kw = nb / 32;
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(st.ek.n, kw), ea = L1(st.e_a, 2 * kw), eb = L1(st.e_b, 2 * kw), ec = L1(st.e_c, 2 * kw);
    auto vf = L1(f, kw), vz1 = L1(z1, 2 * kw), vz2 = L1(z2, 2 * kw), ved = L1(e_d, 2 * kw), vedb = L1(e_db, 2 * kw);
    uint8_t v = 9;
    e.check(zkp_mul_proof_verify_batch(e.ctx(), nb, 1, n.data(), 0, ea.data(), eb.data(), ec.data(), vf.data(), vz1.data(), vz2.data(), ved.data(), vedb.data(),
                                       &v, 0), "zkp_mul_proof_verify_batch");
    if (v == ZKP_VERDICT_MALFORMED) throw Panic("called `Option::unwrap()` on a `None` value (mod_inv, multiplication_proof.rs:135)");
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ CorrectMessageProof (src/zkproofs/correct_message.rs)
class CorrectMessageProof {
 public:
  std::vector<BigInt> e_vec, z_vec, a_vec;
  BigInt ciphertext;
  std::vector<BigInt> valid_messages;
  EncryptionKey ek;
  static constexpr size_t B = 256;   // :19
  static CorrectMessageProof prove(const EncryptionKey& ek, const std::vector<BigInt>& valid_messages, const BigInt& message_to_encrypt) {   // :35-123
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t K = valid_messages.size();
    if (K == 0) throw Panic("attempt to subtract with overflow (num_of_message - 1, correct_message.rs:58)");
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(ek.n, kw), msg = L1(message_to_encrypt, kw), r = L1(BigInt::sample_below(ek.n), kw), w = L1(BigInt::sample_below(ek.n), kw);   // :43, :65
    std::vector<uint32_t> valid(K * kw), esim((K - 1) * 8 + 8), zsim((K - 1) * kw + kw);
    for (size_t i = 0; i < K; i++) valid_messages[i].to_limbs(&valid[i * kw], kw);
    for (size_t j = 0; j + 1 < K; j++) {
      BigInt::sample(B).to_limbs(&esim[j * 8], 8);                       // :59-61
      BigInt::sample_below(ek.n).to_limbs(&zsim[j * kw], kw);            // :62-64
    }
    std::vector<uint32_t> ct(2 * kw), ev(K * 8), zv(K * kw), av(K * 2 * kw);
    uint8_t status = 9;
    e.check(zkp_correct_message_prove_batch(e.ctx(), nb, 1, (uint32_t)K, n.data(), 0, valid.data(), msg.data(), r.data(), esim.data(), zsim.data(), w.data(),
                                            ct.data(), ev.data(), zv.data(), av.data(), &status, 0), "zkp_correct_message_prove_batch");
    if (status != 0) throw Panic("index out of bounds (correct_message.rs:72: no valid message equals the encrypted one)");
    CorrectMessageProof p;
    p.ciphertext = BigInt::from_limbs(ct.data(), 2 * kw); p.valid_messages = valid_messages; p.ek = ek;
    for (size_t i = 0; i < K; i++) {
      p.e_vec.push_back(BigInt::from_limbs(&ev[i * 8], 8)); p.z_vec.push_back(BigInt::from_limbs(&zv[i * kw], kw));
      p.a_vec.push_back(BigInt::from_limbs(&av[i * 2 * kw], 2 * kw));
    }
    return p;
  }
  Result verify() const {                                                 // :124-162
    Engine& e = Engine::instance();
    const uint32_t nb = width_for(ek.n), kw = nb / 32;
    const size_t K = valid_messages.size();
    if (e_vec.size() < K || z_vec.size() < K || a_vec.size() < K) throw Panic("index out of bounds");
    auto L1 = [&](const BigInt& v, size_t n) { std::vector<uint32_t> o(n); v.to_limbs(o.data(), n); return o; };
    auto n = L1(ek.n, kw), ct = L1(ciphertext, 2 * kw);
    std::vector<uint32_t> valid(K * kw), ev(K * 8), zv(K * kw), av(K * 2 * kw);
    for (size_t i = 0; i < K; i++) {
      valid_messages[i].to_limbs(&valid[i * kw], kw); e_vec[i].to_limbs(&ev[i * 8], 8); z_vec[i].to_limbs(&zv[i * kw], kw); a_vec[i].to_limbs(&av[i * 2 * kw], 2 * kw);
    }
    uint8_t v = 9;
    e.check(zkp_correct_message_verify_batch(e.ctx(), nb, 1, (uint32_t)K, n.data(), 0, valid.data(), ct.data(), ev.data(), zv.data(), av.data(), &v, 0),
            "zkp_correct_message_verify_batch");
    if (v == ZKP_VERDICT_MALFORMED) throw Panic("assertion failed: `(left == right)` chal / ei_sum (correct_message.rs:132)");
    return Result(v == ZKP_VERDICT_ACCEPT);
  }
};

// ------------------------------------------------------------------ wire format (src/serialize.rs + the serde derives)
// Writers produce what serde_json::to_string produces for the reference's types; readers go through the batched
// ingestion entry points of the C ABI (decimal strings are converted on the GPU).
namespace serde_json {
// BigInt::to_str_radix(10) for many values at once (zkp_limbs_to_decimal_batch)
inline std::vector<std::string> to_decimal(const std::vector<BigInt>& v, uint32_t words) {
  Engine& e = Engine::instance();
  const size_t cnt = v.size();
  const uint32_t pitch = zkp_decimal_pitch(words);
  std::vector<uint32_t> src(cnt * words), len(cnt);
  std::vector<char> txt(cnt * (size_t)pitch);
  for (size_t i = 0; i < cnt; i++) v[i].to_limbs(&src[i * words], words);
  if (cnt) e.check(zkp_limbs_to_decimal_batch(e.ctx(), src.data(), words, words, cnt, txt.data(), pitch, len.data(), 0), "zkp_limbs_to_decimal_batch");
  std::vector<std::string> out;
  for (size_t i = 0; i < cnt; i++) out.emplace_back(&txt[i * (size_t)pitch + pitch - len[i]], len[i]);
  return out;
}
inline std::string vecbigint(const std::vector<std::string>& d, size_t lo, size_t n) {   // serialize.rs:33-46
  std::string s = "[";
  for (size_t i = 0; i < n; i++) { if (i) s += ","; s += "\"" + d[lo + i] + "\""; }
  return s + "]";
}
inline std::string to_string(const EncryptedPairs& p, const EncryptionKey& ek) {          // range_proof.rs:32-39
  const uint32_t w = 2 * width_for(ek.n) / 32;
  std::vector<BigInt> all(p.c1); all.insert(all.end(), p.c2.begin(), p.c2.end());
  auto d = to_decimal(all, w);
  return "{\"c1\":" + vecbigint(d, 0, p.c1.size()) + ",\"c2\":" + vecbigint(d, p.c1.size(), p.c2.size()) + "}";
}
inline std::string to_string(const Proof& p, const EncryptionKey& ek) {                   // range_proof.rs:53-81
  const uint32_t w = width_for(ek.n) / 32;
  std::vector<BigInt> all;
  for (const Response& r : p.responses) {
    if (r.kind == Response::Open) { all.push_back(r.w1); all.push_back(r.r1); all.push_back(r.w2); all.push_back(r.r2); }
    else { all.push_back(r.masked_x); all.push_back(r.masked_r); }
  }
  auto d = to_decimal(all, w);
  std::string s = "[";
  size_t k = 0;
  for (size_t i = 0; i < p.responses.size(); i++) {
    const Response& r = p.responses[i];
    if (i) s += ",";
    if (r.kind == Response::Open) { s += "{\"Open\":{\"w1\":\"" + d[k] + "\",\"r1\":\"" + d[k + 1] + "\",\"w2\":\"" + d[k + 2] + "\",\"r2\":\"" + d[k + 3] + "\"}}"; k += 4; }
    else { s += "{\"Mask\":{\"j\":" + std::to_string((unsigned)r.j) + ",\"masked_x\":\"" + d[k] + "\",\"masked_r\":\"" + d[k + 1] + "\"}}"; k += 2; }
  }
  return s + "]";
}
inline std::string to_string(const NiCorrectKeyProof& p, const EncryptionKey& ek) {       // correct_key_ni.rs:35-39
  auto d = to_decimal(p.sigma_vec, width_for(ek.n) / 32);
  return "{\"sigma_vec\":" + vecbigint(d, 0, d.size()) + "}";
}

// serde_json::from_str for many documents of one shape; an Err (or a value the fixed-width ABI cannot carry) throws
inline std::vector<std::pair<EncryptedPairs, Proof>> range_from_str(const EncryptionKey& ek, size_t error_factor, const std::vector<std::string>& pairs_docs,
                                                                    const std::vector<std::string>& proof_docs) {
  Engine& e = Engine::instance();
  const uint32_t nb = width_for(ek.n), kw = nb / 32;
  const size_t B = pairs_docs.size(), EF = error_factor, rows = B * EF;
  if (proof_docs.size() != B) throw std::invalid_argument("range_from_str: one EncryptedPairs and one Proof document per proof");
  std::vector<uint32_t> c1(rows * 2 * kw), c2(rows * 2 * kw), w1(rows * kw), r1(rows * kw), w2(rows * kw), r2(rows * kw);
  std::vector<uint8_t> kind(rows), jj(rows), st1(B), st2(B);
  zkp_range_ni_proofs p{};
  p.n_bits = nb; p.error_factor = (uint32_t)EF; p.batch = B; p.c1 = c1.data(); p.c2 = c2.data(); p.resp_kind = kind.data(); p.resp_j = jj.data();
  p.resp_w1 = w1.data(); p.resp_r1 = r1.data(); p.resp_w2 = w2.data(); p.resp_r2 = r2.data();
  auto run = [&](const std::vector<std::string>& docs, bool proofs, std::vector<uint8_t>& st) {
    std::string text; std::vector<uint64_t> off, len;
    for (auto& d : docs) { off.push_back(text.size()); len.push_back(d.size()); text += d; }
    if (proofs) e.check(zkp_json_range_proof_batch(e.ctx(), text.data(), off.data(), len.data(), &p, st.data(), 0), "zkp_json_range_proof_batch");
    else e.check(zkp_json_encrypted_pairs_batch(e.ctx(), text.data(), off.data(), len.data(), &p, st.data(), 0), "zkp_json_encrypted_pairs_batch");
  };
  run(pairs_docs, false, st1); run(proof_docs, true, st2);
  std::vector<std::pair<EncryptedPairs, Proof>> out(B);
  for (size_t b = 0; b < B; b++) {
    if (st1[b] || st2[b]) throw std::runtime_error("serde_json: document " + std::to_string(b) + " is not a valid EncryptedPairs / Proof of this width");
    for (size_t i = 0; i < EF; i++) {
      const size_t t = b * EF + i;
      out[b].first.c1.push_back(BigInt::from_limbs(&c1[t * 2 * kw], 2 * kw)); out[b].first.c2.push_back(BigInt::from_limbs(&c2[t * 2 * kw], 2 * kw));
      Response rs;
      if (kind[t] == ZKP_RESP_OPEN) {
        rs.kind = Response::Open;
        rs.w1 = BigInt::from_limbs(&w1[t * kw], kw); rs.r1 = BigInt::from_limbs(&r1[t * kw], kw); rs.w2 = BigInt::from_limbs(&w2[t * kw], kw); rs.r2 = BigInt::from_limbs(&r2[t * kw], kw);
      } else {
        rs.kind = Response::Mask; rs.j = jj[t];
        rs.masked_x = BigInt::from_limbs(&w1[t * kw], kw); rs.masked_r = BigInt::from_limbs(&r1[t * kw], kw);
      }
      out[b].second.responses.push_back(std::move(rs));
    }
  }
  return out;
}
inline NiCorrectKeyProof correct_key_from_str_host(const std::string& doc);
inline NiCorrectKeyProof correct_key_from_str(const EncryptionKey& ek, const std::string& doc) {
  Engine& e = Engine::instance();
  const uint32_t nb = width_for(ek.n), kw = nb / 32;
  std::vector<uint32_t> sig(NiCorrectKeyProof::M2 * kw);
  uint64_t off = 0, len = doc.size();
  uint8_t st = 9;
  e.check(zkp_json_correct_key_proof_batch(e.ctx(), doc.data(), &off, &len, nb, 1, sig.data(), &st, 0), "zkp_json_correct_key_proof_batch");
  if (st == ZKP_DOC_HOST_PATH) return correct_key_from_str_host(doc);     // a negative or over-wide root: a valid NiCorrectKeyProof, parsed here
  if (st) throw std::runtime_error("serde_json: not a NiCorrectKeyProof of this width");
  NiCorrectKeyProof p;
  for (size_t i = 0; i < NiCorrectKeyProof::M2; i++) p.sigma_vec.push_back(BigInt::from_limbs(&sig[i * kw], kw));
  return p;
}

// ---- serde_json::from_str::<RangeProofNi> on the HOST, for the documents the GPU reader hands back (status ZKP_DOC_HOST_PATH: a
// well-formed document holding a number the fixed-width ABI cannot carry — negative or over-wide — or another row count).  Numbers
// become signed BigInts of any size, so RangeProofNi::verify_batch gives such a proof the reference's verdict (verify_general).
// encrypted_pairs / proof: decimal strings (serialize.rs:1-78, mpz_set_str: a leading '-' is a sign).  ek.n and the bare BigInts
// range / ciphertext are un-annotated in the reference (range_proof_ni.rs:38-40): their text forms are the caller's to name,
// separately (kzen-paillier's EncryptionKey and curv's BigInt need not agree), as for zkp_json_range_proof_ni_batch.
enum class BigintText { Dec = ZKP_BIGINT_DEC, Hex = ZKP_BIGINT_HEX, Bytes = ZKP_BIGINT_BYTES };
namespace detail_json {
struct Cur {
  const std::string& t; size_t p = 0;
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("serde_json: ") + what + " at byte " + std::to_string(p)); }
  void ws() { while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) p++; }
  bool eat(char c) { ws(); if (p < t.size() && t[p] == c) { p++; return true; } return false; }
  void expect(char c) { if (!eat(c)) fail("unexpected token"); }
  std::string str() {
    expect('"');
    std::string out;
    while (p < t.size() && t[p] != '"') {
      if (t[p] == '\\') {
        if (++p >= t.size()) fail("bad escape");
        const char c = t[p++];
        if (c == 'u') {
          if (p + 4 > t.size()) fail("bad \\u escape");
          const unsigned v = (unsigned)std::stoul(t.substr(p, 4), nullptr, 16); p += 4;
          if (v > 0x7f) fail("non-ASCII escape in a number or field name");
          out.push_back((char)v);
        } else out.push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c == 'b' ? '\b' : c == 'f' ? '\f' : c);
      } else out.push_back(t[p++]);
    }
    if (p >= t.size()) fail("unterminated string");
    p++;
    return out;
  }
  uint64_t uint() {
    ws();
    if (p >= t.size() || t[p] < '0' || t[p] > '9') fail("expected an unsigned integer");
    uint64_t v = 0;
    while (p < t.size() && t[p] >= '0' && t[p] <= '9') { if (v > (~0ull - 9) / 10) fail("integer overflow"); v = v * 10 + (uint64_t)(t[p++] - '0'); }
    if (p < t.size() && (t[p] == '.' || t[p] == 'e' || t[p] == 'E')) fail("expected an integer");
    return v;
  }
  void skip() {
    ws();
    if (p >= t.size()) fail("unexpected end");
    if (t[p] == '"') { (void)str(); return; }
    if (t[p] == '{' || t[p] == '[') {
      const char close = t[p] == '{' ? '}' : ']';
      p++;
      if (eat(close)) return;
      do { if (close == '}') { (void)str(); expect(':'); } skip(); } while (eat(','));
      expect(close); return;
    }
    while (p < t.size() && t[p] != ',' && t[p] != '}' && t[p] != ']' && t[p] != ' ' && t[p] != '\n') p++;
  }
  template <class F> void object(F&& field) {      // field(name) consumes the value and returns true, or returns false (unknown: skipped)
    expect('{');
    if (eat('}')) return;
    do { const std::string nm = str(); expect(':'); if (!field(nm)) skip(); } while (eat(','));
    expect('}');
  }
  BigInt dec() { const std::string s = str(); try { return BigInt::from_str_radix10(s); } catch (const std::invalid_argument&) { fail("not a decimal integer"); } }
  BigInt bigint(BigintText form) {
    if (form == BigintText::Dec) return dec();
    if (form == BigintText::Bytes) {
      std::vector<uint8_t> b;
      expect('[');
      if (!eat(']')) { do { const uint64_t v = uint(); if (v > 255) fail("byte out of range"); b.push_back((uint8_t)v); } while (eat(',')); expect(']'); }
      return BigInt::from_bytes(b);
    }
    std::string h = str();
    const bool minus = !h.empty() && h[0] == '-';
    if (minus) h.erase(0, 1);
    if (h.empty()) fail("empty hex integer");
    BigInt r;
    r.l.assign((h.size() + 7) / 8, 0);
    for (size_t i = 0; i < h.size(); i++) {
      const char c = h[h.size() - 1 - i];
      const int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
      if (v < 0) fail("not a hex integer");
      r.l[i / 8] |= (uint32_t)v << (4 * (i % 8));
    }
    r.trim();
    r.neg = minus && !r.is_zero();
    return r;
  }
};
}  // namespace detail_json

// {"sigma_vec":["..", ..]} (correct_key_ni.rs:35-39) with signed roots of any size and any count
inline NiCorrectKeyProof correct_key_from_str_host(const std::string& doc) {
  detail_json::Cur j{doc};
  NiCorrectKeyProof out;
  bool seen = false;
  j.object([&](const std::string& nm) {
    if (nm != "sigma_vec") return false;
    if (seen) j.fail("duplicate field");
    seen = true;
    j.expect('[');
    if (!j.eat(']')) { do out.sigma_vec.push_back(j.dec()); while (j.eat(',')); j.expect(']'); }
    return true;
  });
  if (!seen) j.fail("missing field `sigma_vec`");
  j.ws();
  if (j.p != doc.size()) j.fail("trailing characters");
  return out;
}

inline RangeProofNi range_proof_ni_from_str(const std::string& doc, BigintText key_form = BigintText::Dec, BigintText bigint_form = BigintText::Dec) {
  detail_json::Cur j{doc};
  RangeProofNi out;
  unsigned seen = 0;
  auto once = [&](unsigned bit) { if (seen & bit) j.fail("duplicate field"); seen |= bit; };
  auto vec = [&](std::vector<BigInt>& v) { j.expect('['); if (!j.eat(']')) { do v.push_back(j.dec()); while (j.eat(',')); j.expect(']'); } };
  j.object([&](const std::string& nm) {
    if (nm == "ek") {
      once(1);
      bool has_n = false;
      j.object([&](const std::string& f) { if (f != "n") return false; if (has_n) j.fail("duplicate field"); has_n = true; out.ek.n = j.bigint(key_form); return true; });
      if (!has_n) j.fail("missing field `n`");
      out.ek.nn = out.ek.n * out.ek.n;       // MinimalEncryptionKey -> EncryptionKey { n, nn = n * n } [upstream kzen-paillier]
    } else if (nm == "range") { once(2); out.range = j.bigint(bigint_form); }
    else if (nm == "ciphertext") { once(4); out.ciphertext = j.bigint(bigint_form); }
    else if (nm == "encrypted_pairs") {
      once(8);
      unsigned s2 = 0;
      j.object([&](const std::string& f) {
        if (f == "c1") { if (s2 & 1) j.fail("duplicate field"); s2 |= 1; vec(out.encrypted_pairs.c1); return true; }
        if (f == "c2") { if (s2 & 2) j.fail("duplicate field"); s2 |= 2; vec(out.encrypted_pairs.c2); return true; }
        return false;
      });
      if (s2 != 3) j.fail("missing field of EncryptedPairs");
    } else if (nm == "proof") {
      once(16);
      j.expect('[');
      if (!j.eat(']')) {
        do {
          Response rs;
          int variants = 0;
          j.object([&](const std::string& var) {
            if (++variants > 1) j.fail("expected a single-variant enum map");
            unsigned s3 = 0;
            if (var == "Open") {
              rs.kind = Response::Open;
              j.object([&](const std::string& f) {
                BigInt* dst = f == "w1" ? &rs.w1 : f == "r1" ? &rs.r1 : f == "w2" ? &rs.w2 : f == "r2" ? &rs.r2 : nullptr;
                if (!dst) return false;
                const unsigned bit = f == "w1" ? 1 : f == "r1" ? 2 : f == "w2" ? 4 : 8;
                if (s3 & bit) j.fail("duplicate field");
                s3 |= bit;
                *dst = j.dec();
                return true;
              });
              if (s3 != 15) j.fail("missing field of Response::Open");
            } else if (var == "Mask") {
              rs.kind = Response::Mask;
              j.object([&](const std::string& f) {
                if (f == "j") { if (s3 & 1) j.fail("duplicate field"); s3 |= 1; const uint64_t v = j.uint(); if (v > 255) j.fail("j is a u8"); rs.j = (uint8_t)v; return true; }
                if (f == "masked_x") { if (s3 & 2) j.fail("duplicate field"); s3 |= 2; rs.masked_x = j.dec(); return true; }
                if (f == "masked_r") { if (s3 & 4) j.fail("duplicate field"); s3 |= 4; rs.masked_r = j.dec(); return true; }
                return false;
              });
              if (s3 != 7) j.fail("missing field of Response::Mask");
            } else j.fail("unknown variant of Response");
            return true;
          });
          if (variants != 1) j.fail("expected a single-variant enum map");
          out.proof.responses.push_back(std::move(rs));
        } while (j.eat(','));
        j.expect(']');
      }
    } else if (nm == "error_factor") { once(32); out.error_factor = (size_t)j.uint(); }
    else return false;
    return true;
  });
  if (seen != 63u) j.fail("missing field of RangeProofNi");
  j.ws();
  if (j.p != doc.size()) j.fail("trailing characters");
  return out;
}
}  // namespace serde_json

}  // namespace zkproofs
