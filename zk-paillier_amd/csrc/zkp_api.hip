// zkp_api.hip — host side of libzkp_hip.so: the C ABI of include/zkp_hip.h.
// One ctx = one GPU + one HIP stream; all device scratch (Montgomery constants, window
// tables, work lists) is owned by the ctx and grown on demand.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zkp_hip.h"
#include "../../include/zkp_hip_diag.h"
#include "kernels_modexp.hpp"
#include "kernels_proofs.hpp"
#include "kernels_inv.hpp"
#include "kernels_serde.hpp"
#if ZKP_W == 36 || ZKP_W == 18 || ZKP_W == 9
#define ZKP_HAS_BASEN 1
#include "kernels_basen.hpp"
#ifndef ZKP_R2L5_ITEMS_PER_CU
#define ZKP_R2L5_ITEMS_PER_CU 1ull      /* launches of up to this many Enc per compute unit take five wavefronts per Enc (k_enc_basen_r2l5): one proof 13.3 / 9.3 -> 10.2 / 6.0 ms;
                                           two proofs (two workgroups per CU) 13.8 / 9.6 against 13.9 / 9.7 on one wavefront per Enc, but the verify time is bimodal there (9.6 or 12.4 ms:
                                           a transcript-hash wavefront that shares its SIMD with three others); three: 17.7 / 13.4 against 13.9 / 9.9 (profiles/r05/r2l5/) */
#endif
#include "kernels_basen_r2l.hpp"      // (W = 9 only: one Enc per wavefront, the five-group right-to-left ladder of calls of a few proofs)          // the Paillier kernels in base-n form: 2 / 4 lanes per n-sized integer in the throughput engine (W = 36), 8 / 16 in the latency engine (W = 9)
#else
#define ZKP_HAS_BASEN 0
#endif

using namespace zkp;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // constants buffers only (run_setup): the tag of the ONE key whose record the buffer holds (kernels_modexp.hpp), and what tells the host
  // that the record is no longer that key's — `gen` moves with every reallocation and every launch of several keys into the buffer
  uint32_t* tag = nullptr;
  uint64_t gen = 0, tag_gen = 0;
};

// staging blocks kept by the ctx between calls (host-pointer mode): a call takes the best-fitting cached block or
// allocates one, and hands everything back when it returns; zkp_ctx_release_staging / zkp_ctx_destroy free them
struct StageBlock { void* p; size_t cap; };

struct zkp_ctx {
  std::vector<StageBlock> stage_free;
  uint32_t* setup_flag = nullptr;      // device word: k_setup ORs the status of every modulus it rejects into it
  uint32_t* setup_flag_host = nullptr; // its pinned host mirror
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true;
  int cus = 0;
  // the latency engine's twin of this ctx (same device, same stream); null in the latency engine itself or when it is not loaded
  // (two secondary engines: [0] libzkp_hip_lat.so, 9 limbs per lane — calls of a few proofs; [1] libzkp_hip_mid.so, 18 — mid-size Paillier
  //  calls under one key.  `lat` / `lat_ctx` name the one the current — or most recent — routed call runs on)
  const struct LatEngine* eng[2] = {nullptr, nullptr};
  zkp_ctx* eng_ctx[2] = {nullptr, nullptr};
  const struct LatEngine* lat = nullptr;
  zkp_ctx* lat_ctx = nullptr;
  int geometry = 0;                    // 0 = automatic, else the limbs per lane every call must run on
  int last_geometry = 0;
  // second stream + fork / join events for calls of a few proofs (hash next to the Enc checks); created on first use
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_busy = false;        // work forked onto `side` whose join has not been enqueued on `stream` yet (see ~Stage)
  // copy stream + events of a host-pointer call that runs as a pipeline of proof blocks (Piped below); created on first use
  hipStream_t copy = nullptr;
  std::vector<hipEvent_t> ev_pipe;
  bool copy_busy = false;        // copies enqueued on `copy` that nothing has waited for yet
  int last_host_blocks = 0;      // proof blocks of the most recent RangeProofNi host-pointer call (1: the plain path)
  // A RangeProofNi call of 65 ... 96 proofs under one 2048-bit key runs as TWO concurrent calls (range_split in zkp_api_proofs.inc): one
  // wavefront per SIMD of the mid engine (64 proofs) on this ctx's stream, the rest on a second ctx of the latency engine with a stream of
  // its own — created on first use.  $ZKP_SPLIT=0 at zkp_ctx_create / zkp_diag_set_split turn it off.
  zkp_ctx* split_ctx = nullptr;
  hipStream_t split_stream = nullptr;
  hipEvent_t ev_split[2] = {nullptr, nullptr};
  int split_calls = 1;
  int last_split = 0;            // proofs the most recent RangeProofNi call sent to the latency engine beside the mid engine (0: it was not split)
  int key_cache = 1;             // keep the constants of the ONE key of a shared-key call across calls (setup_tag below); $ZKP_KEY_CACHE=0 at zkp_ctx_create or zkp_diag_set_key_cache turn it off
  int host_chunks = -1;          // $ZKP_HOST_CHUNKS at zkp_ctx_create: unset (-1) or 1 = a host-pointer call is one block, N = N equal blocks, 0 = uneven blocks (host_blocks)
  std::string err;
  DevBuf consts, consts2, table, scratch[48];
  DevBuf bn_ncst, bn_consts, bn_table, bn_expected, bn_raw, bn_left;
  int enc_form = ZKP_ENC_FORM_AUTO;    // which Paillier launches take the base-n form (zkp_diag_set_enc_form; $ZKP_BASEN is read ONCE, at zkp_ctx_create)
  int bn_occ[2][2] = {{0, 0}, {0, 0}}; // resident workgroups per CU of k_enc_basen<G> / k_enc_basen_keys<G> ([per-key][n = 4096]; 0: not asked yet)
  int bn_last_g = 0;                   // lanes per n-sized integer of the most recent base-n launch (0: none yet)
  bool bn_last_per_key = false;        // ... and whether it ran under per-proof keys
  bool bn_last_r2l = false;            // ... and whether it was the one-Enc-per-wavefront ladder of the latency engine (kernels_basen_r2l.hpp)
  int bn_r2l_lanes = 0;                // ... its lane geometry: 0 = the library's rule (five wavefronts of 36 lanes x 2 limbs per Enc while every Enc finds a CU's worth of
                                       // SIMDs, else one wavefront of five groups of 12 lanes x 6 limbs); 36 / 12 / 8 (x 9 limbs) pin one ($ZKP_R2L_LANES at ctx create, zkp_diag_set_r2l_lanes)
  int bn_last_r2l_lanes = 0;           // ... and the geometry the most recent such launch ran on
  int bn_r2l = 1;                      // that ladder: 0 = never, 1 = the library's rule (launches of up to two wavefronts per SIMD), 2 = whenever it can run (tests); $ZKP_R2L at ctx create
  // The transcript hashes of a verify call of a few proofs as workgroups of its Enc launch (k_enc_basen_r2l5): range_verify_impl names them here
  // before launch_k_enc, the launch that takes them says so; $ZKP_FUSE_HASH=0 at zkp_ctx_create keeps them in a launch of their own on `side`.
  const RangeHashArgs* fuse_hash = nullptr;
  bool fuse_hash_taken = false;
  int fuse_hash_on = 1;
  int grid_expected = 1;               // the base-n launch of a verify sized by the items expected, not by the bound (launch_basen); $ZKP_GRID_EXPECTED=0 at zkp_ctx_create: A/B runs
  DevBuf bn_flag;                      // device word: every key of the last batched base-n set-up qualified   // base-n form (kernels_basen.hpp): set-up record of n, its base-n constants, window tables, Mask-row products
  // timing of the dominant kernels
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
  uint64_t timed_launches = 0, timed_modexps = 0;
  // device-resident work-item counts of timed verify launches land here (pinned host memory)
  static constexpr size_t PINNED_SLOTS = 4096;
  unsigned long long* pinned_counts = nullptr;
  size_t pinned_used = 0;
};

#define HIPCHK(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
      return e_ == hipErrorOutOfMemory ? ZKP_ENOMEM : ZKP_EDEVICE;                                \
    }                                                                                             \
  } while (0)

// nothing may propagate through the C ABI: every extern "C" entry point is a function-try-block ending in ZKP_CATCH
static int32_t zkp_caught(zkp_ctx* c, int32_t st, const char* what) noexcept {
  if (c) { try { c->err = what; } catch (...) {} }
  return st;
}
#define ZKP_CATCH(ctx)                                                                                         \
  catch (const std::bad_alloc&) { return zkp_caught((ctx), ZKP_ENOMEM, "out of host memory"); }               \
  catch (const std::exception& e_) { return zkp_caught((ctx), ZKP_EDEVICE, e_.what()); }                      \
  catch (...) { return zkp_caught((ctx), ZKP_EDEVICE, "unknown exception"); }

// ---- the latency engine -----------------------------------------------------------------------
// libzkp_hip_lat.so is THIS source built with W = 9 (-DZKP_SECONDARY_ENGINE): four times the lanes per big integer, so a
// chain of dependent Montgomery products finishes in less than half the time while the launch is too small to fill the GPU
// anyway.  It is loaded once per process from the directory of this library (or $ZKP_HIP_LAT_LIB); every ctx gets a twin
// ctx of it on the same stream, and the batch entry points hand small calls over (ZKP_ROUTE).
#define ZKP_LAT_FUNCS(X)                                                                                                        \
  X(zkp_ctx_create_on_stream) X(zkp_ctx_destroy) X(zkp_last_error_string) X(zkp_build_limbs_per_lane) X(zkp_ctx_release_staging) \
  X(zkp_timing_reset) X(zkp_timing_get) X(zkp_modexp_batch) X(zkp_paillier_enc_batch) X(zkp_paillier_enc_check_batch)            \
  X(zkp_range_ni_prove_batch) X(zkp_range_ni_verify_batch) X(zkp_range_generate_encrypted_pairs_batch) X(zkp_range_challenge_batch)                           \
  X(zkp_range_verifier_output_batch) X(zkp_correct_key_ni_verify_batch) X(zkp_dlog_prove_batch) X(zkp_dlog_verify_batch)         \
  X(zkp_zero_proof_prove_batch) X(zkp_zero_proof_verify_batch) X(zkp_ciphertext_proof_prove_batch)                               \
  X(zkp_ciphertext_proof_verify_batch) X(zkp_verlin_proof_prove_batch) X(zkp_verlin_proof_verify_batch)                          \
  X(zkp_mul_proof_prove_batch) X(zkp_mul_proof_verify_batch) X(zkp_correct_message_prove_batch) X(zkp_correct_message_verify_batch)           \
  X(zkp_diag_basen) X(zkp_diag_basen_last) X(zkp_diag_set_enc_form) X(zkp_diag_set_r2l) X(zkp_diag_r2l_last) X(zkp_diag_set_r2l_lanes) X(zkp_diag_r2l_lanes_last) X(zkp_diag_set_key_cache) X(zkp_diag_key_cache_state) \
  X(zkp_diag_set_fuse_hash) X(zkp_diag_last_fused_hash)

struct LatEngine {
  void* handle = nullptr;
  int limbs_per_lane = 0;
#define X(f) decltype(&f) p_##f = nullptr;
  ZKP_LAT_FUNCS(X)
#undef X
};

// status of a call made on the twin ctx, with its error text copied over (the twin only exists in the throughput build)
static int32_t lat_forward_plain(zkp_ctx* c, int32_t st) {
  if (st && c->lat_ctx) { try { c->err = c->lat->p_zkp_last_error_string(c->lat_ctx); } catch (...) {} }
  return st;
}

// the items a verify's work list is expected to hold when the launch is sized for `bound` (2 per row: an Open row has two Enc checks, a Mask
// row one, the challenge bits are fair): three quarters, + 3 % (5 proofs: 960 + 40 of 1280; the spread of 5 proofs is 13)
static uint64_t expected_items(uint64_t bound) { return (3 * bound + 3) / 4 + bound / 32; }
// which of the ctx's secondary engines has this many limbs per lane (-1: none; always -1 inside a secondary engine)
static int engine_with(const zkp_ctx* c, int limbs_per_lane) {
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k] && c->eng[k]->limbs_per_lane == limbs_per_lane) return k;
  return -1;
}

#ifndef ZKP_SECONDARY_ENGINE
// directory of this shared library, resolved when it is loaded (a relative load path stops meaning anything once the process
// changes its working directory)
static const std::string& own_directory() {
  static const std::string dir = [] {
    Dl_info info;
    if (!dladdr((const void*)&zkp_backend_name, &info) || !info.dli_fname) return std::string();
    char* real = realpath(info.dli_fname, nullptr);
    std::string path = real ? real : info.dli_fname;
    std::free(real);
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
  }();
  return dir;
}
__attribute__((constructor)) static void resolve_own_directory() { try { (void)own_directory(); } catch (...) {} }

// secondary engine `which`: 0 = libzkp_hip_lat.so ($ZKP_HIP_LAT_LIB), 1 = libzkp_hip_mid.so ($ZKP_HIP_MID_LIB), next to this library
static const LatEngine* secondary_engine(int which) {
  static LatEngine engs[2];
  static std::once_flag once[2];
  std::call_once(once[which], [which] {
    LatEngine& eng = engs[which];
    std::string path;
    if (const char* e = std::getenv(which ? "ZKP_HIP_MID_LIB" : "ZKP_HIP_LAT_LIB")) path = e;
    else if (!own_directory().empty()) path = own_directory() + (which ? "/libzkp_hip_mid.so" : "/libzkp_hip_lat.so");
    else return;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    bool all = true;
#define X(f) eng.p_##f = (decltype(&f))dlsym(h, #f); all = all && eng.p_##f;
    ZKP_LAT_FUNCS(X)
#undef X
    if (!all || eng.p_zkp_build_limbs_per_lane() == W) { dlclose(h); return; }   // a stale or identical build: not an engine
    eng.limbs_per_lane = eng.p_zkp_build_limbs_per_lane();
    eng.handle = h;
  });
  return engs[which].handle ? &engs[which] : nullptr;
}
// the engine the call in hand runs on
static void select_engine(zkp_ctx* c, int k) {
  c->lat = c->eng[k]; c->lat_ctx = c->eng_ctx[k];
  c->last_geometry = c->lat->limbs_per_lane;
}

// does this call (items independent modexp chains under mod_bits-bit moduli) go to the latency engine?
// one_key_paillier: the items are Paillier Enc under ONE 2048-bit key — the latency engine then runs them in base-n form, 8 Enc per wavefront
// (k_enc_basen<8>), and stays ahead of the throughput engine up to three wavefronts per SIMD.  Measured round 5 (tools/dev/size_sweep.py,
// profiles/r05/size_sweep.jsonl, prove / verify ms): 48 proofs 43.9 / 43.5 against 49.3 / 44.2 on the n^2-sized throughput kernels, 64 proofs
// 45.1 / 44.3 against 51.0 / 44.3, 96 proofs 60.7 / 59.7 against 64.9 / 61.6 on the throughput engine's base-n kernels, 128 proofs 76.7 / 74.5
// against 66.7 / 61.7: the hand-over is at 96 proofs.
// With the mid engine loaded (18 limbs per lane: 16 Enc per wavefront in base-n form, k_enc_basen<4>) the one-key Paillier calls are cut
// three ways (same sweep, profiles/r05/size_sweep_w18.jsonl): up to 40 proofs the latency engine (32 proofs 33.1 / 29.5 ms against 40.0 / 38.3),
// from there to 64 proofs — one wavefront per SIMD at 16 Enc each — the mid engine (64 proofs 41.4 / 38.5 against 45.9 / 44.4), the latency
// engine again up to 96 (80 proofs 61.1 / 59.7 against 65.2 / 63.9), the throughput engine beyond — except 129 ... 192 proofs, where two mid
// wavefronts per SIMD beat its second round of 32-Enc claims (160 proofs 90.7 / 67.7 against 111.7 / 66.1, 192: 93.2 / 89.8 against 113.6 / 111.2).
// `listed`: the items are the bound of a verify's work list (2 per row; about three quarters exist): the mid engine's one-wavefront-per-SIMD
// window is then judged by the items EXPECTED — its grid is sized by them too (launch_basen) — so that 65 ... 81 proofs verify in one round
// there (37.4 ms) instead of as two concurrent calls (46.3; tools/dev/two_streams_sweep.py with the engine pinned, profiles/r06/expected_items/).
static bool route_latency(zkp_ctx* c, uint64_t items, uint32_t mod_bits, bool one_key_paillier = false, bool listed = false) {
  c->last_geometry = W;
  if (!c->eng_ctx[0] && !c->eng_ctx[1]) return false;
  if (c->geometry == W || items == 0) return false;
  if (c->geometry) {                                              // pinned to a secondary engine
    const int k = engine_with(c, c->geometry);
    if (k < 0) return false;
    select_engine(c, k);
    return true;
  }
  const uint64_t simds = 4 * (uint64_t)c->cus;
  if (one_key_paillier && mod_bits == 4096 && c->enc_form != ZKP_ENC_FORM_N2) {
    const int lat9 = engine_with(c, 9), mid18 = engine_with(c, 18);
    const bool fits_one_round = items <= 16 * simds || (listed && c->grid_expected && expected_items(items) <= 16 * simds);
    if (mid18 >= 0 && ((items > 10 * simds && fits_one_round) || (items > 32 * simds && items <= 48 * simds))) { select_engine(c, mid18); return true; }
    if (lat9 >= 0 && items <= 3 * simds * 8) { select_engine(c, lat9); return true; }
    if (lat9 >= 0) return false;
  }
  if (!c->eng_ctx[0]) return false;
  {
    // automatic: the latency engine wins while its launch stays within a few wavefronts per SIMD — measured on MI355X
    // (tools/dev/sweep.py, both engines pinned): RangeProofNi n = 2048 (16 lanes per integer) 48 proofs = 3 waves per SIMD:
    // 56 ms against 62 ms, 64 proofs: 70 against 63; NiCorrectKeyProof (2048-bit moduli, 8 lanes) 4096 keys = 5.5 per SIMD:
    // 62 against 70 ms, 8192 keys: 101 against 96; CompositeDLogProof 16384 proofs = 4 per SIMD: 11 against 18 ms
    const uint64_t limbs = mod_bits <= 2048 ? 72 : mod_bits <= 4096 ? 144 : 288;
    const uint64_t lanes = limbs / (uint64_t)c->eng[0]->limbs_per_lane;
    // (round 3, both engines with squarings where their product allows: RangeProofNi n = 2048 32 proofs 40.7 / 37.5 ms against 49.2 / 44.4,
    // 48 proofs 55.3 / 54.9 against 49.5 / 44.2 -> 2.5 waves per SIMD; NiCorrectKeyProof 4096 keys 56.9 against 59.0 ms, 8192 keys 90 against 81)
    const uint64_t half_waves_per_simd = mod_bits <= 2048 ? 12 : 5;
    if (2 * items <= (half_waves_per_simd * 4 * (uint64_t)c->cus * 64) / lanes) { select_engine(c, 0); return true; }
  }
  return false;
}
#define ZKP_ROUTE(c, items, mod_bits, fn, ...)                                              \
  if ((c) && route_latency((c), (items), (mod_bits))) return lat_forward_plain((c), (c)->lat->p_##fn((c)->lat_ctx, __VA_ARGS__));
#define ZKP_ROUTE_ENC(c, items, mod_bits, one_key, fn, ...)                                 \
  if ((c) && route_latency((c), (items), (mod_bits), (one_key))) return lat_forward_plain((c), (c)->lat->p_##fn((c)->lat_ctx, __VA_ARGS__));
#define ZKP_ROUTE_VERIFY(c, items, mod_bits, one_key, fn, ...)                              \
  if ((c) && route_latency((c), (items), (mod_bits), (one_key), true)) return lat_forward_plain((c), (c)->lat->p_##fn((c)->lat_ctx, __VA_ARGS__));
// (diagnostics: only when the caller PINNED the latency engine)
#define ZKP_ROUTE_PINNED(c, fn, ...)                                                        \
  if ((c) && (c)->geometry && engine_with((c), (c)->geometry) >= 0) { select_engine((c), engine_with((c), (c)->geometry)); return lat_forward_plain((c), (c)->lat->p_##fn((c)->lat_ctx, __VA_ARGS__)); }
#else
#define ZKP_ROUTE(c, items, mod_bits, fn, ...)
#define ZKP_ROUTE_ENC(c, items, mod_bits, one_key, fn, ...)
#define ZKP_ROUTE_VERIFY(c, items, mod_bits, one_key, fn, ...)
#define ZKP_ROUTE_PINNED(c, fn, ...)
#endif

static int32_t ensure(zkp_ctx* c, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes) return ZKP_OK;
  if (b.p) HIPCHK(c, hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  b.gen++;
  HIPCHK(c, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return ZKP_OK;
}
// the tag of a constants buffer for a launch of `count` keys into it: the device pointer to hand to the kernel (count == 1 and the cache is
// on), nullptr otherwise; cleared on the stream when the buffer has changed hands since the tag was written
static int32_t setup_tag(zkp_ctx* c, DevBuf& b, uint64_t count, uint32_t** out);

// ---- staging of host buffers --------------------------------------------------------------
struct Stage {
  zkp_ctx* c;
  bool dev;
  std::vector<StageBlock> owned;
  struct Out { void* d; void* h; size_t n; };
  std::vector<Out> outs;
  int32_t st = ZKP_OK;
  Stage(zkp_ctx* c_, uint32_t flags) : c(c_), dev((flags & ZKP_F_DEVICE_PTRS) != 0) {}
  Stage(const Stage&) = delete;
  Stage& operator=(const Stage&) = delete;
  ~Stage() {                       // an error return that skipped finish(): nothing is copied back, the blocks go back to the ctx
    join_side();
    if (owned.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    give_back();
  }
  // An error return between the fork onto the ctx's second stream and the join: the kernels there may still read the staging
  // blocks and the ctx scratch that the next call reuses — wait for them before anything is handed back.
  void join_side() {
    if (c->copy_busy) { (void)hipStreamSynchronize(c->copy); c->copy_busy = false; }
    if (!c->side_busy) return;
    (void)hipStreamSynchronize(c->side);
    c->side_busy = false;
  }
  // The blocks of this call return to the ctx's cache (the next call of the same shape allocates nothing).  The cache is bounded:
  // beyond twice this call's own footprint (at least 64 MiB) the least recently returned blocks are freed, so a long-lived ctx
  // that serves many batch shapes does not pile up device memory it will never use again.
  void give_back() {
    size_t mine = 0;
    for (auto& b : owned) { c->stage_free.push_back(b); mine += b.cap; }
    const size_t keep_from = c->stage_free.size() - owned.size();
    owned.clear();
    size_t total = 0;
    for (auto& b : c->stage_free) total += b.cap;
    const size_t limit = std::max<size_t>(2 * mine, size_t(64) << 20);
    size_t drop = 0;
    while (drop < keep_from && total > limit) { total -= c->stage_free[drop].cap; (void)hipFree(c->stage_free[drop].p); drop++; }
    if (drop) c->stage_free.erase(c->stage_free.begin(), c->stage_free.begin() + (ptrdiff_t)drop);
  }
  // best fit among the cached blocks (no more than twice the size asked for), else a fresh allocation
  void* take(size_t bytes) {
    bytes = std::max<size_t>((bytes + 255) & ~size_t(255), 256);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < c->stage_free.size(); i++) {
      const size_t cap = c->stage_free[i].cap;
      if (cap >= bytes && cap <= 2 * bytes && (best == SIZE_MAX || cap < c->stage_free[best].cap)) best = i;
    }
    StageBlock b{nullptr, bytes};
    if (best != SIZE_MAX) { b = c->stage_free[best]; c->stage_free[best] = c->stage_free.back(); c->stage_free.pop_back(); }
    else if (hipMalloc(&b.p, bytes) != hipSuccess) { st = ZKP_ENOMEM; c->err = "hipMalloc (staging)"; return nullptr; }
    owned.push_back(b);
    return b.p;
  }
  template <class T> const T* in(const T* p, size_t count) {
    if (dev || !p || st) return p;
    void* d = take(count * sizeof(T));
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) { st = ZKP_EDEVICE; c->err = "H2D copy"; }
    return (const T*)d;
  }
  // a host buffer in BOTH memory modes (short byte strings such as a salt)
  template <class T> const T* host_in(const T* p, size_t count) {
    if (!p || st) return nullptr;
    void* d = take(count * sizeof(T));
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) { st = ZKP_EDEVICE; c->err = "H2D copy"; }
    return (const T*)d;
  }
  template <class T> T* out(T* p, size_t count, bool copy_in = false) {
    if (dev || !p || st) return p;
    void* d = take(count * sizeof(T));
    if (!d) return nullptr;
    if (copy_in) { if (hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) { st = ZKP_EDEVICE; c->err = "H2D copy"; } }
    else (void)hipMemsetAsync(d, 0, count * sizeof(T), c->stream);
    outs.push_back({d, (void*)p, count * sizeof(T)});
    return (T*)d;
  }
  int32_t finish() {
    for (auto& o : outs)
      if (!st && hipMemcpyAsync(o.h, o.d, o.n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { st = ZKP_EDEVICE; c->err = "D2H copy"; }
    join_side();
    if (!dev || st) { if (hipStreamSynchronize(c->stream) != hipSuccess && !st) { st = ZKP_EDEVICE; c->err = "stream sync"; } }
    give_back();
    outs.clear();
    return st;
  }
};

// ---- timing ---------------------------------------------------------------------------------
struct TimedRegion {
  zkp_ctx* c; bool on; size_t slot = 0;
  TimedRegion(zkp_ctx* c_, uint64_t modexps) : c(c_), on(c_->timing) {
    if (!on) return;
    if (c->ev_used == c->ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
      c->ev.push_back({a, b});
    }
    slot = c->ev_used++;
    (void)hipEventRecord(c->ev[slot].first, c->stream);
    c->timed_launches++; c->timed_modexps += modexps;
  }
  ~TimedRegion() { if (on) (void)hipEventRecord(c->ev[slot].second, c->stream); }
};

// ---- host-pointer calls of large batches: the copies run under the kernels -------------------------------------------------------
// A host-pointer call copies every input in, runs the kernels, copies every output out, all on one stream.  Proofs are independent and the
// batch is structure-of-arrays, so a block of proof indices is a contiguous slice of every array: such a call CAN be cut into a few blocks,
// block k + 1 copied in and block k - 1 copied out on a second stream while the kernels of block k run (only the first block's inputs and the
// last block's outputs stay exposed).  Measured on MI355X boxes of this pool (round 5, profiles/r05/host_pipeline/): pageable host arrays move
// at ~30 GB/s, so the copies of a 4096-proof call are 25 - 35 ms of 1250 - 1700 ms, while every extra launch of the persistent Enc kernel
// has a tail of its own (a quarter of the batch is 3 - 4 claims per wavefront): cut in 2 / 3 blocks the call is 20 - 45 ms SLOWER than whole.
// So the library does not cut by itself; $ZKP_HOST_CHUNKS=N at ctx create (N equal blocks; 0 = a quarter | the rest | [a quarter]) is for hosts
// whose copies are slow (no large BAR, a remote NUMA node, PCIe Gen3).  What round 4 read as 0.36 s of staging in the host API was the tails
// and first-touch page faults of ITS four-chunk pipeline, fixed there (host/zkproofs.hpp: prove_batch).
struct Piped {
  zkp_ctx* c;
  Stage& s;
  struct Field { char* dev; const char* hin; char* hout; size_t per; };        // per: bytes per proof
  std::vector<Field> f;
  Piped(zkp_ctx* c_, Stage& s_) : c(c_), s(s_) {}
  int32_t streams(size_t blocks) {
    if (!c->copy) HIPCHK(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    while (c->ev_pipe.size() < 2 * blocks) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->ev_pipe.push_back(e);
    }
    return ZKP_OK;
  }
  // an input array of `per` elements per proof: a device block for all of it, copied block by block (h2d)
  template <class T> const T* in(const T* host, size_t per, size_t B) {
    if (!host || s.st) return host;
    char* d = (char*)s.take(per * B * sizeof(T));
    if (!d) return nullptr;
    f.push_back({d, (const char*)host, nullptr, per * sizeof(T)});
    return (const T*)d;
  }
  // an output array: zeroed now (on the compute stream, ahead of every kernel), copied back block by block (d2h)
  template <class T> T* out(T* host, size_t per, size_t B) {
    if (!host || s.st) return host;
    char* d = (char*)s.take(per * B * sizeof(T));
    if (!d) return nullptr;
    (void)hipMemsetAsync(d, 0, per * B * sizeof(T), c->stream);
    f.push_back({d, nullptr, (char*)host, per * sizeof(T)});
    return (T*)d;
  }
  // inputs of proofs [lo, hi) on the copy stream; event `ev` says they are resident
  int32_t h2d(size_t lo, size_t hi, hipEvent_t ev) {
    for (const Field& x : f)
      if (x.hin) HIPCHK(c, hipMemcpyAsync(x.dev + lo * x.per, x.hin + lo * x.per, (hi - lo) * x.per, hipMemcpyHostToDevice, c->copy));
    c->copy_busy = true;
    HIPCHK(c, hipEventRecord(ev, c->copy));
    return ZKP_OK;
  }
  // outputs of proofs [lo, hi) once the kernels behind `done` (recorded on the compute stream) have finished
  int32_t d2h(size_t lo, size_t hi, hipEvent_t done) {
    HIPCHK(c, hipStreamWaitEvent(c->copy, done, 0));
    for (const Field& x : f)
      if (x.hout) HIPCHK(c, hipMemcpyAsync(x.hout + lo * x.per, x.dev + lo * x.per, (hi - lo) * x.per, hipMemcpyDeviceToHost, c->copy));
    c->copy_busy = true;
    return ZKP_OK;
  }
  int32_t finish() {
    const hipError_t e = hipStreamSynchronize(c->copy);
    c->copy_busy = false;
    if (e != hipSuccess) { c->err = "copy stream sync"; return ZKP_EDEVICE; }
    return ZKP_OK;
  }
};

// The blocks of a host-pointer call of B proofs of `rows` rows each (empty: one block, the plain path).  first_small: the exposed copy is
// the first block's input (verify: [1/4, 3/4]); else both ends are exposed (prove: inputs in, ciphertexts and responses out: [1/4, 1/2, 1/4]).
static std::vector<size_t> host_blocks(const zkp_ctx* c, size_t B, size_t rows, bool first_small_only) {
  std::vector<size_t> b;
  if (c->host_chunks < 0 || c->host_chunks == 1) return b;              // the default: one block
  if (c->host_chunks > 1) {
    const size_t n = std::min<size_t>((size_t)c->host_chunks, B);
    for (size_t k = 0; k <= n; k++) b.push_back(B * k / n);
    return n > 1 ? b : std::vector<size_t>();
  }
  // 0: uneven blocks.  first_small_only: the exposed copy is the first block's input (verify: [1/4, 3/4]); else both ends are exposed
  // (prove: inputs in, ciphertexts and responses out: [1/4, 1/2, 1/4])
  if (B * rows < 2048ull * ZKP_SECURITY_PARAMETER) return b;          // below that a block would leave SIMDs idle (one claim per wavefront is 256 proofs)
  const size_t q = (B / 4 + 63) & ~size_t(63);
  if (first_small_only) b = {0, q, B};
  else b = {0, q, B - q, B};
  return b;
}

// ---- geometry helpers -----------------------------------------------------------------------
// lanes per big integer: 72 / 144 / 288 limbs of 29 bits over W limbs per lane
constexpr int GA = 72 / W, GB = 144 / W, GC = 288 / W;
#ifdef ZKP_SPLIT_TU
// compiled in zkp_kernels_keys.hip under another machine-scheduler strategy (see there); launched from here
extern template __global__ void zkp::k_enc<GA, false, false>(EncArgs);
extern template __global__ void zkp::k_enc<GB, false, false>(EncArgs);
extern template __global__ void zkp::k_enc<GC, false, false>(EncArgs);
extern template __global__ void zkp::k_ck_check<GA, false>(CkCheckArgs);
extern template __global__ void zkp::k_ck_check<GB, false>(CkCheckArgs);
#endif
static int group_for_bits(uint32_t mod_bits) { return mod_bits <= 2048 ? GA : mod_bits <= 4096 ? GB : mod_bits <= 8192 ? GC : 0; }

template <int G, class K> static int resident_blocks(zkp_ctx* c, K kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, LdsLayout<G>::BYTES_PER_BLOCK) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu * c->cus;
}

static int32_t setup_tag(zkp_ctx* c, DevBuf& b, uint64_t count, uint32_t** out) {
  *out = nullptr;
  if (count != 1 || !c->key_cache) {                          // several keys (or the cache is off): whatever single-key record the buffer held is gone
    b.gen++;
    return ZKP_OK;
  }
  if (!b.tag) {
    HIPCHK(c, hipMalloc((void**)&b.tag, SETUP_TAG_WORDS * sizeof(uint32_t)));
    HIPCHK(c, hipMemsetAsync(b.tag, 0, SETUP_TAG_WORDS * sizeof(uint32_t), c->stream));
    b.tag_gen = b.gen;
  } else if (b.tag_gen != b.gen) {
    HIPCHK(c, hipMemsetAsync(b.tag, 0, 4, c->stream));
    b.tag_gen = b.gen;
  }
  *out = b.tag;
  return ZKP_OK;
}

template <int G> static int32_t run_setup(zkp_ctx* c, const uint32_t* src, uint64_t stride, int src_words, int square, uint64_t count, DevBuf& buf) {
  using CL = ConstLayout<G>;
  using LL = LdsLayoutFull<G>;
  int32_t st = ensure(c, buf, count * CL::WORDS * sizeof(uint32_t));
  if (st) return st;
  const unsigned blocks = (unsigned)((count + LL::GROUPS_PER_BLOCK - 1) / LL::GROUPS_PER_BLOCK);
  uint32_t* tag = nullptr;
  if ((st = setup_tag(c, buf, count, &tag))) return st;
  hipLaunchKernelGGL(k_setup<G>, dim3(blocks), dim3(LL::THREADS), LL::BYTES_PER_BLOCK, c->stream, src, stride, src_words, square, count, (uint32_t*)buf.p, c->setup_flag, tag);
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}

// "did any set-up since the last call reject its modulus": one word, reduced on the device by k_setup itself
static int32_t clear_setup_flag(zkp_ctx* c) {
  HIPCHK(c, hipMemsetAsync(c->setup_flag, 0, 4, c->stream));
  return ZKP_OK;
}
static int32_t read_setup_flag(zkp_ctx* c, bool* any_bad) {
  HIPCHK(c, hipMemcpyAsync(c->setup_flag_host, c->setup_flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *any_bad = *c->setup_flag_host != 0;
  return ZKP_OK;
}

// sliding-window schedule for a launch-uniform exponent (device resident, ctx scratch slot 15)
static int32_t build_schedule(zkp_ctx* c, const uint32_t* exp_words, uint32_t exp_bits, const uint8_t** out) {
  DevBuf& b = c->scratch[15];
  int32_t st = ensure(c, b, sched_buffer_bytes((int)exp_bits));
  if (st) return st;
  hipLaunchKernelGGL(k_sliding_schedule, dim3(1), dim3(64), 0, c->stream, exp_words, (int)exp_bits, (uint8_t*)b.p);
  HIPCHK(c, hipGetLastError());
  *out = (const uint8_t*)b.p;
  return ZKP_OK;
}

// zeroed work counter for the next k_enc launch (ctx scratch slot 18; one 8-byte slot per launch in flight is enough
// because launches of one ctx are ordered on its stream)
static int32_t fresh_work_counter(zkp_ctx* c, unsigned long long** out) {
  DevBuf& b = c->scratch[18];
  int32_t st = ensure(c, b, 64);
  if (st) return st;
  HIPCHK(c, hipMemsetAsync(b.p, 0, 64, c->stream));
  *out = (unsigned long long*)b.p;
  return ZKP_OK;
}

// the ctx's second stream (work forked from and joined back into c->stream inside one call)
static int32_t side_stream(zkp_ctx* c) {
  if (c->side) return ZKP_OK;
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  return ZKP_OK;
}

template <int G, class K> static int32_t table_for(zkp_ctx* c, K kernel, uint64_t items, unsigned* blocks_out) {
  using LL = LdsLayout<G>;
  const uint64_t need = (items + LL::GROUPS_PER_BLOCK - 1) / LL::GROUPS_PER_BLOCK;
  const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(need, (uint64_t)resident_blocks<G>(c, kernel)));
  *blocks_out = blocks;
  return ensure(c, c->table, (size_t)blocks * LL::GROUPS_PER_BLOCK * TAB * Geo<G>::L * sizeof(uint32_t));
}

// may `items` exponentiations take the pair ladder (two groups of G lanes each)?  Latency engine only, and only while the
// launch with twice the lanes still leaves every SIMD at most one wavefront (tools/dev/pair_sweep.py on MI355X, RangeProofNi
// n = 2048: 8 proofs 24.0 / 22.2 ms either way; 12 proofs 27.2 / 22.8 ms on the window ladder, 34.6 / 29.2 ms on pairs that
// share SIMDs two by two).
constexpr bool PAIR_LADDER_BUILD = ZKP_W <= 9;            // the pair kernels are instantiated in the latency engine only
template <int G> static bool pair_ladder(const zkp_ctx* c, uint64_t items) {
  if (c->enc_form == ZKP_ENC_FORM_ALWAYS) return false;      // (tests that pin the base-n kernels of the latency engine on small batches: they need the window script)
  if constexpr (PAIR_LADDER_BUILD && 2 * G <= 64) return items * 2 * G <= 4ull * (uint64_t)c->cus * 64;
  (void)c; (void)items;
  return false;
}

// k_enc is instantiated per ladder kind: one shared exponent n (sliding-window script) or one key per item (fixed windows).
// The latency engine has a third: the right-to-left ladder on pairs of groups (kernels_modexp.hpp: powm_pair), taken while a
// launch with twice the lanes per item still leaves every SIMD at most one wavefront — the call is then a single chain of
// products per item, and that chain is 14 % shorter.
#if ZKP_HAS_BASEN
// lanes per n-sized integer of the base-n kernels: 72 / 144 limbs over W limbs per lane
constexpr int BN_GA = 72 / W, BN_GB = 144 / W;
#ifdef ZKP_SPLIT_TU
// compiled in zkp_kernels_basen.hip
extern template __global__ void zkp::k_enc_basen<BN_GA>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
extern template __global__ void zkp::k_enc_basen<BN_GB>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
extern template __global__ void zkp::k_enc_basen_keys<BN_GA>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
extern template __global__ void zkp::k_enc_basen_keys<BN_GB>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
extern template __global__ void zkp::k_basen_finish<BN_GA>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*);
extern template __global__ void zkp::k_basen_finish<BN_GB>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*);
extern template __global__ void zkp::k_setup_basen<BN_GA>(const uint32_t*, uint32_t*, uint64_t, uint32_t*, const uint32_t*, uint32_t*);
extern template __global__ void zkp::k_setup_basen<BN_GB>(const uint32_t*, uint32_t*, uint64_t, uint32_t*, const uint32_t*, uint32_t*);
extern template __global__ void zkp::k_expected<2 * BN_GA>(EncArgs, uint32_t*, const uint32_t*);
extern template __global__ void zkp::k_expected<2 * BN_GB>(EncArgs, uint32_t*, const uint32_t*);
extern template __global__ void zkp::k_diag_basen<BN_GA>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
extern template __global__ void zkp::k_diag_basen<BN_GB>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
#endif
constexpr size_t BASEN_DIAG_LDS = 4096;
// Which Paillier launches take the base-n form is a property of the ctx (include/zkp_hip_diag.h: zkp_diag_set_enc_form):
//   AUTO   the library's own rule — every launch that fills the chip (launch_basen below), under one key or under per-proof keys
//   N2     none: every launch stays on the n^2-sized kernels (A/B runs, the parity tests that pin the two forms against each other)
//   SHARED as AUTO, but launches under per-proof keys stay on the n^2-sized kernels
//   ALWAYS also the launches too small to gain from it (the parity tests of the small shapes)
// $ZKP_BASEN (0 | shared | always) presets it when the ctx is created; nothing reads the environment after that.
static int enc_form_from_env() {
  const char* e = std::getenv("ZKP_BASEN");
  if (e && e[0] == '0') return ZKP_ENC_FORM_N2;
  if (e && e[0] == 's') return ZKP_ENC_FORM_SHARED;
  if (e && e[0] == 'a') return ZKP_ENC_FORM_ALWAYS;
  return ZKP_ENC_FORM_AUTO;
}
// base-n constants of `nkeys` keys (G lanes per n-sized integer): k_setup<G> on every n, then k_setup_basen<G>.  Everything stays on the
// stream; whether the keys qualify (odd, long enough, digit sums of M~ within the fast-product bound) is a device word the kernels read:
// c->bn_all_ok is 1 only when every key of the batch did.
template <int G> static int32_t basen_prepare(zkp_ctx* c, const uint32_t* n, uint64_t n_stride, uint64_t nkeys, uint32_t n_bits) {
  int32_t st = run_setup<G>(c, n, n_stride, (int)(n_bits / 32), 0, nkeys, c->bn_ncst);
  if (st) return st;
  if ((st = ensure(c, c->bn_consts, (size_t)(nkeys + 1) * BnConst<G>::STRIDE * sizeof(uint32_t)))) return st;
  // record `nkeys`: all zero — what the groups of keys outside the form compute on (k_enc_basen_keys)
  HIPCHK(c, hipMemsetAsync((char*)c->bn_consts.p + (size_t)nkeys * BnConst<G>::STRIDE * sizeof(uint32_t), 0, BnConst<G>::STRIDE * sizeof(uint32_t), c->stream));
  if ((st = ensure(c, c->bn_flag, 64))) return st;
  HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->bn_flag.p, 1, 2, c->stream));      // word 0: every key of the batch qualified (cleared by k_setup_basen); word 1: the constant 1
  constexpr unsigned GPB = 64 / G;
  uint32_t* tag = nullptr;
  if ((st = setup_tag(c, c->bn_consts, nkeys, &tag))) return st;
  hipLaunchKernelGGL(k_setup_basen<G>, dim3((unsigned)((nkeys + GPB - 1) / GPB)), dim3(64), GPB * BN_SETUP_LDS_WORDS * sizeof(uint32_t), c->stream,
                     (const uint32_t*)c->bn_ncst.p, (uint32_t*)c->bn_consts.p, nkeys, (uint32_t*)c->bn_flag.p, tag ? (const uint32_t*)c->bn_ncst.tag : nullptr, tag);
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}
// Does a launch of the one-wavefront-per-Enc ladder over at most `count` items (`listed`: a verify's work list, of which about three quarters
// exist) fit one wavefront per SIMD together with `hashes` transcript-hash wavefronts?
static bool r2l_one_per_simd(const zkp_ctx* c, uint64_t count, bool listed, uint64_t hashes) {
  const uint64_t simds = 4ull * (uint64_t)c->cus;
  const uint64_t expect = (listed && c->grid_expected) ? expected_items(count) : count;
  return hashes < simds && expect + hashes <= simds;
}
// The base-n launch of an Enc call (GS: lanes per n^2-sized integer of the k_enc launch it stands in for).  It claims work from the SAME
// counter as the k_enc launch that follows it: when the keys qualify it leaves nothing to claim, when they do not it returns at once
// and k_enc runs as before.  Returns false when nothing was launched.
template <int GS> static bool launch_basen(zkp_ctx* c, const EncArgs& a_in, EncArgs* rest) {
  if constexpr (GS != 2 * BN_GA && GS != 2 * BN_GB) { (void)c; (void)a_in; (void)rest; return false; }
  else {
    constexpr int G = GS / 2;
    using BL = BnLds<G>;
    EncArgs a = a_in;
    const int kw = a.n_bits / 32;
    const bool per_key = a.n_stride != 0;
    c->bn_last_g = 0;                                             // (zkp_diag_basen_last: this launch has not taken the form yet)
    const int mode = c->enc_form;
    if (mode == ZKP_ENC_FORM_N2 || (per_key && mode == ZKP_ENC_FORM_SHARED) || a.n_bits != (G == BN_GA ? 2048 : 4096)) return false;
    // The latency engine's smallest calls — up to two wavefronts per SIMD at ONE Enc per wavefront: 8 proofs at n = 2048 — take the five-group
    // right-to-left ladder (kernels_basen_r2l.hpp): half the chain of the pair ladder that served them (the launch behind this one, which then
    // finds nothing to claim).  c->bn_r2l: 0 = never, 1 = the library's rule, 2 = whenever the kernel can take the launch (tests).
    bool r2l_launch = false;
    // (a verify's work list: judged by the items expected — 9 and 10 proofs, 2304 / 2560 by the bound, still fit two wavefronts per SIMD: 14.5 ms
    //  there against 19.2 on the window ladder)
    const uint64_t r2l_items = (a.count_ptr && a.mode == 1 && c->grid_expected) ? std::min<uint64_t>(a.count, expected_items(a.count)) : a.count;
#if ZKP_W == 9
    if constexpr (G == 8)
      r2l_launch = !per_key && c->bn_r2l && a.n_bits == 2048 && (c->bn_r2l == 2 || (mode != ZKP_ENC_FORM_ALWAYS && r2l_items <= 2ull * 4 * (uint64_t)c->cus));
#endif
    if (!per_key && !a.sched && !r2l_launch) return false;       // (a shared key whose launch takes the pair ladder of the latency engine)
    if (per_key && a.n_stride != (uint64_t)kw) return false;
    if (a.mode == 0 && ((a.m_words > kw) || (a.r_words > kw))) return false;
    // A launch whose n^2-sized wavefronts (64 / GS items each) all find a SIMD of their own is a single chain per wavefront either way,
    // and the base-n chain is the longer one (27.4 M against 22.3 M VALU instructions per claim, for twice the items): measured at
    // n = 2048, 64 proofs: prove 51 -> 64 ms, verify 45 -> 61 ms; from 96 proofs on: 88 -> 65 ms (profiles/r04/basen/midsize_sweep.jsonl)
    if (mode != ZKP_ENC_FORM_ALWAYS && !r2l_launch && a.count <= 4ull * (uint64_t)c->cus * (64 / GS)) return false;
    uint64_t nkeys = 1;
    if (per_key) {
      const uint64_t items = (a.mode == 0 && a.half) ? a.half : a.count;
      nkeys = a.mode == 1 ? a.count / (2 * (uint64_t)a.ef) : (items + a.items_per_key - 1) / a.items_per_key;      // mode 1: a.count = 2 * batch * ef
      if (nkeys == 0) return false;
    }
    // (any failure below — out of memory for the form's tables — leaves the launch to the n^2-sized kernel behind this one: not an error)
    auto give_up = [&]() { c->err.clear(); *rest = a_in; return false; };      // (`rest`: nothing was taken off the launch behind this one)
    if (basen_prepare<G>(c, a.n, a.n_stride, nkeys, (uint32_t)a.n_bits)) return give_up();
    int& per_cu = c->bn_occ[per_key ? 1 : 0][G == BN_GB ? 1 : 0];     // per ctx and per kernel: the two differ in registers, and contexts run on threads of their own
    if (!per_cu) {
      const hipError_t e = per_key ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_enc_basen_keys<G>, 256, BL::BYTES_PER_BLOCK)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_enc_basen<G>, 256, BL::BYTES_PER_BLOCK);
      if (e != hipSuccess || per_cu < 1) per_cu = 1;
    }
    const uint64_t need = (a.count + BL::GROUPS_PER_BLOCK - 1) / BL::GROUPS_PER_BLOCK;
    uint64_t grid = std::min<uint64_t>(need, (uint64_t)per_cu * c->cus);
    // A verify's work list (a.count_ptr) holds 128 + the Open rows of every proof — about three quarters — of the 2 Enc per row the launch is
    // sized for.  A workgroup claims until nothing is left, so the grid only has to hold the items EXPECTED: when those need fewer workgroups
    // per compute unit than the bound, the launch stops there (40 proofs on the latency engine: 320 workgroups of 32 Enc for 7680 items gave
    // a quarter of the units a second workgroup, and three calls of six took 27 ... 37 ms instead of 26; 65 ... 81 proofs on the mid engine:
    // one round, 37.3 ms — profiles/r06/expected_items/).  expected_items: 3 % above the mean.
    if (a.count_ptr && a.mode == 1 && c->grid_expected) {
      const uint64_t need_e = (expected_items(a.count) + BL::GROUPS_PER_BLOCK - 1) / BL::GROUPS_PER_BLOCK;
      const uint64_t per_cu_e = (need_e + (uint64_t)c->cus - 1) / (uint64_t)c->cus;
      grid = std::min<uint64_t>(grid, per_cu_e * (uint64_t)c->cus);
    }
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, grid);
    const size_t entries = per_key ? BN_KEYS_TAB_ENTRIES : BN_TAB_ENTRIES;
    if (!r2l_launch && ensure(c, c->bn_table, (size_t)blocks * BL::GROUPS_PER_BLOCK * entries * 2 * Geo<G>::L * sizeof(uint32_t))) return give_up();
    if (ensure(c, c->bn_raw, (size_t)a.count * 2 * Geo<G>::L * sizeof(uint32_t))) return give_up();
    // the word that says "this launch runs in base-n form": the key's own flag under one key; under per-proof keys the launch always runs
    // (the constant 1 behind the batch's flag) and partitions its items key by key: those of keys the form does not take go on a list that
    // the n^2-sized launch behind this one works off (`rest`: remapped items, a counter of its own)
    const uint32_t* ok = per_key ? (const uint32_t*)c->bn_flag.p + 1 : (const uint32_t*)c->bn_consts.p + BnConst<G>::OFF_OK;
    if (per_key) {
      if (ensure(c, c->bn_left, 16 + (size_t)a.count * sizeof(uint32_t))) return give_up();
      if (hipMemsetAsync(c->bn_left.p, 0, 16, c->stream) != hipSuccess) return give_up();
      a.left_count = (unsigned long long*)c->bn_left.p;
      a.left_list = (uint32_t*)((char*)c->bn_left.p + 16);
      rest->remap = a.left_list; rest->remap_count = a.left_count;
      rest->work_counter = a.work_counter + 1;                     // (fresh_work_counter zeroes 64 bytes: the second slot)
    }
    const bool products = a.mode == 1 || (a.mode == 2 && a.cipher_x);
    if (products) {
      if (ensure(c, c->bn_expected, (size_t)a.count * 2 * kw * sizeof(uint32_t))) return give_up();
      using LS = LdsLayout<GS>;
      const uint64_t eneed = (a.count + LS::GROUPS_PER_BLOCK - 1) / LS::GROUPS_PER_BLOCK;
      const unsigned eblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(eneed, 2ull * c->cus));
      hipLaunchKernelGGL(k_expected<GS>, dim3(eblocks), dim3(256), LS::BYTES_PER_BLOCK, c->stream, a, (uint32_t*)c->bn_expected.p, ok);
    }
    c->bn_last_g = G; c->bn_last_per_key = per_key; c->bn_last_r2l = r2l_launch;
#if ZKP_W == 9
    if (r2l_launch) {
      if constexpr (G == 8) {
        const unsigned waves = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(a.count, r2l_items < a.count ? std::max<uint64_t>(r2l_items, 2ull * 4 * (uint64_t)c->cus) : 8ull * 4 * (uint64_t)c->cus));
        // five wavefronts per Enc (36 lanes x 2 limbs each, k_enc_basen_r2l5) while the launch leaves a CU to every Enc: one proof;
        // one wavefront of five groups of 12 lanes x 6 limbs beyond; 8 lanes x 9 limbs only when pinned (A/B runs)
        const int lanes = c->bn_r2l_lanes ? c->bn_r2l_lanes : (a.count <= ZKP_R2L5_ITEMS_PER_CU * (uint64_t)c->cus ? 36 : 12);
        c->bn_last_r2l_lanes = lanes;
        if (lanes == 36) {
          // (with the call's transcript hashes aboard: one workgroup each in front of the Enc workgroups, the launch as a whole within one workgroup per
          //  compute unit — an Enc workgroup claims items until none is left, so fewer of them than items is only a longer loop for some)
          RangeHashArgs h{};
          uint64_t enc_wgs = std::max<uint64_t>(1, std::min<uint64_t>(a.count, 4ull * (uint64_t)c->cus));
          size_t dyn = 0;
          if (c->fuse_hash && c->fuse_hash->batch < (uint64_t)c->cus) {
            h = *c->fuse_hash;
            c->fuse_hash_taken = true;
            enc_wgs = std::max<uint64_t>(1, std::min<uint64_t>(enc_wgs, (uint64_t)c->cus - h.batch));
            dyn = (size_t)hw_lds_words((int)h.kw) * sizeof(uint32_t);
          }
          hipLaunchKernelGGL(k_enc_basen_r2l5, dim3((unsigned)(h.batch + enc_wgs)), dim3(320), dyn, c->stream, a, (const uint32_t*)c->bn_consts.p, (uint32_t*)c->bn_raw.p, h);
        }
        else {
          // one wavefront per Enc: the call's transcript hashes aboard — at 64 blocks per batch while the launch as a whole stays within one
          // wavefront per SIMD (2 - 5 proofs), at 16 (6.6 KB of LDS on every workgroup) beyond (6 - 8 proofs: two wavefronts per SIMD)
          // (a verify's work list holds 128 + the Open rows of every proof — about 192 — of the 256 Enc per proof the launch is sized for: while the
          //  items EXPECTED fit one wavefront per SIMD the grid stops there — five proofs: 960 of 1280 — and a wavefront claims until none is left)
          RangeHashArgs h{};
          uint64_t enc_wgs = waves;
          size_t dyn = 0;
          const uint64_t simds = 4ull * (uint64_t)c->cus;
          const uint64_t hashes = c->fuse_hash ? c->fuse_hash->batch : 0;
          if (r2l_one_per_simd(c, a.count, a.count_ptr != nullptr, hashes)) {
            enc_wgs = std::max<uint64_t>(1, std::min<uint64_t>(enc_wgs, simds - hashes));
            if (c->fuse_hash) {
              h = *c->fuse_hash;
              h.wave_blocks = HW_BLOCKS;
              c->fuse_hash_taken = true;
              dyn = (size_t)hw_lds_words<HW_BLOCKS>((int)h.kw) * sizeof(uint32_t);
            }
          } else if (c->fuse_hash && hashes < simds && r2l_items <= 2 * simds) {
            // two wavefronts per SIMD: the hashes among them (the first workgroups) instead of beside them — with the small LDS footprint
            h = *c->fuse_hash;
            h.wave_blocks = 16;
            c->fuse_hash_taken = true;
            enc_wgs = std::max<uint64_t>(1, std::min<uint64_t>(enc_wgs, 2 * simds - hashes));
            dyn = (size_t)hw_lds_words<16>((int)h.kw) * sizeof(uint32_t);
          }
          const dim3 grid((unsigned)(h.batch + enc_wgs));
          if (lanes == 8) hipLaunchKernelGGL(k_enc_basen_r2l<9>, grid, dim3(64), dyn, c->stream, a, (const uint32_t*)c->bn_consts.p, (uint32_t*)c->bn_raw.p, h);
          else hipLaunchKernelGGL(k_enc_basen_r2l<6>, grid, dim3(64), dyn, c->stream, a, (const uint32_t*)c->bn_consts.p, (uint32_t*)c->bn_raw.p, h);
        }
      }
    } else
#endif
    if (per_key)
      hipLaunchKernelGGL(k_enc_basen_keys<G>, dim3(blocks), dim3(256), BL::BYTES_PER_BLOCK, c->stream, a, (const uint32_t*)c->bn_consts.p, (const uint32_t*)c->bn_consts.p + (size_t)nkeys * BnConst<G>::STRIDE, (uint32_t*)c->bn_table.p,
                         (uint32_t*)c->bn_raw.p);
    else
      hipLaunchKernelGGL(k_enc_basen<G>, dim3(blocks), dim3(256), BL::BYTES_PER_BLOCK, c->stream, a, (const uint32_t*)c->bn_consts.p, (uint32_t*)c->bn_table.p,
                         (uint32_t*)c->bn_raw.p);
    const unsigned fblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(need, 8ull * c->cus));
    hipLaunchKernelGGL(k_basen_finish<G>, dim3(fblocks), dim3(256), BL::BYTES_PER_BLOCK, c->stream, a, (const uint32_t*)c->bn_consts.p, ok, per_key ? 1 : 0,
                       (const uint32_t*)c->bn_raw.p, (const uint32_t*)c->bn_expected.p, (const uint32_t*)c->bn_consts.p + (size_t)nkeys * BnConst<G>::STRIDE);
    return true;
  }
}
#endif

// Will launch_basen<GS> hand a verify launch of at most `count` Enc under the key of this call to k_enc_basen_r2l5 (five wavefronts per Enc, one
// workgroup per compute unit) or to k_enc_basen_r2l at one wavefront per SIMD — the launches that take the call's transcript hashes aboard?  The same conditions as below, asked ahead of the launch by range_verify_impl; a launch that then does not take
// the call's transcript hashes aboard (out of memory for the form's buffers) leaves them to a launch of their own.
template <int GS> static bool basen_r2l_takes_hashes(const zkp_ctx* c, uint64_t n_stride, uint32_t n_bits, uint64_t count, uint64_t hashes) {
#if ZKP_W == 9
  if constexpr (GS == 2 * BN_GA) {
    const int mode = c->enc_form;
    if (mode == ZKP_ENC_FORM_N2 || n_stride != 0 || n_bits != 2048 || !c->bn_r2l) return false;
    const uint64_t items = c->grid_expected ? std::min<uint64_t>(count, expected_items(count)) : count;      // (as launch_basen judges a verify's work list)
    if (!(c->bn_r2l == 2 || (mode != ZKP_ENC_FORM_ALWAYS && items <= 2ull * 4 * (uint64_t)c->cus))) return false;
    const bool five = c->bn_r2l_lanes ? c->bn_r2l_lanes == 36 : count <= ZKP_R2L5_ITEMS_PER_CU * (uint64_t)c->cus;
    return five || r2l_one_per_simd(c, count, true, hashes) || (hashes < 4ull * (uint64_t)c->cus && items <= 8ull * (uint64_t)c->cus);  // k_enc_basen_r2l5, or one wavefront per Enc at one or two per SIMD
  }
#endif
  (void)c; (void)n_stride; (void)n_bits; (void)count; (void)hashes;
  return false;
}
template <int G> static void launch_k_enc(zkp_ctx* c, unsigned blocks, const EncArgs& a_in) {
  using LL = LdsLayout<G>;
  EncArgs a = a_in;
#if ZKP_HAS_BASEN
  (void)launch_basen<G>(c, a_in, &a);      // (under per-proof keys `a` now names the items the base-n launch leaves to this one)
#endif
  if constexpr (PAIR_LADDER_BUILD && 2 * G <= 64) {
    if (pair_ladder<G>(c, a.count)) {                     // (a.count: an upper bound when the count is device resident, verify work list)
      const unsigned pair_blocks = (unsigned)std::max<uint64_t>(1, (a.count + LL::GROUPS_PER_BLOCK / 2 - 1) / (LL::GROUPS_PER_BLOCK / 2));
      hipLaunchKernelGGL((k_enc<G, false, true>), dim3(pair_blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a);
      return;
    }
  }
  if (a.sched) hipLaunchKernelGGL((k_enc<G, true>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a);
  else hipLaunchKernelGGL((k_enc<G, false>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a);
}

// ---- ctx ------------------------------------------------------------------------------------
static int32_t ctx_create(int32_t device_id, hipStream_t stream, bool own_stream, zkp_ctx** out) {
  if (!out) return ZKP_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return ZKP_EDEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return ZKP_EDEVICE;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device_id) != hipSuccess) return ZKP_EDEVICE;
  if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) return ZKP_EDEVICE;   // kernels are built for gfx950 only
  zkp_ctx* c = new zkp_ctx();
  c->device = device_id;
  c->cus = p.multiProcessorCount;
  c->last_geometry = W;
#if ZKP_HAS_BASEN
  c->enc_form = enc_form_from_env();
#endif
  if (const char* hc = std::getenv("ZKP_HOST_CHUNKS")) c->host_chunks = std::atoi(hc);
  if (const char* kc = std::getenv("ZKP_KEY_CACHE")) c->key_cache = std::atoi(kc) != 0;
  if (const char* sp = std::getenv("ZKP_SPLIT")) c->split_calls = std::atoi(sp) != 0;
  if (const char* rl = std::getenv("ZKP_R2L")) c->bn_r2l = std::atoi(rl);
  if (const char* fh = std::getenv("ZKP_FUSE_HASH")) c->fuse_hash_on = std::atoi(fh);
  if (const char* ge = std::getenv("ZKP_GRID_EXPECTED")) c->grid_expected = std::atoi(ge);
  if (const char* rl = std::getenv("ZKP_R2L_LANES")) { const int v = std::atoi(rl); c->bn_r2l_lanes = (v == 8 || v == 12 || v == 36) ? v : 0; }
  c->owns_stream = own_stream;
  c->stream = stream;
  if (own_stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ZKP_EDEVICE; }
  if (hipHostMalloc((void**)&c->pinned_counts, zkp_ctx::PINNED_SLOTS * sizeof(unsigned long long)) != hipSuccess) c->pinned_counts = nullptr;
  if (hipMalloc((void**)&c->setup_flag, 64) != hipSuccess || hipHostMalloc((void**)&c->setup_flag_host, 64) != hipSuccess ||
      hipMemset(c->setup_flag, 0, 64) != hipSuccess) { (void)zkp_ctx_destroy(c); return ZKP_EDEVICE; }
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++)
    if (const LatEngine* eng = secondary_engine(k)) {
      const int32_t st = eng->p_zkp_ctx_create_on_stream(device_id, (void*)c->stream, &c->eng_ctx[k]);
      if (st) { (void)zkp_ctx_destroy(c); return st; }
      c->eng[k] = eng;
      if (!c->lat_ctx) { c->lat = eng; c->lat_ctx = c->eng_ctx[k]; }
    }
  if (const char* g = std::getenv("ZKP_GEOMETRY")) {                        // testing aid: the same as zkp_ctx_set_geometry
    const int32_t st = zkp_ctx_set_geometry(c, std::atoi(g));
    if (st) { (void)zkp_ctx_destroy(c); return st; }
  }
#endif
  *out = c;
  return ZKP_OK;
}

extern "C" int32_t zkp_ctx_create(int32_t device_id, zkp_ctx** out) try {
  return ctx_create(device_id, nullptr, true, out);
} ZKP_CATCH(nullptr)

extern "C" int32_t zkp_ctx_create_on_stream(int32_t device_id, void* hip_stream, zkp_ctx** out) try {
  return ctx_create(device_id, (hipStream_t)hip_stream, false, out);
} ZKP_CATCH(nullptr)

extern "C" int32_t zkp_ctx_set_geometry(zkp_ctx* c, int32_t limbs_per_lane) try {
  if (!c) return ZKP_EINVAL;
  if (limbs_per_lane != 0 && limbs_per_lane != W && engine_with(c, limbs_per_lane) < 0) {
    c->err = "zkp_ctx_set_geometry: no engine with " + std::to_string(limbs_per_lane) + " limbs per lane is loaded";
    return ZKP_EINVAL;
  }
  c->geometry = limbs_per_lane;
  return ZKP_OK;
} ZKP_CATCH(c)
extern "C" int32_t zkp_ctx_last_geometry(zkp_ctx* c) { return c ? c->last_geometry : 0; }
extern "C" int32_t zkp_ctx_latency_limbs_per_lane(zkp_ctx* c) { return (c && c->eng[0]) ? c->eng[0]->limbs_per_lane : 0; }

extern "C" int32_t zkp_ctx_destroy(zkp_ctx* c) try {
  if (!c) return ZKP_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->split_ctx) { (void)hipStreamSynchronize(c->split_stream); (void)c->eng[0]->p_zkp_ctx_destroy(c->split_ctx); }
  if (c->split_stream) (void)hipStreamDestroy(c->split_stream);
  for (hipEvent_t e : c->ev_split) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_ctx_destroy(c->eng_ctx[k]);
  for (DevBuf* b : {&c->consts, &c->consts2, &c->table, &c->bn_ncst, &c->bn_consts, &c->bn_table, &c->bn_expected, &c->bn_raw, &c->bn_flag, &c->bn_left}) { if (b->p) (void)hipFree(b->p); if (b->tag) (void)hipFree(b->tag); }
  for (auto& b : c->scratch) if (b.p) (void)hipFree(b.p);
  for (auto& e : c->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (c->pinned_counts) (void)hipHostFree(c->pinned_counts);
  for (auto& b : c->stage_free) (void)hipFree(b.p);
  if (c->setup_flag) (void)hipFree(c->setup_flag);
  if (c->setup_flag_host) (void)hipHostFree(c->setup_flag_host);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
  for (hipEvent_t e : c->ev_pipe) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ZKP_OK;
} ZKP_CATCH(c)

extern "C" int32_t zkp_ctx_release_staging(zkp_ctx* c) try {
  if (!c) return ZKP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& b : c->stage_free) (void)hipFree(b.p);
  c->stage_free.clear();
  // the base-n form's per-launch areas (window tables of pairs: 1.4 GB under one key, 2.5 GB under per-proof keys; raw pairs; Mask-row
  // products) are sized by the largest launch so far and rebuilt on demand
  for (DevBuf* b : {&c->bn_table, &c->bn_raw, &c->bn_expected}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
  if (c->split_ctx) (void)c->eng[0]->p_zkp_ctx_release_staging(c->split_ctx);
  for (int k = 0; k < 2; k++)
    if (c->eng_ctx[k]) { const int32_t st = c->eng[k]->p_zkp_ctx_release_staging(c->eng_ctx[k]); if (st) { c->err = c->eng[k]->p_zkp_last_error_string(c->eng_ctx[k]); return st; } }
  return ZKP_OK;
} ZKP_CATCH(c)

extern "C" const char* zkp_backend_name(void) { return "hip-gfx950"; }
extern "C" int32_t zkp_build_limbs_per_lane(void) { return W; }
#if ZKP_HAS_BASEN
extern "C" int32_t zkp_diag_basen_engine(void) { return (ZKP_BN_ASM && W == 36) ? 1 : 0; }
#else
extern "C" int32_t zkp_diag_basen_engine(void) { return 0; }
#endif
extern "C" const char* zkp_last_error_string(zkp_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
extern "C" void* zkp_ctx_stream(zkp_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int32_t zkp_ctx_synchronize(zkp_ctx* c) try {
  if (!c) return ZKP_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ZKP_OK;
} ZKP_CATCH(c)

// diagnostic: a known amount of table traffic (k_table_traffic) for the calibration of the PMC byte counters
extern "C" int32_t zkp_diag_table_traffic(zkp_ctx* c, int32_t mode, int32_t passes, uint64_t* out_bytes) try {
  if (!c || passes < 1 || (mode != 0 && mode != 1)) return ZKP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  constexpr int G = GB;
  unsigned blocks = 0;
  int32_t st = table_for<G>(c, k_enc<G, true>, ~0ull >> 8, &blocks);      // the resident grid of the Paillier kernels and its table
  if (st) return st;
  if ((st = ensure(c, c->scratch[18], 64))) return st;
  {
    TimedRegion tr(c, 0);
    hipLaunchKernelGGL(k_table_traffic<G>, dim3(blocks), dim3(256), 0, c->stream, (uint32_t*)c->table.p, (int)mode, (int)passes, (uint32_t*)c->scratch[18].p);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (out_bytes) *out_bytes = (uint64_t)blocks * (256 / G) * TABS * Geo<G>::L * sizeof(uint32_t) * (uint64_t)passes;
  return ZKP_OK;
} ZKP_CATCH(c)

#if ZKP_HAS_BASEN
// diagnostic: one base-n operation on raw limbs (kernels_basen.hpp: k_diag_basen); tests/test_gpu_basen.py checks it against tests/basen_model.py
extern "C" int32_t zkp_diag_basen(zkp_ctx* c, uint32_t n_bits, const uint32_t* n, int32_t op, const uint32_t* xa, const uint32_t* xb, const uint32_t* ya,
                                  const uint32_t* yb, uint32_t* out) try {
  if (!c || !n || !out || (n_bits != 2048 && n_bits != 4096) || op < 0 || op > 3) return ZKP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  ZKP_ROUTE_PINNED(c, zkp_diag_basen, n_bits, n, op, xa, xb, ya, yb, out)
  const int G = n_bits == 2048 ? BN_GA : BN_GB;
  const size_t L = (size_t)G * W, kw = n_bits / 32;
  Stage s(c, 0);
  const uint32_t* dn = s.in(n, kw);
  const uint32_t* dxa = s.in(xa, L); const uint32_t* dxb = s.in(xb, L); const uint32_t* dya = s.in(ya, L); const uint32_t* dyb = s.in(yb, L);
  uint32_t* dout = s.out(out, 4 * L + 4);
  uint32_t* dscr = (uint32_t*)s.take(2 * L * sizeof(uint32_t));
  int32_t st = s.st;
  if (!st) st = G == BN_GA ? basen_prepare<BN_GA>(c, dn, 0, 1, n_bits) : basen_prepare<BN_GB>(c, dn, 0, 1, n_bits);
  if (!st) {
    if (G == BN_GA) hipLaunchKernelGGL(k_diag_basen<BN_GA>, dim3(1), dim3(64), BASEN_DIAG_LDS, c->stream, (const uint32_t*)c->bn_consts.p, (int)op, dxa, dxb, dya, dyb, dout, dscr);
    else hipLaunchKernelGGL(k_diag_basen<BN_GB>, dim3(1), dim3(64), BASEN_DIAG_LDS, c->stream, (const uint32_t*)c->bn_consts.p, (int)op, dxa, dxb, dya, dyb, dout, dscr);
    if (hipGetLastError() != hipSuccess) st = ZKP_EDEVICE;
  }
  const int32_t fin = s.finish();
  return st ? st : fin;
} ZKP_CATCH(c)
// diagnostic: did the most recent shared-key Paillier launch of this ctx run in base-n form?  out_lanes: lanes per n-sized integer of that
// launch (0: there was none), out_qualified: the key passed k_setup_basen (else the n^2-sized kernel did the work)
extern "C" int32_t zkp_diag_basen_last(zkp_ctx* c, int32_t* out_lanes, uint32_t* out_qualified) try {
  if (!c || !out_lanes || !out_qualified) return ZKP_EINVAL;
#ifndef ZKP_SECONDARY_ENGINE
  if (c->lat_ctx && c->last_geometry == c->lat->limbs_per_lane)        // the most recent call ran on the latency engine: its twin ctx knows
    return lat_forward_plain(c, c->lat->p_zkp_diag_basen_last(c->lat_ctx, out_lanes, out_qualified));
#endif
  *out_lanes = c->bn_last_g; *out_qualified = 0;
  if (!c->bn_last_g) return ZKP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t off = (c->bn_last_g == BN_GA ? BnConst<BN_GA>::OFF_OK : BnConst<BN_GB>::OFF_OK) * sizeof(uint32_t);
  const void* word = c->bn_last_per_key ? c->bn_flag.p : (const void*)((const char*)c->bn_consts.p + off);
  HIPCHK(c, hipMemcpyAsync(c->setup_flag_host + 8, word, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out_qualified = c->setup_flag_host[8];
  return ZKP_OK;
} ZKP_CATCH(c)
#else
extern "C" int32_t zkp_diag_basen_last(zkp_ctx* c, int32_t* out_lanes, uint32_t* out_qualified) {
  if (!c || !out_lanes || !out_qualified) return ZKP_EINVAL;
  *out_lanes = 0; *out_qualified = 0;
  return ZKP_OK;
}
extern "C" int32_t zkp_diag_basen(zkp_ctx* c, uint32_t, const uint32_t*, int32_t, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*) {
  if (c) c->err = "zkp_diag_basen: the base-n kernels are built into the throughput engine only";
  return ZKP_EINVAL;
}
#endif

// which Paillier launches take the base-n form (include/zkp_hip_diag.h); the twin ctx of the latency engine follows
extern "C" int32_t zkp_diag_set_enc_form(zkp_ctx* c, int32_t form) try {
  if (!c || form < ZKP_ENC_FORM_AUTO || form > ZKP_ENC_FORM_ALWAYS) return ZKP_EINVAL;
  c->enc_form = form;
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_diag_set_enc_form(c->eng_ctx[k], form);
#endif
  return ZKP_OK;
} ZKP_CATCH(c)
extern "C" int32_t zkp_diag_enc_form(zkp_ctx* c) { return c ? c->enc_form : -1; }
extern "C" int32_t zkp_diag_last_host_blocks(zkp_ctx* c) { return c ? c->last_host_blocks : -1; }
extern "C" int32_t zkp_diag_mid_limbs_per_lane(zkp_ctx* c) { return (c && c->eng[1]) ? c->eng[1]->limbs_per_lane : 0; }
// the one-Enc-per-wavefront ladder of the latency engine (kernels_basen_r2l.hpp): 0 = never, 1 = the library's rule, 2 = whenever it can run
extern "C" int32_t zkp_diag_set_r2l(zkp_ctx* c, int32_t mode) try {
  if (!c || mode < 0 || mode > 2) return ZKP_EINVAL;
  c->bn_r2l = mode;
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_diag_set_r2l(c->eng_ctx[k], mode);
#endif
  return ZKP_OK;
} ZKP_CATCH(c)
// The transcript hashes of a verify of 1 ... 8 proofs as workgroups of its Enc launch (k_enc_basen_r2l5 / k_enc_basen_r2l; csrc/zkp_api_proofs.inc range_verify_impl): on / off;
// did the most recent verify call of the ctx run that way?
extern "C" int32_t zkp_diag_set_fuse_hash(zkp_ctx* c, int32_t on) try {
  if (!c) return ZKP_EINVAL;
  c->fuse_hash_on = on != 0;
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_diag_set_fuse_hash(c->eng_ctx[k], on);
#endif
  return ZKP_OK;
} ZKP_CATCH(c)
extern "C" int32_t zkp_diag_last_fused_hash(zkp_ctx* c) {
  if (!c) return -1;
#ifndef ZKP_SECONDARY_ENGINE
  if (c->lat_ctx && c->last_geometry == c->lat->limbs_per_lane) return c->lat->p_zkp_diag_last_fused_hash(c->lat_ctx);
#endif
  return c->fuse_hash_taken ? 1 : 0;
}
// ... and its lane geometry: 0 = the library's rule, 36 = five wavefronts per Enc, 12 / 8 = one wavefront of five 12- / 8-lane groups
extern "C" int32_t zkp_diag_set_r2l_lanes(zkp_ctx* c, int32_t lanes) try {
  if (!c || (lanes != 0 && lanes != 8 && lanes != 12 && lanes != 36)) return ZKP_EINVAL;
  c->bn_r2l_lanes = lanes;
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_diag_set_r2l_lanes(c->eng_ctx[k], lanes);
#endif
  return ZKP_OK;
} ZKP_CATCH(c)
extern "C" int32_t zkp_diag_r2l_lanes_last(zkp_ctx* c) {
  if (!c) return -1;
#ifndef ZKP_SECONDARY_ENGINE
  if (c->lat_ctx && c->last_geometry == c->lat->limbs_per_lane) return c->lat->p_zkp_diag_r2l_lanes_last(c->lat_ctx);
#endif
  return (c->bn_last_g && c->bn_last_r2l) ? c->bn_last_r2l_lanes : 0;
}
// the per-key constants kept across calls (setup_tag): on / off for this ctx and its secondary engines
extern "C" int32_t zkp_diag_set_key_cache(zkp_ctx* c, int32_t on) try {
  if (!c) return ZKP_EINVAL;
  c->key_cache = on != 0;
#ifndef ZKP_SECONDARY_ENGINE
  for (int k = 0; k < 2; k++) if (c->eng_ctx[k]) (void)c->eng[k]->p_zkp_diag_set_key_cache(c->eng_ctx[k], on);
#endif
  return ZKP_OK;
} ZKP_CATCH(c)
// the tag header of one constants buffer of the engine the most recent routed call ran on (which: 0 = n^2 / modexp constants, 1 = the
// second set (mod n), 2 = the n-sized record behind the base-n form, 3 = the base-n record): out[0] = 1 valid, out[1] = 1 when the last
// set-up launch into it returned early, out[2] = its epoch (set-ups that computed).  All zero when the buffer has no tag.  Synchronises.
extern "C" int32_t zkp_diag_key_cache_state(zkp_ctx* c, int32_t which, uint32_t* out) try {
  if (!c || !out || which < 0 || which > 3) return ZKP_EINVAL;
#ifndef ZKP_SECONDARY_ENGINE
  if (c->lat_ctx && c->last_geometry == c->lat->limbs_per_lane) return lat_forward_plain(c, c->lat->p_zkp_diag_key_cache_state(c->lat_ctx, which, out));
#endif
  out[0] = out[1] = out[2] = 0;
  DevBuf* b = which == 0 ? &c->consts : which == 1 ? &c->consts2 : which == 2 ? &c->bn_ncst : &c->bn_consts;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!b->tag || b->tag_gen != b->gen) return ZKP_OK;
  uint32_t head[SETUP_TAG_HEAD];
  HIPCHK(c, hipMemcpy(head, b->tag, sizeof(head), hipMemcpyDeviceToHost));
  out[0] = head[0] == SETUP_TAG_MAGIC; out[1] = head[4]; out[2] = head[5];
  return ZKP_OK;
} ZKP_CATCH(c)
// calls of 65 ... 96 proofs as two concurrent calls on two engines (range_split): on / off; how many proofs of the most recent
// RangeProofNi call went to the latency engine beside the mid engine (0: the call was not split)
extern "C" int32_t zkp_diag_set_split(zkp_ctx* c, int32_t on) { if (!c) return ZKP_EINVAL; c->split_calls = on != 0; return ZKP_OK; }
extern "C" int32_t zkp_diag_last_split(zkp_ctx* c) { return c ? c->last_split : -1; }
// did the most recent Paillier launch of this ctx run on it?
extern "C" int32_t zkp_diag_r2l_last(zkp_ctx* c) {
  if (!c) return -1;
#ifndef ZKP_SECONDARY_ENGINE
  if (c->lat_ctx && c->last_geometry == c->lat->limbs_per_lane) return c->lat->p_zkp_diag_r2l_last(c->lat_ctx);
#endif
  return (c->bn_last_g && c->bn_last_r2l) ? 1 : 0;
}

extern "C" int32_t zkp_timing_reset(zkp_ctx* c, int32_t enable) try {
  if (!c) return ZKP_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->timing = enable != 0;
  c->ev_used = 0; c->timed_launches = 0; c->timed_modexps = 0; c->pinned_used = 0;
  for (int k = 0; k < 2; k++)
    if (c->eng_ctx[k]) { const int32_t st = c->eng[k]->p_zkp_timing_reset(c->eng_ctx[k], enable); if (st) { c->err = c->eng[k]->p_zkp_last_error_string(c->eng_ctx[k]); return st; } }
  // (the tail of a split call — csrc/zkp_api_proofs.inc range_split_run — runs on a second ctx of the latency engine: its share counts too)
  if (c->split_ctx) { const int32_t st = c->eng[0]->p_zkp_timing_reset(c->split_ctx, enable); if (st) { c->err = c->eng[0]->p_zkp_last_error_string(c->split_ctx); return st; } }
  return ZKP_OK;
} ZKP_CATCH(c)

extern "C" int32_t zkp_timing_get(zkp_ctx* c, double* ms, uint64_t* launches, uint64_t* modexps) try {
  if (!c) return ZKP_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double total = 0;
  for (size_t i = 0; i < c->ev_used; i++) {
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev[i].first, c->ev[i].second));
    total += t;
  }
  uint64_t extra = 0, lat_launches = 0, lat_modexps = 0;
  for (size_t i = 0; i < c->pinned_used; i++) extra += c->pinned_counts[i];
  for (int k = 0; k < 2; k++)
    if (c->eng_ctx[k]) {                             // the twin contexts' share of the calls since the reset
      double e_ms = 0; uint64_t e_launches = 0, e_modexps = 0;
      const int32_t st = c->eng[k]->p_zkp_timing_get(c->eng_ctx[k], &e_ms, &e_launches, &e_modexps);
      if (st) { c->err = c->eng[k]->p_zkp_last_error_string(c->eng_ctx[k]); return st; }
      total += e_ms; lat_launches += e_launches; lat_modexps += e_modexps;
    }
  if (c->split_ctx) {
    double e_ms = 0; uint64_t e_launches = 0, e_modexps = 0;
    const int32_t st = c->eng[0]->p_zkp_timing_get(c->split_ctx, &e_ms, &e_launches, &e_modexps);
    if (st) { c->err = c->eng[0]->p_zkp_last_error_string(c->split_ctx); return st; }
    total += e_ms; lat_launches += e_launches; lat_modexps += e_modexps;
  }
  if (ms) *ms = total;
  if (launches) *launches = c->timed_launches + lat_launches;
  if (modexps) *modexps = c->timed_modexps + extra + lat_modexps;
  return ZKP_OK;
} ZKP_CATCH(c)

// ---- L1 primitives --------------------------------------------------------------------------
// modexp over constants that are already set up in c->consts (per-item when const_stride != 0)
template <int G>
static int32_t modexp_core(zkp_ctx* c, uint32_t exp_bits, uint64_t count, const uint32_t* base, const uint32_t* exp, uint64_t exp_stride,
                           bool per_item_mod, uint32_t* out, int io_words, int out_words = 0) {
  using CL = ConstLayout<G>;
  using LL = LdsLayout<G>;
  unsigned blocks = 0;
  int32_t st;
  if constexpr (PAIR_LADDER_BUILD && 2 * G <= 64) {
    if (pair_ladder<G>(c, count)) {                       // a few items on the latency engine: right-to-left ladder on pairs of groups
      unsigned long long* wcp = nullptr;
      if ((st = fresh_work_counter(c, &wcp))) return st;
      ModexpArgs pa{base, exp, exp_stride, (const uint32_t*)c->consts.p, per_item_mod ? (uint64_t)CL::WORDS : 0, out, nullptr, count, (int)exp_bits, io_words,
                    out_words ? out_words : io_words, nullptr, wcp, {}, 0};
      const unsigned pair_blocks = (unsigned)std::max<uint64_t>(1, (count + LL::GROUPS_PER_BLOCK / 2 - 1) / (LL::GROUPS_PER_BLOCK / 2));
      {
        TimedRegion tr(c, count);
        hipLaunchKernelGGL((k_modexp<G, false, false, true>), dim3(pair_blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, pa);
      }
      HIPCHK(c, hipGetLastError());
      return ZKP_OK;
    }
  }
  // One exponent for the whole call: the sliding-window ladder (5 % fewer products) on the latency engine, whose calls are
  // chains; on the throughput engine the fixed-window kernel, reading the one exponent with stride 0, is as fast or faster
  // (65 536 exponentiations, 4096 / 2048-bit moduli: 171.1 / 46.4 ms against 172.4 / 49.7 ms, tools/dev/perf_ladders.py: k_modexp's
  // sliding variant keeps scratch accesses in its product loops, k_enc's does not) — one kernel less to build and to keep fast.
  constexpr bool SLIDING_MODEXP = !COL_NEEDS_CARE;
  if ((st = table_for<G>(c, k_modexp<G, false>, count, &blocks))) return st;
  const uint8_t* sched = nullptr;
  if (SLIDING_MODEXP && exp_stride == 0 && (st = build_schedule(c, exp, exp_bits, &sched))) return st;
  unsigned long long* wc = nullptr;
  if ((st = fresh_work_counter(c, &wc))) return st;
  ModexpArgs a{base, exp, exp_stride, (const uint32_t*)c->consts.p, per_item_mod ? (uint64_t)CL::WORDS : 0, out, (uint32_t*)c->table.p, count, (int)exp_bits, io_words, out_words ? out_words : io_words, sched, wc, {}, 0};
  {
    TimedRegion tr(c, count);
    bool launched = false;
    if constexpr (SLIDING_MODEXP) {
      if (a.sched) { hipLaunchKernelGGL((k_modexp<G, true>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a); launched = true; }
    }
    if (!launched) hipLaunchKernelGGL((k_modexp<G, false>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a);
  }
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}

// Up to three exponentiations per item (per-item exponents, the same per-item moduli already set up in c->consts) in ONE
// launch: ModexpArgs::more.  Longest exponent first, so that the long chains start first.
struct ModexpCall { uint32_t exp_bits; const uint32_t* base; const uint32_t* exp; uint64_t exp_stride; uint32_t* out; int io_words; int out_words; };
template <int G>
static int32_t modexp_multi(zkp_ctx* c, uint64_t count, bool per_item_mod, std::initializer_list<ModexpCall> calls_in) {
  using CL = ConstLayout<G>;
  using LL = LdsLayout<G>;
  std::vector<ModexpCall> calls(calls_in);
  std::stable_sort(calls.begin(), calls.end(), [](const ModexpCall& x, const ModexpCall& y) { return x.exp_bits > y.exp_bits; });
  if (calls.empty() || calls.size() > 3) { c->err = "modexp_multi: 1..3 segments"; return ZKP_EINVAL; }
  for (const ModexpCall& k : calls) if (k.exp_stride == 0) { c->err = "modexp_multi: per-item exponents only"; return ZKP_EINVAL; }
  int32_t st;
  bool pair = false;
  if constexpr (PAIR_LADDER_BUILD && 2 * G <= 64) pair = pair_ladder<G>(c, count * calls.size());
  const uint64_t IPW = pair ? 64 / (2 * G) : 64 / G;      // items per claim, as the kernel counts them
  const uint64_t claims = (count + IPW - 1) / IPW * IPW * calls.size();
  if (!pair && (calls.size() == 1 || claims > (uint64_t)resident_blocks<G>(c, k_modexp<G, false, true>) * LL::GROUPS_PER_BLOCK)) {
    // the launch fills the GPU anyway: one launch per exponentiation on the single-segment kernel (no segment bookkeeping in
    // its product loops)
    for (const ModexpCall& k : calls)
      if ((st = modexp_core<G>(c, k.exp_bits, count, k.base, k.exp, k.exp_stride, per_item_mod, k.out, k.io_words, k.out_words))) return st;
    return ZKP_OK;
  }
  unsigned blocks = 0;
  if (pair) blocks = (unsigned)std::max<uint64_t>(1, (claims + LL::GROUPS_PER_BLOCK / 2 - 1) / (LL::GROUPS_PER_BLOCK / 2));
  else if ((st = table_for<G>(c, k_modexp<G, false, true>, claims, &blocks))) return st;
  unsigned long long* wc = nullptr;
  if ((st = fresh_work_counter(c, &wc))) return st;
  const ModexpCall& f = calls[0];
  ModexpArgs a{f.base, f.exp, f.exp_stride, (const uint32_t*)c->consts.p, per_item_mod ? (uint64_t)CL::WORDS : 0, f.out, (uint32_t*)c->table.p, count,
               (int)f.exp_bits, f.io_words, f.out_words ? f.out_words : f.io_words, nullptr, wc, {}, (int)calls.size() - 1};
  for (size_t k = 1; k < calls.size(); k++) {
    const ModexpCall& m = calls[k];
    a.more[k - 1] = {m.base, m.exp, m.exp_stride, m.out, count, (int)m.exp_bits, m.io_words, m.out_words ? m.out_words : m.io_words};
  }
  {
    TimedRegion tr(c, count * calls.size());
    bool launched = false;
    if constexpr (PAIR_LADDER_BUILD && 2 * G <= 64) {
      if (pair) { hipLaunchKernelGGL((k_modexp<G, false, true, true>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a); launched = true; }
    }
    if (!launched) hipLaunchKernelGGL((k_modexp<G, false, true>), dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, a);
  }
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}

template <int G>
static int32_t modexp_impl(zkp_ctx* c, uint32_t exp_bits, uint64_t count, const uint32_t* base, const uint32_t* exp, uint64_t exp_stride,
                           const uint32_t* mod, uint64_t mod_stride, uint32_t* out) {
  using LL = LdsLayout<G>;
  const uint64_t nmod = mod_stride ? count : 1;
  int32_t st = clear_setup_flag(c);
  if (st) return st;
  if ((st = run_setup<G>(c, mod, mod_stride, LL::NW, 0, nmod, c->consts))) return st;
  bool bad = false;
  if ((st = modexp_core<G>(c, exp_bits, count, base, exp, exp_stride, mod_stride != 0, out, LL::NW))) return st;
  if ((st = read_setup_flag(c, &bad))) return st;
  if (bad) { c->err = "even or trivial modulus in batch (outputs of those items are untouched)"; return ZKP_ENONCANONICAL; }
  return ZKP_OK;
}

extern "C" int32_t zkp_modexp_batch(zkp_ctx* c, uint32_t mod_bits, uint32_t exp_bits, uint64_t count, const uint32_t* base,
                                    const uint32_t* exp, uint64_t exp_stride, const uint32_t* mod, uint64_t mod_stride, uint32_t* out,
                                    uint32_t flags) try {
  ZKP_ROUTE(c, count, mod_bits, zkp_modexp_batch, mod_bits, exp_bits, count, base, exp, exp_stride, mod, mod_stride, out, flags)
  if (!c) return ZKP_EINVAL;
  if (count == 0) return ZKP_OK;
  if (!base || !exp || !mod || !out || (mod_bits != 2048 && mod_bits != 4096 && mod_bits != 8192) || exp_bits == 0 || exp_bits % 32 ||
      exp_bits > mod_bits || count > (1ull << 40) || (exp_stride && exp_stride < exp_bits / 32) || (mod_stride && mod_stride < mod_bits / 32)) { c->err = "zkp_modexp_batch: invalid argument"; return ZKP_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t L = mod_bits / 32, E = exp_bits / 32;
  Stage s(c, flags);
  const uint32_t* dbase = s.in(base, count * L);
  const uint32_t* dexp = s.in(exp, exp_stride ? count * exp_stride : E);
  const uint32_t* dmod = s.in(mod, mod_stride ? count * mod_stride : L);
  uint32_t* dout = s.out(out, count * L);
  int32_t st = s.st;
  if (!st) {
    switch (group_for_bits(mod_bits)) {
      case GA: st = modexp_impl<GA>(c, exp_bits, count, dbase, dexp, exp_stride, dmod, mod_stride, dout); break;
      case GB: st = modexp_impl<GB>(c, exp_bits, count, dbase, dexp, exp_stride, dmod, mod_stride, dout); break;
      default: st = modexp_impl<GC>(c, exp_bits, count, dbase, dexp, exp_stride, dmod, mod_stride, dout); break;
    }
  }
  const int32_t fin = s.finish();
  return st ? st : fin;
} ZKP_CATCH(c)

template <int G>
static int32_t modmul_impl(zkp_ctx* c, uint64_t count, const uint32_t* a, const uint32_t* b, const uint32_t* mod, uint64_t mod_stride, uint32_t* out) {
  using CL = ConstLayout<G>;
  using LL = LdsLayout<G>;
  const uint64_t nmod = mod_stride ? count : 1;
  int32_t st = clear_setup_flag(c);
  if (st) return st;
  if ((st = run_setup<G>(c, mod, mod_stride, LL::NW, 0, nmod, c->consts))) return st;
  bool bad = false;
  ModmulArgs args{a, b, (const uint32_t*)c->consts.p, mod_stride ? (uint64_t)CL::WORDS : 0, out, count, LL::NW};
  const unsigned blocks = (unsigned)((count + LL::GROUPS_PER_BLOCK - 1) / LL::GROUPS_PER_BLOCK);
  hipLaunchKernelGGL(k_modmul<G>, dim3(blocks), dim3(256), LL::BYTES_PER_BLOCK, c->stream, args);
  HIPCHK(c, hipGetLastError());
  if ((st = read_setup_flag(c, &bad))) return st;
  if (bad) { c->err = "even or trivial modulus in batch"; return ZKP_ENONCANONICAL; }
  return ZKP_OK;
}

extern "C" int32_t zkp_modmul_batch(zkp_ctx* c, uint32_t mod_bits, uint64_t count, const uint32_t* a, const uint32_t* b, const uint32_t* mod,
                                    uint64_t mod_stride, uint32_t* out, uint32_t flags) try {
  if (!c) return ZKP_EINVAL;
  if (count == 0) return ZKP_OK;
  if (!a || !b || !mod || !out || (mod_bits != 2048 && mod_bits != 4096 && mod_bits != 8192) || count > (1ull << 31) || (mod_stride && mod_stride < mod_bits / 32)) { c->err = "zkp_modmul_batch: invalid argument"; return ZKP_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t L = mod_bits / 32;
  Stage s(c, flags);
  const uint32_t* da = s.in(a, count * L);
  const uint32_t* db = s.in(b, count * L);
  const uint32_t* dm = s.in(mod, mod_stride ? count * mod_stride : L);
  uint32_t* dout = s.out(out, count * L);
  int32_t st = s.st;
  if (!st) {
    switch (group_for_bits(mod_bits)) {
      case GA: st = modmul_impl<GA>(c, count, da, db, dm, mod_stride, dout); break;
      case GB: st = modmul_impl<GB>(c, count, da, db, dm, mod_stride, dout); break;
      default: st = modmul_impl<GC>(c, count, da, db, dm, mod_stride, dout); break;
    }
  }
  const int32_t fin = s.finish();
  return st ? st : fin;
} ZKP_CATCH(c)

// Paillier contexts: modulus n^2, group size from 2*n_bits
template <int G>
static int32_t enc_setup(zkp_ctx* c, uint32_t n_bits, const uint32_t* n, uint64_t n_stride, uint64_t nkeys) {
  return run_setup<G>(c, n, n_stride, (int)(n_bits / 32), 1, nkeys, c->consts);
}

template <int G>
static int32_t enc_impl(zkp_ctx* c, uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride, const uint32_t* m, const uint32_t* r,
                        uint32_t* out) {
  using CL = ConstLayout<G>;
  const uint64_t nkeys = n_stride ? count : 1;
  int32_t st = enc_setup<G>(c, n_bits, n, n_stride, nkeys);
  if (st) return st;
  unsigned blocks = 0;
  if ((st = table_for<G>(c, k_enc<G, true>, count, &blocks))) return st;
  EncArgs a{};
  a.n = n; a.n_stride = n_stride; a.consts = (const uint32_t*)c->consts.p; a.const_stride = n_stride ? (uint64_t)CL::WORDS : 0;
  a.table = (uint32_t*)c->table.p; a.count = count; a.n_bits = (int)n_bits; a.mode = 0;
  a.m = m; a.r = r; a.out = out; a.items_per_key = n_stride ? 1 : count;
  if (n_stride == 0 && !pair_ladder<G>(c, a.count) && (st = build_schedule(c, n, n_bits, &a.sched))) return st;
  if ((st = fresh_work_counter(c, &a.work_counter))) return st;
  {
    TimedRegion tr(c, count);
    launch_k_enc<G>(c, blocks, a);
  }
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}

extern "C" int32_t zkp_paillier_enc_batch(zkp_ctx* c, uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride, const uint32_t* m,
                                          const uint32_t* r, uint32_t* out_c, uint32_t flags) try {
  ZKP_ROUTE_ENC(c, count, 2 * n_bits, n_stride == 0, zkp_paillier_enc_batch, n_bits, count, n, n_stride, m, r, out_c, flags)
  if (!c) return ZKP_EINVAL;
  if (count == 0) return ZKP_OK;
  if (!n || !m || !r || !out_c || (n_bits != 1024 && n_bits != 2048 && n_bits != 4096) || count > (1ull << 40) || (n_stride && n_stride < n_bits / 32)) { c->err = "zkp_paillier_enc_batch: invalid argument"; return ZKP_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t kw = n_bits / 32;
  Stage s(c, flags);
  const uint32_t* dn = s.in(n, n_stride ? count * n_stride : kw);
  const uint32_t* dm = s.in(m, count * kw);
  const uint32_t* dr = s.in(r, count * kw);
  uint32_t* dout = s.out(out_c, count * 2 * kw);
  int32_t st = s.st;
  if (!st) {
    switch (group_for_bits(2 * n_bits)) {
      case GA: st = enc_impl<GA>(c, n_bits, count, dn, n_stride, dm, dr, dout); break;
      case GB: st = enc_impl<GB>(c, n_bits, count, dn, n_stride, dm, dr, dout); break;
      default: st = enc_impl<GC>(c, n_bits, count, dn, n_stride, dm, dr, dout); break;
    }
  }
  const int32_t fin = s.finish();
  return st ? st : fin;
} ZKP_CATCH(c)

// Enc-and-compare (CorrectOpening::verify_opening, correct_opening.rs:17-30; the verifier's equality tests range_proof.rs:280-298,324-337)
template <int G>
static int32_t enc_check_impl(zkp_ctx* c, uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride, const uint32_t* m, const uint32_t* r,
                              const uint32_t* exp_or_a, const uint32_t* mulc_b, uint8_t* out_ok) {
  using CL = ConstLayout<G>;
  const uint64_t nkeys = n_stride ? count : 1;
  int32_t st = enc_setup<G>(c, n_bits, n, n_stride, nkeys);
  if (st) return st;
  unsigned blocks = 0;
  if ((st = table_for<G>(c, k_enc<G, true>, count, &blocks))) return st;
  EncArgs a{};
  a.n = n; a.n_stride = n_stride; a.consts = (const uint32_t*)c->consts.p; a.const_stride = n_stride ? (uint64_t)CL::WORDS : 0;
  a.table = (uint32_t*)c->table.p; a.count = count; a.n_bits = (int)n_bits; a.mode = 2;
  a.m = m; a.r = r; a.items_per_key = n_stride ? 1 : count;
  a.c1 = exp_or_a; a.cipher_x = mulc_b; a.verdict = out_ok;
  if (n_stride == 0 && !pair_ladder<G>(c, a.count) && (st = build_schedule(c, n, n_bits, &a.sched))) return st;
  if ((st = fresh_work_counter(c, &a.work_counter))) return st;
  {
    TimedRegion tr(c, count);
    launch_k_enc<G>(c, blocks, a);
  }
  HIPCHK(c, hipGetLastError());
  return ZKP_OK;
}

extern "C" int32_t zkp_paillier_enc_check_batch(zkp_ctx* c, uint32_t n_bits, uint64_t count, const uint32_t* n, uint64_t n_stride, const uint32_t* m,
                                                const uint32_t* r, const uint32_t* mulc_a, const uint32_t* mulc_b, const uint32_t* expected,
                                                uint8_t* out_ok, uint32_t flags) try {
  ZKP_ROUTE_ENC(c, count, 2 * n_bits, n_stride == 0, zkp_paillier_enc_check_batch, n_bits, count, n, n_stride, m, r, mulc_a, mulc_b, expected, out_ok, flags)
  if (!c) return ZKP_EINVAL;
  if (count == 0) return ZKP_OK;
  const bool product = mulc_a || mulc_b;
  if (!n || !m || !r || !out_ok || (n_bits != 1024 && n_bits != 2048 && n_bits != 4096) || count > (1ull << 40) || (n_stride && n_stride != n_bits / 32) ||
      (product ? (!mulc_a || !mulc_b || expected) : !expected)) { c->err = "zkp_paillier_enc_check_batch: invalid argument"; return ZKP_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t kw = n_bits / 32;
  Stage s(c, flags);
  const uint32_t* dn = s.in(n, n_stride ? count * n_stride : kw);
  const uint32_t* dm = s.in(m, count * kw);
  const uint32_t* dr = s.in(r, count * kw);
  const uint32_t* da = s.in(product ? mulc_a : expected, count * 2 * kw);
  const uint32_t* db = s.in(mulc_b, count * 2 * kw);
  uint8_t* dok = s.out(out_ok, count);
  int32_t st = s.st;
  if (!st) {
    switch (group_for_bits(2 * n_bits)) {
      case GA: st = enc_check_impl<GA>(c, n_bits, count, dn, n_stride, dm, dr, da, db, dok); break;
      case GB: st = enc_check_impl<GB>(c, n_bits, count, dn, n_stride, dm, dr, da, db, dok); break;
      default: st = enc_check_impl<GC>(c, n_bits, count, dn, n_stride, dm, dr, da, db, dok); break;
    }
  }
  const int32_t fin = s.finish();
  return st ? st : fin;
} ZKP_CATCH(c)

#include "zkp_api_proofs.inc"
#include "zkp_api_mul.inc"
#include "zkp_api_serde.inc"
#ifndef ZKP_SECONDARY_ENGINE
#include "zkp_api_multi.inc"
#endif
