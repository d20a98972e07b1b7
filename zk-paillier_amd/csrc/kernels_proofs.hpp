// kernels_proofs.hpp — proof-level device code around the L1 kernels: Fiat-Shamir transcript
// hashing, challenge-bit driven response selection (prove) and work-list planning (verify)
// for RangeProofNi, plus the NiCorrectKeyProof and CompositeDLogProof checks.
#pragma once
#include "kernels_modexp.hpp"
#include "kernels_gcd.hpp"
#include "sha256_dev.hpp"

namespace zkp {

// ---- small word-array helpers (per thread, little-endian words in global memory) ------------
__device__ __forceinline__ int cmp_words(const uint32_t* a, const uint32_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    const uint32_t x = a[i], y = b[i];
    if (x != y) return x > y ? 1 : -1;
  }
  return 0;
}

// bit_vec::BitVec::from_bytes indexing (range_proof.rs:221,225,267,272): MSB first
__device__ __forceinline__ int challenge_bit(const uint8_t* e, uint32_t i) { return (e[i >> 3] >> (7 - (i & 7))) & 1; }

// ------------------------------------------------------------------------------------------
// Fiat-Shamir challenge of RangeProofNi (range_proof_ni.rs:58-61 / 89-92 / 110-113):
//   e = to_bytes(from_bytes(SHA256(to_bytes(n) || to_bytes(c1[0..EF)) || to_bytes(c2[0..EF)))))
// one thread per proof.  Also derives T = floor(range/3) and 2T (range_proof.rs:219-220, 264-265).
struct RangeHashArgs {
  const uint32_t* n; uint64_t n_stride;
  const uint32_t* c1; const uint32_t* c2; const uint32_t* range;
  uint32_t kw, ef; uint64_t batch;
  uint8_t* e;          // [B][32] left aligned
  uint8_t* e_len;      // [B]
  uint32_t* third;     // [B][kw]
  uint32_t* two_thirds;// [B][kw]
  uint8_t* verdict;    // [B] verify: ACCEPT or MALFORMED to start with; prove: status 0 / MALFORMED
  uint8_t ok_value;    // value written when the challenge is long enough
  const uint8_t* e_in; const uint8_t* e_len_in;   // externally supplied challenge (interactive protocol): no hashing
  uint32_t wave_blocks;   // hashes that travel with an Enc launch (kernels_basen_r2l.hpp): blocks per batch of the wave hash, 16 or 64 (0 = 64)
};

// One wavefront per 64 proofs, one proof per lane (SHA-256 is sequential per proof).  The operands are read with
// wave-cooperative coalesced loads (512 B per instruction) into an LDS tile of 64 rows (row stride 2kw+1 words: the
// lanes then walk their own rows conflict-free), instead of 64 lanes each chasing its own 4-byte stream through HBM.
constexpr int HASH_PROOFS = 64;

__device__ __forceinline__ void hash_stage_value(uint32_t* tile, int row_stride, const uint32_t* base, uint64_t proof_stride_words,
                                                  int nwords, int nproofs, int lane) {
  // tile[p][w] = base[p * proof_stride_words + w].  32 loads in flight per wait: the loop is bound by HBM latency, not bandwidth
  wave_lds_fence();
  for (int w0 = 0; w0 < nwords; w0 += 64) {
    const int w = w0 + lane;
    const bool inw = w < nwords;
    for (int p0 = 0; p0 < HASH_PROOFS; p0 += 32) {
      uint32_t t[32];
#pragma unroll
      for (int q = 0; q < 32; q++) t[q] = (inw && p0 + q < nproofs) ? base[(uint64_t)(p0 + q) * proof_stride_words + w] : 0u;
#pragma unroll
      for (int q = 0; q < 32; q++) if (inw && p0 + q < nproofs) tile[(p0 + q) * row_stride + w] = t[q];
    }
  }
  wave_lds_fence();
}

#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(HASH_PROOFS) k_range_hash(RangeHashArgs a) {
  extern __shared__ __align__(16) uint32_t hash_lds[];
  const int lane = threadIdx.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * HASH_PROOFS;
  const uint64_t b = b0 + lane;
  const bool live = b < a.batch;
  const int nproofs = (int)((a.batch - b0) < HASH_PROOFS ? (a.batch - b0) : HASH_PROOFS);
  const int kw = (int)a.kw, row_stride = 2 * kw + 1;
  uint32_t* shabuf = hash_lds;                         // [16][64]
  uint32_t* tile = hash_lds + 16 * HASH_PROOFS;        // [64][2kw+1]
  const uint32_t* row = tile + lane * row_stride;
  int elen;
  uint8_t* e = a.e + b * 32;
  if (a.e_in) {                                        // wave-uniform: the whole launch either hashes or copies
    if (!live) return;
    for (int i = 0; i < 32; i++) e[i] = a.e_in[b * 32 + i];
    elen = a.e_len_in[b];
  } else {
    Sha256 s;
    s.init(shabuf + lane, HASH_PROOFS);
    hash_stage_value(tile, row_stride, a.n + b0 * a.n_stride, a.n_stride, kw, nproofs, lane);
    if (live) s.put_bigint(row, kw);
    const uint64_t pstride = (uint64_t)a.ef * 2 * kw;    // words between consecutive proofs in c1 / c2
    for (int half = 0; half < 2; half++) {
      const uint32_t* cbase = (half ? a.c2 : a.c1) + b0 * pstride;
      for (uint32_t i = 0; i < a.ef; i++) {
        hash_stage_value(tile, row_stride, cbase + (uint64_t)i * 2 * kw, pstride, 2 * kw, nproofs, lane);
        if (live) s.put_bigint(row, 2 * kw);
      }
    }
    if (!live) return;
    uint32_t d[8];
    s.finish(d);
    uint8_t db[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { db[4 * i] = d[i] >> 24; db[4 * i + 1] = d[i] >> 16; db[4 * i + 2] = d[i] >> 8; db[4 * i + 3] = d[i]; }
    int lead = 0;
    while (lead < 31 && db[lead] == 0) lead++;      // BigInt round trip drops leading zero bytes; zero -> "00"
    for (int i = 0; i < 32; i++) e[i] = (i + lead < 32) ? db[i + lead] : 0;
    elen = 32 - lead;
  }
  a.e_len[b] = (uint8_t)elen;
  // bits_of_e[i] for i < EF must exist, otherwise the reference panics (index out of bounds)
  a.verdict[b] = ((uint32_t)elen * 8 >= a.ef) ? a.ok_value : (uint8_t)ZKP_VERDICT_MALFORMED;
  if (!a.range) return;                            // challenge only (zkp_range_challenge_batch)
  // T = range div_floor 3, 2T
  const uint32_t* rg = a.range + b * a.kw;
  uint32_t* t1 = a.third + b * a.kw;
  uint32_t* t2 = a.two_thirds + b * a.kw;
  uint64_t rem = 0;
  for (int w = (int)a.kw - 1; w >= 0; w--) {
    const uint64_t cur = (rem << 32) | rg[w];
    t1[w] = (uint32_t)(cur / 3);
    rem = cur % 3;
  }
  uint32_t carry = 0;
  for (uint32_t w = 0; w < a.kw; w++) { const uint32_t v = t1[w]; t2[w] = (v << 1) | carry; carry = v >> 31; }
}
#endif

// ------------------------------------------------------------------------------------------
// The same challenge, ONE WAVEFRONT PER PROOF: for calls of a few proofs, where the one-lane-per-proof kernel above leaves 63
// lanes of its wavefront idle for the ~2 000 chained compressions of a transcript.  SHA-256's rounds are a serial chain, its
// message schedule is not: the lanes assemble the byte stream together (every value is staged into LDS once, its words
// funnel-shifted into stream position by the pending bytes), each lane expands the schedule of one of 64 consecutive blocks
// (W[r] + K[r] into LDS), and then all lanes walk the 64 x 64 rounds in lockstep reading those sums as broadcasts — about half
// the instructions of a compression leave the serial path (9.5 -> ~5 ms per 131 KB transcript).
// NB = blocks per batch: 64 — one schedule per lane, 22.8 KB of LDS at n = 2048 — wherever the hash has a launch or a compute unit to itself;
// 16 (6.6 KB; the schedules of a batch on a quarter of the lanes: ~4 % more instructions per transcript) where its LDS rides on every
// workgroup of an Enc launch of two wavefronts per SIMD (k_enc_basen_r2l: 5 - 8 proofs).
constexpr int HW_BLOCKS = 64;
constexpr int HW_KW_STRIDE = 68;                   // 64 sums per block, padded: 16-byte aligned rows for ds_read_b128
__host__ __device__ constexpr int hw_pidx(int s) { return s + (s >> 4); }      // stream word -> LDS slot (a block's 16 words stay apart in the banks)
template <int NB = HW_BLOCKS> __host__ __device__ constexpr int hw_kw_offset(int kw) { return (hw_pidx(16 * NB + 2 * kw + 8) + 4) & ~3; }   // 16-byte aligned
template <int NB = HW_BLOCKS> __host__ __device__ constexpr int hw_lds_words(int kw) { return hw_kw_offset<NB>(kw) + NB * HW_KW_STRIDE + (2 * kw + 4); }

struct WaveShaState { uint32_t h[8]; };

// Compress the first nblk (<= 64) blocks of the chunk buffer.  One out-of-line copy (two unrolled SHA bodies per call site in
// one function is more than the register allocator of this compiler survives); the LDS areas are derived from the kernel's
// dynamic shared array here so that they stay LDS pointers (a pointer handed through memory becomes a flat one).
template <int NB>
__device__ __noinline__ WaveShaState hw_compress(WaveShaState st, int nblk, int kw, int lane) {
  extern __shared__ __align__(16) uint32_t hash_lds[];
  uint32_t* chunk = hash_lds;
  uint32_t* kwbuf = hash_lds + hw_kw_offset<NB>(kw);
  wave_lds_fence();
  if (lane < nblk) {                                          // this lane's block: schedule + round constants
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = chunk[hw_pidx(16 * lane + i)];
    uint32_t* out = kwbuf + lane * HW_KW_STRIDE;
    // (16 at a time: with all 64 round constants as literals the iterative-ilp scheduler of this compiler crashes the register allocator)
#pragma unroll 16
    for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) wi = w[i];
      else {
        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        wi = w[i & 15] + Sha256::xor3(Sha256::ror(w15, 7), Sha256::ror(w15, 18), w15 >> 3) + w[(i - 7) & 15] +
             Sha256::xor3(Sha256::ror(w2, 17), Sha256::ror(w2, 19), w2 >> 10);
        w[i & 15] = wi;
      }
      out[i] = wi + SHA_K[i];
    }
  }
  wave_lds_fence();
#pragma unroll 1
  for (int j = 0; j < nblk; j++) {                            // the serial chain, identical on every lane
    const uint4* kp = reinterpret_cast<const uint4*>(kwbuf + j * HW_KW_STRIDE);
    uint32_t a = st.h[0], b = st.h[1], c = st.h[2], d = st.h[3], e = st.h[4], f = st.h[5], g = st.h[6], hh = st.h[7];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint4 k4 = kp[q];
      const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t S1 = Sha256::xor3(Sha256::ror(e, 6), Sha256::ror(e, 11), Sha256::ror(e, 25));
        const uint32_t t1 = hh + S1 + Sha256::ch(e, f, g) + kk[r];
        const uint32_t S0 = Sha256::xor3(Sha256::ror(a, 2), Sha256::ror(a, 13), Sha256::ror(a, 22));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + Sha256::maj(b, c, d);      // (b, c, d: a, b, c of this round, already shifted)
      }
    }
    st.h[0] += a; st.h[1] += b; st.h[2] += c; st.h[3] += d; st.h[4] += e; st.h[5] += f; st.h[6] += g; st.h[7] += hh;
  }
  wave_lds_fence();
  return st;
}

template <int NB>
struct WaveShaT {
  WaveShaState st;
  uint32_t* chunk;      // stream words not yet compressed (padded index)
  uint32_t* row;        // one staged value + two words of head room
  int wpos;             // words in chunk
  uint32_t pend; int npend;
  uint64_t nbytes;
  int lane, kw;

  __device__ __forceinline__ void init(uint32_t* lds, int kw_, int lane_) {
    st.h[0] = 0x6a09e667; st.h[1] = 0xbb67ae85; st.h[2] = 0x3c6ef372; st.h[3] = 0xa54ff53a;
    st.h[4] = 0x510e527f; st.h[5] = 0x9b05688c; st.h[6] = 0x1f83d9ab; st.h[7] = 0x5be0cd19;
    chunk = lds;
    row = lds + hw_kw_offset<NB>(kw_) + NB * HW_KW_STRIDE;
    wpos = 0; pend = 0; npend = 0; nbytes = 0; lane = lane_; kw = kw_;
  }

  // compress the first nblk blocks and move the rest of the chunk down
  __device__ __forceinline__ void flush(int nblk) {
    st = hw_compress<NB>(st, nblk, kw, lane);
    const int rest = wpos - 16 * nblk;                       // <= 2kw + 8 words
    uint32_t t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { const int q = lane + 64 * k; t[k] = q < rest ? chunk[hw_pidx(16 * nblk + q)] : 0u; }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 5; k++) { const int q = lane + 64 * k; if (q < rest) chunk[hw_pidx(q)] = t[k]; }
    wpos = rest;
    wave_lds_fence();
  }

  // append the minimal big-endian bytes of a value (nwords <= 2kw little-endian words at src; zero -> one 00 byte)
  __device__ __forceinline__ void put_bigint(const uint32_t* __restrict__ src, int nwords) {
    wave_lds_fence();
    unsigned long long nz[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int w = lane + 64 * k;
      const uint32_t x = w < nwords ? src[w] : 0u;
      if (w < nwords + 2) row[w] = x;
      nz[k] = __ballot(x != 0);
    }
    int top = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) if (nz[k]) top = 64 * k + 63 - __builtin_clzll(nz[k]);
    wave_lds_fence();
    const uint32_t tw = row[top];
    const int kb = tw == 0 ? 1 : (4 - (__clz(tw) >> 3));        // bytes of the top word
    // E = pend || value as one little-endian number: the pending bytes sit right above the value's top byte
    if (npend && lane == 0) {
      if (kb < 4) { row[top] = tw | (pend << (8 * kb)); row[top + 1] = kb + npend > 4 ? pend >> (8 * (4 - kb)) : 0u; }
      else row[top + 1] = pend;
    }
    wave_lds_fence();
    const int elen = 4 * top + kb + npend, nfull = elen >> 2, rem = elen & 3, sh = 8 * rem;
    for (int q = lane; q < nfull; q += 64) {                    // stream word q = the 4 bytes of E below byte offset elen - 4q
      const int u = elen - 4 - 4 * q;
      const uint32_t lo = row[u >> 2], hi = row[(u >> 2) + 1];
      chunk[hw_pidx(wpos + q)] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    }
    pend = rem ? row[0] & ((1u << sh) - 1) : 0u;
    npend = rem;
    nbytes += (uint64_t)(4 * top + kb);
    wpos += nfull;
  }

  __device__ __forceinline__ void pad() {
    const uint64_t bits = nbytes * 8;
    wave_lds_fence();
    int p = wpos;
    if (lane == 0) chunk[hw_pidx(p)] = ((pend << 8) | 0x80u) << (8 * (3 - npend));
    p++;
    while ((p & 15) != 14) { if (lane == 0) chunk[hw_pidx(p)] = 0; p++; }
    if (lane == 0) { chunk[hw_pidx(p)] = (uint32_t)(bits >> 32); chunk[hw_pidx(p + 1)] = (uint32_t)bits; }
    wpos = p + 2;
  }
};

// The transcript hash of proof b on ONE wavefront (the body of k_range_hash_wave; k_enc_basen_r2l5 runs it in a workgroup of its own launch
// for a one-proof verify: kernels_basen_r2l.hpp).  The caller's launch provides hw_lds_words(kw) words of dynamic LDS.
template <int NB = HW_BLOCKS>
__device__ __forceinline__ void range_hash_wave_body(const RangeHashArgs& a, int lane, uint64_t b) {
  extern __shared__ __align__(16) uint32_t hash_lds[];
  const int kw = (int)a.kw;
  WaveShaT<NB> s;
  s.init(hash_lds, kw, lane);
  const int nvalues = 1 + 2 * (int)a.ef;                      // n, c1[0..ef), c2[0..ef)
  const uint64_t pstride = (uint64_t)a.ef * 2 * kw;
  bool padded = false;
#pragma unroll 1
  for (int v = 0;; v++) {                                     // ONE call site of put_bigint / flush for the whole stream
    if (v < nvalues) {
      const int idx = v - 1, half = idx >= (int)a.ef;
      const uint32_t* src = v == 0 ? a.n + b * a.n_stride : (half ? a.c2 : a.c1) + b * pstride + (uint64_t)(idx - half * (int)a.ef) * 2 * kw;
      s.put_bigint(src, v == 0 ? kw : 2 * kw);
      if (s.wpos < 16 * NB) continue;
    } else if (!padded) {
      s.pad();
      padded = true;
    }
    if (s.wpos == 0) break;
    s.flush(s.wpos / 16 < NB ? s.wpos / 16 : NB);
  }
  if (lane != 0) return;
  // the digest as bytes in LDS (byte j at byte address j): the dynamic byte indexing below stays out of the register file
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < 8; i++) s.row[i] = __builtin_bswap32(s.st.h[i]);
  wave_lds_fence();
  const uint8_t* db = reinterpret_cast<const uint8_t*>(s.row);
  int lead = 0;
  while (lead < 31 && db[lead] == 0) lead++;      // BigInt round trip drops leading zero bytes; zero -> "00"
  uint8_t* e = a.e + b * 32;
  for (int i = 0; i < 32; i++) e[i] = (i + lead < 32) ? db[i + lead] : 0;
  const int elen = 32 - lead;
  a.e_len[b] = (uint8_t)elen;
  a.verdict[b] = ((uint32_t)elen * 8 >= a.ef) ? a.ok_value : (uint8_t)ZKP_VERDICT_MALFORMED;
  if (!a.range) return;                            // challenge only (zkp_range_challenge_batch)
  const uint32_t* rg = a.range + b * a.kw;
  uint32_t* t1 = a.third + b * a.kw;
  uint32_t* t2 = a.two_thirds + b * a.kw;
  uint64_t rem = 0;
  for (int w = (int)a.kw - 1; w >= 0; w--) {
    const uint64_t cur = (rem << 32) | rg[w];
    t1[w] = (uint32_t)(cur / 3);
    rem = cur % 3;
  }
  uint32_t carry = 0;
  for (uint32_t w = 0; w < a.kw; w++) { const uint32_t v = t1[w]; t2[w] = (v << 1) | carry; carry = v >> 31; }
}

#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(64) k_range_hash_wave(RangeHashArgs a) { range_hash_wave_body(a, (int)threadIdx.x, blockIdx.x); }
#endif

// ------------------------------------------------------------------------------------------
// Verify planning: one thread per (proof, row).  Applies the kind/bit match and the range
// predicates of verifier_output (range_proof.rs:270-348) and appends the Enc checks the row
// needs to the work list: Open -> (w1,r1)->c1[i] and (w2,r2)->c2[i]; Mask -> (masked_x,masked_r).
struct VerifyPlanArgs {
  const uint8_t* e; const uint8_t* resp_kind;
  const uint32_t* resp_w1; const uint32_t* resp_w2;
  const uint32_t* third; const uint32_t* two_thirds;
  uint32_t kw, ef; uint64_t batch;
  uint8_t* verdict;
  uint32_t* item_proof; uint32_t* item_row; unsigned long long* counter;
};

// Two further shapes serve calls of a few proofs, where the transcript hash (one lane per proof, ~10 ms) would otherwise sit
// in front of an almost empty GPU: e == nullptr builds the work list from the response kinds alone (an Open row always costs
// two Enc checks, a Mask row one; rows whose kind contradicts the challenge bit reject the proof whatever their Encs say), so
// that k_enc can start while the hash runs on a second stream; item_proof == nullptr applies the predicates only.
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_verify_plan(VerifyPlanArgs a) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool in = t < a.batch * a.ef;
  uint32_t nitems = 0;
  uint64_t b = 0; uint32_t i = 0;
  if (in && !a.e) {
    b = t / a.ef; i = (uint32_t)(t % a.ef);
    const int kind = a.resp_kind[t];
    nitems = kind == ZKP_RESP_OPEN ? 2 : kind == ZKP_RESP_MASK ? 1 : 0;
  } else if (in) {
    b = t / a.ef; i = (uint32_t)(t % a.ef);
    if (a.verdict[b] != ZKP_VERDICT_MALFORMED) {
      const int ei = challenge_bit(a.e + b * 32, i);
      const int kind = a.resp_kind[t];
      const uint32_t* T1 = a.third + b * a.kw;
      const uint32_t* T2 = a.two_thirds + b * a.kw;
      const uint32_t* w1 = a.resp_w1 + t * a.kw;
      if (!ei && kind == ZKP_RESP_OPEN) {
        const uint32_t* w2 = a.resp_w2 + t * a.kw;
        const int w1_T1 = cmp_words(w1, T1, a.kw), w2_T1 = cmp_words(w2, T1, a.kw);
        const bool flag = (w2_T1 < 0 && w1_T1 > 0 && cmp_words(w1, T2, a.kw) < 0) ||
                          (w1_T1 < 0 && w2_T1 > 0 && cmp_words(w2, T2, a.kw) < 0);      // range_proof.rs:300-305
        if (!flag) a.verdict[b] = ZKP_VERDICT_REJECT;
        nitems = 2;
      } else if (ei && kind == ZKP_RESP_MASK) {
        if (cmp_words(w1, T1, a.kw) < 0 || cmp_words(w1, T2, a.kw) > 0) a.verdict[b] = ZKP_VERDICT_REJECT;   // :338
        nitems = 1;
      } else {
        a.verdict[b] = ZKP_VERDICT_REJECT;                                              // :345
      }
    }
  }
  if (!a.item_proof) return;
  // wave-aggregated append
  const unsigned long long m2 = __ballot(nitems == 2), m1 = __ballot(nitems == 1);
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1;
  const uint32_t before = 2 * __popcll(m2 & lt) + __popcll(m1 & lt);
  const uint32_t total = 2 * __popcll(m2) + __popcll(m1);
  unsigned long long base = 0;
  if (lane == 0 && total) base = atomicAdd(a.counter, (unsigned long long)total);
  base = __shfl(base, 0);
  for (uint32_t k = 0; k < nitems; k++) {
    a.item_proof[base + before + k] = (uint32_t)b;
    a.item_row[base + before + k] = (i << 1) | k;
  }
}
#endif

// Joins the two strands of a small verify call: verdict[] holds what the hash + predicate strand decided, enc_verdict[] what
// the Enc checks of the kind-derived work list found.  The result is what the one-stream sequence writes: a failed Enc check
// rejects; an even key (k_enc says MALFORMED) marks the proof only if it has a row whose Encs the one-stream plan would have
// scheduled (kind matching its challenge bit).
struct VerdictMergeArgs {
  uint8_t* verdict; const uint8_t* enc_verdict; const uint8_t* e; const uint8_t* resp_kind; uint32_t ef; uint64_t batch;
};
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_verdict_merge(VerdictMergeArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const uint8_t v = a.verdict[b], ev = a.enc_verdict[b];
  if (v == ZKP_VERDICT_MALFORMED || ev == ZKP_VERDICT_ACCEPT) return;
  if (ev == ZKP_VERDICT_REJECT) { a.verdict[b] = ZKP_VERDICT_REJECT; return; }
  bool any = false;
  for (uint32_t i = 0; i < a.ef; i++) {
    const int ei = challenge_bit(a.e + b * 32, i), kind = a.resp_kind[b * a.ef + i];
    any = any || (!ei && kind == ZKP_RESP_OPEN) || (ei && kind == ZKP_RESP_MASK);
  }
  if (any) a.verdict[b] = ZKP_VERDICT_MALFORMED;
}
#endif

// ------------------------------------------------------------------------------------------
// Prove: generate_proof (range_proof.rs:210-252), one group (n context) per (proof, row).
struct ResponseArgs {
  const uint32_t* consts; uint64_t const_stride;     // modulus n
  const uint32_t* x; const uint32_t* r;              // [B][kw]
  const uint32_t* w1; const uint32_t* w2; const uint32_t* r1; const uint32_t* r2;   // [B][EF][kw]
  const uint8_t* e; const uint32_t* third; const uint32_t* two_thirds;
  uint8_t* resp_kind; uint8_t* resp_j; uint32_t* resp_w1; uint32_t* resp_r1; uint32_t* resp_w2; uint32_t* resp_r2;
  uint8_t* status;
  uint32_t kw, ef; uint64_t batch;
};

template <int G>
__global__ void __launch_bounds__(LdsLayoutFull<G>::THREADS) k_range_responses(ResponseArgs a) {
  using CL = ConstLayout<G>;
  using LL = LdsLayoutFull<G>;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Grp<G, LL> g;
  grp_init<G>(g, lds_raw);
  const uint64_t rows = a.batch * a.ef;
  const uint64_t gid = (uint64_t)blockIdx.x * LL::GROUPS_PER_BLOCK + (threadIdx.x / G);
  const bool live = gid < rows;
  const uint64_t t = live ? gid : rows - 1;
  const uint64_t b = t / a.ef;
  const uint32_t i = (uint32_t)(t % a.ef);
  const int kw = (int)a.kw;
  const uint32_t* cst = a.consts + b * a.const_stride;
  load_modulus_consts<G>(g, cst);
  const bool malformed = a.status[b] == ZKP_VERDICT_MALFORMED;
  const int ei = malformed ? 0 : challenge_bit(a.e + b * 32, i);
  // lane 0: masked_x = x + w1 (or x + w2), decision T < x+w1 < 2T (both strict, range_proof.rs:233-234)
  uint32_t* sum = g.expw();   // (scr() is clobbered by canonical_words below; the exponent area is unused here)
  uint32_t sel = 1, ovf = 0;
  if (g.gl == 0) {
    const uint32_t* xs = a.x + b * kw;
    const uint32_t* ws = a.w1 + t * kw;
    uint64_t c = 0;
    for (int w = 0; w < kw; w++) { c += (uint64_t)xs[w] + ws[w]; sum[w] = (uint32_t)c; c >>= 32; }
    const bool in = c == 0 && cmp_words(sum, a.third + b * kw, kw) > 0 && cmp_words(sum, a.two_thirds + b * kw, kw) < 0;
    if (!in) {
      sel = 2;
      ws = a.w2 + t * kw;
      c = 0;
      for (int w = 0; w < kw; w++) { c += (uint64_t)xs[w] + ws[w]; sum[w] = (uint32_t)c; c >>= 32; }
      ovf = c != 0;     // x + w2 does not fit the fixed-width ABI
    }
  }
  wave_lds_fence();
  sel = bcast0<G>(sel);
  ovf = bcast0<G>(ovf);
  // masked_r = secret_r * r_j % n  (range_proof.rs:239,245)
  uint32_t X[W], Y[W], R[W];
  load_value<G>(g, X, a.r + b * kw, kw);
  load_limbs_global<G>(Y, cst + CL::OFF_R2, g.gl);
  stageB<G>(g, Y);
  mm<G>(g, R, X);                                                   // r * R
  load_value<G>(g, Y, (sel == 1 ? a.r1 : a.r2) + t * kw, kw);
  stageB<G>(g, Y);
  mm<G>(g, X, R);                                                   // r * r_j  (< 2M)
  load_limbs_global<G>(Y, cst + CL::OFF_R2, g.gl);
  stageB<G>(g, Y);
  mm<G>(g, R, X);
  stage_one<G>(g);
  mm<G>(g, X, R);
  canonical_words<G>(g, X, cst + CL::OFF_N);                        // words() = masked_r
  if (!live || malformed) return;
  uint32_t* ow1 = a.resp_w1 + t * kw; uint32_t* or1 = a.resp_r1 + t * kw;
  uint32_t* ow2 = a.resp_w2 + t * kw; uint32_t* or2 = a.resp_r2 + t * kw;
  if (!ei) {                                                        // Open (range_proof.rs:226-232)
    for (int w = g.gl; w < kw; w += G) {
      ow1[w] = a.w1[t * kw + w]; or1[w] = a.r1[t * kw + w]; ow2[w] = a.w2[t * kw + w]; or2[w] = a.r2[t * kw + w];
    }
    if (g.gl == 0) { a.resp_kind[t] = ZKP_RESP_OPEN; a.resp_j[t] = 0; }
  } else {                                                          // Mask (:236-246)
    for (int w = g.gl; w < kw; w += G) { ow1[w] = sum[w]; or1[w] = g.words()[w]; ow2[w] = 0; or2[w] = 0; }
    if (g.gl == 0) {
      a.resp_kind[t] = ZKP_RESP_MASK; a.resp_j[t] = (uint8_t)sel;
      if (ovf || cst[CL::OFF_ST] != 0) a.status[b] = ZKP_VERDICT_MALFORMED;
    }
  }
}

// ------------------------------------------------------------------------------------------
// NiCorrectKeyProof::verify (correct_key_ni.rs:73-100)
// (1) one thread per proof: salt_bn, the 11 seeds and the MGF output words (mask_generation,
//     :105-117: sum_j SHA256(seed || j) << 256 j, j < key_length/256 + 1), and the small-prime
//     test that is boolean-equivalent to gcd(P, n) == 1 with P = product of primes < 6370.
struct CkHashArgs {
  const uint32_t* n; uint32_t kw; uint64_t batch;
  const uint8_t* salt; uint32_t salt_len;
  uint32_t* mgf;        // [B][11][kw+8] little-endian words of mask_generation(...)
  uint8_t* verdict;     // ACCEPT / REJECT (gcd test) to start with
  const uint32_t* primes; uint32_t nprimes;   // primes < 6370
};

__device__ __forceinline__ void sha_put_u32_as_bigint(Sha256& s, uint32_t v) { s.put_bigint(&v, 1); }

#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_ck_hash(CkHashArgs a) {
  __shared__ uint32_t shabuf[16 * 256];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const uint32_t* n = a.n + b * a.kw;
  const int kw = (int)a.kw;
  Sha256 s;
  // salt_bn = H(to_bytes(from_bytes(salt)))  (:75): leading zero bytes of the salt vanish, empty/zero -> 00
  uint32_t salt_d[8];
  {
    s.init(shabuf + threadIdx.x, 256);
    uint32_t lead = 0;
    while (lead < a.salt_len && a.salt[lead] == 0) lead++;
    if (lead == a.salt_len) s.put_bytes(0, 1);
    for (uint32_t i = lead; i < a.salt_len; i++) s.put_bytes(a.salt[i], 1);
    s.finish(salt_d);
  }
  // key_length = n.bit_length()
  int top = kw - 1;
  while (top > 0 && n[top] == 0) top--;
  const int key_length = n[top] ? top * 32 + 32 - __clz(n[top]) : 0;
  const int msklen = key_length / 256 + 1;
  uint32_t le[8];
  for (uint32_t i = 0; i < ZKP_CORRECT_KEY_M2; i++) {
    // seed = H(n || salt_bn || i)  (:79-83)
    uint32_t seed_d[8];
    s.init(shabuf + threadIdx.x, 256);
    s.put_bigint(n, kw);
#pragma unroll
    for (int k = 0; k < 8; k++) le[k] = salt_d[7 - k];
    s.put_bigint(le, 8);
    sha_put_u32_as_bigint(s, i);
    s.finish(seed_d);
    uint32_t seed_le[8];
#pragma unroll
    for (int k = 0; k < 8; k++) seed_le[k] = seed_d[7 - k];
    uint32_t* out = a.mgf + (b * ZKP_CORRECT_KEY_M2 + i) * (uint64_t)(kw + 8);
    for (int w = 0; w < kw + 8; w++) out[w] = 0;
    for (int j = 0; j < msklen; j++) {
      uint32_t hj[8];
      s.init(shabuf + threadIdx.x, 256);
      s.put_bigint(seed_le, 8);
      sha_put_u32_as_bigint(s, (uint32_t)j);
      s.finish(hj);
      if (8 * j + 8 <= kw + 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) out[8 * j + k] = hj[7 - k];   // the digests occupy disjoint 256-bit slots
      }
    }
  }
  // gcd(P, n) == 1  <=>  no prime < 6370 divides n
  bool coprime = true;
  for (uint32_t pi = 0; pi < a.nprimes; pi++) {
    const uint32_t p = a.primes[pi];
    uint64_t rem = 0;
    for (int w = kw - 1; w >= 0; w--) rem = ((rem << 32) | n[w]) % p;
    if (rem == 0) coprime = false;
  }
  a.verdict[b] = coprime ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
}
#endif

// The same per key on ONE WAVEFRONT, for calls of a few keys (one thread per key spends ~12 ms, mostly in the 830 trial
// divisions): lane i < 11 derives rho_i (its seed hash and mask blocks), and all 64 lanes share the primes.
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(64) k_ck_hash_wave(CkHashArgs a) {
  __shared__ uint32_t shabuf[16 * 64];
  const uint64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t* n = a.n + b * a.kw;
  const int kw = (int)a.kw;
  if (lane < ZKP_CORRECT_KEY_M2) {
    const uint32_t i = (uint32_t)lane;
    Sha256 s;
    uint32_t salt_d[8];
    {
      s.init(shabuf + lane, 64);
      uint32_t lead = 0;
      while (lead < a.salt_len && a.salt[lead] == 0) lead++;
      if (lead == a.salt_len) s.put_bytes(0, 1);
      for (uint32_t k = lead; k < a.salt_len; k++) s.put_bytes(a.salt[k], 1);
      s.finish(salt_d);
    }
    int top = kw - 1;
    while (top > 0 && n[top] == 0) top--;
    const int key_length = n[top] ? top * 32 + 32 - __clz(n[top]) : 0;
    const int msklen = key_length / 256 + 1;
    uint32_t le[8], seed_d[8];
    s.init(shabuf + lane, 64);
    s.put_bigint(n, kw);
#pragma unroll
    for (int k = 0; k < 8; k++) le[k] = salt_d[7 - k];
    s.put_bigint(le, 8);
    sha_put_u32_as_bigint(s, i);
    s.finish(seed_d);
    uint32_t seed_le[8];
#pragma unroll
    for (int k = 0; k < 8; k++) seed_le[k] = seed_d[7 - k];
    uint32_t* out = a.mgf + (b * ZKP_CORRECT_KEY_M2 + i) * (uint64_t)(kw + 8);
    for (int w = 0; w < kw + 8; w++) out[w] = 0;
    for (int j = 0; j < msklen; j++) {
      uint32_t hj[8];
      s.init(shabuf + lane, 64);
      s.put_bigint(seed_le, 8);
      sha_put_u32_as_bigint(s, (uint32_t)j);
      s.finish(hj);
      if (8 * j + 8 <= kw + 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) out[8 * j + k] = hj[7 - k];
      }
    }
  }
  bool coprime = true;
  for (uint32_t pi = (uint32_t)lane; pi < a.nprimes; pi += 64) {
    const uint32_t p = a.primes[pi];
    uint64_t rem = 0;
    for (int w = kw - 1; w >= 0; w--) rem = ((rem << 32) | n[w]) % p;
    if (rem == 0) coprime = false;
  }
  const bool all = __all(coprime);
  if (lane == 0) a.verdict[b] = all ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
}
#endif

// (2) one group (n context) per (proof, i): sigma_i^n mod n  ==  mask_generation(..) mod n  (:84,:92,:95)
struct CkCheckArgs {
  const uint32_t* n; const uint32_t* sigma; const uint32_t* mgf; const uint32_t* consts;
  uint32_t* table; uint8_t* verdict; uint64_t count; int n_bits;
  unsigned long long* work_counter;
};

// PAIR: two neighbouring groups per item and the right-to-left ladder (kernels_modexp.hpp: powm_pair; latency engine, a few keys)
template <int G, bool PAIR = false>
__global__ void __launch_bounds__(256, ZKP_WPE) k_ck_check(CkCheckArgs a) {
  using CL = ConstLayout<G>;
  using LL = LdsLayout<G>;
  constexpr int L = Geo<G>::L;
  extern __shared__ __align__(16) uint32_t lds_raw[];
  Grp<G> g;
  grp_init<G>(g, lds_raw);
  const uint64_t ggrp = (uint64_t)blockIdx.x * LL::GROUPS_PER_BLOCK + (threadIdx.x / G);
  uint32_t* tab = a.table + ggrp * (uint64_t)((1 << WIN_KEY) * L);
  const int kw = a.n_bits / 32;
  const int lane0 = threadIdx.x & 63;
  for (;;) {
    unsigned long long base = 0;
    constexpr int GP = PAIR ? 2 * G : G;
    const int role = PAIR ? (lane0 / G) & 1 : 0;
    if (lane0 == 0) base = atomicAdd(a.work_counter, (unsigned long long)(64 / GP));
    base = __shfl(base, 0);
    if (base >= a.count) break;
    const uint64_t idx = base + (uint64_t)(lane0 / GP);
    const bool live = idx < a.count && role == 0;
    const uint64_t item = idx < a.count ? idx : a.count - 1;
    const uint64_t b = item / ZKP_CORRECT_KEY_M2;
    const uint32_t* cst = a.consts + b * CL::WORDS;
    load_modulus_consts<G>(g, cst);
    uint32_t X[W], R[W], T[W], RHO[W], SIG[W];
    const int lane = threadIdx.x & 63;
    const unsigned long long gm = ((1ull << G) - 1) << (lane & ~(G - 1));
    // sigma^n mod n FIRST: nothing but the ladder's own operands is live across its product loops (rho, which the comparison
    // needs, is derived afterwards; the modulus is read again) — the register file of the W = 36 loops has no room for bystanders
    load_limbs_global<G>(T, cst + CL::OFF_R2, g.gl);
    stageB<G>(g, T);
    load_value<G>(g, T, a.sigma + item * kw, kw);
    mm<G>(g, X, T);
    if constexpr (PAIR) powm_pair<G>(g, X, a.n_bits, cst, a.n + b * kw, role);
    else powm<G, false, WIN_KEY>(g, X, a.n_bits, tab, cst, nullptr, a.n + b * kw);
    load_modulus_consts<G>(g, cst);
    stage_one<G>(g);
    mm<G>(g, R, X);
    normalize_exact<G>(R, g.gl);
#pragma unroll
    for (int k = 0; k < W; k++) SIG[k] = R[k];           // sigma^n as exact limbs of a value <= n
    // rho = (v_lo + v_hi * 2^(32 kw)) mod n with v = MGF output (kw + 8 words)
    const uint32_t* v = a.mgf + item * (uint64_t)(kw + 8);
    load_limbs_global<G>(T, cst + CL::OFF_R2, g.gl);
    stageB<G>(g, T);                                     // B() = R2
    load_value<G>(g, X, v, kw);
    mm<G>(g, RHO, X);                                    // v_lo * R
    {
      uint32_t P[W];                                     // P = 2^(32 kw) as limbs
      const int bit = 32 * kw;
#pragma unroll
      for (int k = 0; k < W; k++) P[k] = ((g.gl * W + k) == bit / LB) ? (1u << (bit % LB)) : 0u;
      mm<G>(g, R, P);                                    // 2^(32kw) * R
      mm<G>(g, T, R);                                    // 2^(32kw) * R^2   (B() still R2)
      stageB<G>(g, T);
      load_value<G>(g, X, v + kw, 8);
      mm<G>(g, R, X);                                    // v_hi * 2^(32kw) * R
#pragma unroll
      for (int k = 0; k < W; k++) RHO[k] += R[k];        // limbs < 2^30 + 32: still a valid operand
    }
    stage_one<G>(g);
    mm<G>(g, R, RHO);                                    // rho mod n  (<= n)
    normalize_exact<G>(R, g.gl);
    bool is_m = true;
#pragma unroll
    for (int k = 0; k < W; k++) is_m = is_m && (R[k] == g.N[k]);
    if ((__ballot(is_m) & gm) == gm) {
#pragma unroll
      for (int k = 0; k < W; k++) R[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < W; k++) RHO[k] = R[k];           // canonical rho limbs
    bool same = true, eqm = true;
#pragma unroll
    for (int k = 0; k < W; k++) { same = same && (SIG[k] == RHO[k]); eqm = eqm && (SIG[k] == g.N[k]); }
    // R == n means the residue 0
    const bool r_is_m = (__ballot(eqm) & gm) == gm;
    bool rho_zero = true;
#pragma unroll
    for (int k = 0; k < W; k++) rho_zero = rho_zero && (RHO[k] == 0);
    const bool all_same = (__ballot(same) & gm) == gm;
    const bool all_rho_zero = (__ballot(rho_zero) & gm) == gm;
    const bool ok = r_is_m ? all_rho_zero : all_same;
    if (live && g.gl == 0 && (!ok || cst[CL::OFF_ST] != 0)) a.verdict[b] = ZKP_VERDICT_REJECT;
  }
}

// ------------------------------------------------------------------------------------------
// CompositeDLogProof (wi_dlog_proof.rs:46-91), one thread per proof for the byte/word-level parts.

struct DlogHashArgs {
  const uint32_t* N; const uint32_t* g; const uint32_t* ni; const uint32_t* x;   // [B][kw]
  uint32_t kw; uint64_t batch;
  uint32_t* e;            // [B][8] little-endian words of e = H(x || g || N || ni)
  // verify only
  uint8_t* verdict;       // ACCEPT / MALFORMED to start with (nullable in prove)
  // prove only: y = r + e * secret  (wi_dlog_proof.rs:62)
  const uint32_t* secret; const uint32_t* r; uint32_t* y; uint32_t yw;
  uint32_t parts;         // 1 = the challenge (and the prover's y), 2 = the verifier's pre-checks, 3 = both.  A small verify call
                          // runs the pre-checks (two GCDs, ~1 ms on one lane) on the second stream next to the exponentiations
};

constexpr int DLOG_THREADS = 64;
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(DLOG_THREADS) k_dlog_hash(DlogHashArgs a) {
  extern __shared__ __align__(16) uint32_t dlog_lds[];   // [16][64] SHA block buffers | [2*kw][64] gcd operands
  uint32_t* shabuf = dlog_lds;
  const uint64_t b = (uint64_t)blockIdx.x * DLOG_THREADS + threadIdx.x;
  if (b >= a.batch) return;
  const int kw = (int)a.kw;
  const uint32_t *N = a.N + b * kw, *g = a.g + b * kw, *ni = a.ni + b * kw, *x = a.x + b * kw;
  uint32_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (a.parts & 1) {
    Sha256 s;
    s.init(shabuf + threadIdx.x, DLOG_THREADS);
    s.put_bigint(x, kw); s.put_bigint(g, kw); s.put_bigint(N, kw); s.put_bigint(ni, kw);   // :56-61 / :75-80
    uint32_t d[8];
    s.finish(d);
#pragma unroll
    for (int k = 0; k < 8; k++) { e[k] = d[7 - k]; a.e[b * 8 + k] = e[k]; }
  }
  if (a.verdict && (a.parts & 2)) {
    // assert!(N > 2^128) :69 ; gcd(g, N) == 1 :72 ; gcd(ni, N) == 1 :73  (panics in the reference)
    bool big = false;
    for (int w = 5; w < kw; w++) big = big || N[w] != 0;
    big = big || N[4] > 1 || (N[4] == 1 && (N[0] | N[1] | N[2] | N[3]) != 0);
    bool ok = big && (N[0] & 1);    // even N: Montgomery path undefined -> reported as malformed (documented deviation)
    if (ok) {
      uint32_t* u = dlog_lds + 16 * DLOG_THREADS + threadIdx.x;
      uint32_t* v = u + kw * DLOG_THREADS;
      ok = wb_coprime_to_odd(g, N, kw, u, v, DLOG_THREADS) && wb_coprime_to_odd(ni, N, kw, u, v, DLOG_THREADS);
    }
    a.verdict[b] = ok ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_MALFORMED;
  }
  if (a.y && (a.parts & 1)) {
    const uint32_t* sc = a.secret + b * 8;
    const uint32_t* r = a.r + b * 16;
    uint32_t* y = a.y + b * a.yw;
    uint64_t acc[17];
    for (int k = 0; k < 17; k++) acc[k] = k < 16 ? r[k] : 0;
    // e * secret: 8 x 8 words
    uint32_t prod[16];
    for (int k = 0; k < 16; k++) prod[k] = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < 8; j++) {
        const uint64_t t = (uint64_t)e[i] * sc[j] + prod[i + j] + carry;
        prod[i + j] = (uint32_t)t; carry = t >> 32;
      }
      prod[i + 8] = (uint32_t)carry;
    }
    uint64_t c = 0;
    for (uint32_t k = 0; k < a.yw; k++) {
      c += (k < 17 ? acc[k] : 0) + (k < 16 ? prod[k] : 0);
      y[k] = (uint32_t)c; c >>= 32;
    }
  }
}
#endif

// final comparison x ==? g^y * ni^e mod N (:83-90): plain word compare, only ACCEPT can be downgraded
struct DlogCmpArgs { const uint32_t* x; const uint32_t* t; const uint32_t* consts; uint64_t const_stride; int st_off; uint32_t kw; uint64_t batch; uint8_t* verdict; };
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_dlog_compare(DlogCmpArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  if (a.verdict[b] != ZKP_VERDICT_ACCEPT) return;
  bool same = true;
  for (uint32_t w = 0; w < a.kw; w++) same = same && a.x[b * a.kw + w] == a.t[b * a.kw + w];
  if (a.consts[b * a.const_stride + a.st_off] != 0) { a.verdict[b] = ZKP_VERDICT_MALFORMED; return; }
  if (!same) a.verdict[b] = ZKP_VERDICT_REJECT;
}
#endif

// ------------------------------------------------------------------------------------------
// ZeroProof / CiphertextProof (zero_enc_proof.rs:44-94, correct_ciphertext.rs:42-97):
// e = H(n || c || a) per proof (a = the commitment: ZeroProof.a / CiphertextProof.c_prime), optionally
// z1 = x' + x*e over the integers (correct_ciphertext.rs:59).
struct SigmaHashArgs {
  const uint32_t* n; uint64_t n_stride; const uint32_t* c; const uint32_t* a;
  uint32_t kw; uint64_t batch;
  uint32_t* e;                                   // [B][8]
  const uint32_t* x; const uint32_t* x_prime; uint32_t* z1; uint32_t z1w;   // nullable
};

#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_sigma_hash(SigmaHashArgs a) {
  __shared__ uint32_t shabuf[16 * 256];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const int kw = (int)a.kw;
  Sha256 s;
  s.init(shabuf + threadIdx.x, 256);
  s.put_bigint(a.n + b * a.n_stride, kw);
  s.put_bigint(a.c + b * 2 * kw, 2 * kw);
  s.put_bigint(a.a + b * 2 * kw, 2 * kw);
  uint32_t d[8], e[8];
  s.finish(d);
#pragma unroll
  for (int k = 0; k < 8; k++) { e[k] = d[7 - k]; a.e[b * 8 + k] = e[k]; }
  if (a.z1) {
    // z1 = x' + x * e : (kw x 8)-word product, row by row straight into the output words
    const uint32_t* x = a.x + b * kw;
    const uint32_t* xp = a.x_prime + b * kw;
    uint32_t* z = a.z1 + b * a.z1w;
    for (uint32_t w = 0; w < a.z1w; w++) z[w] = w < (uint32_t)kw ? xp[w] : 0u;
    for (int i = 0; i < 8; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < kw; j++) {
        const uint64_t t = (uint64_t)e[i] * x[j] + z[i + j] + carry;
        z[i + j] = (uint32_t)t; carry = t >> 32;
      }
      for (uint32_t w = i + kw; carry && w < a.z1w; w++) { const uint64_t t = (uint64_t)z[w] + carry; z[w] = (uint32_t)t; carry = t >> 32; }
    }
  }
}
#endif

// VerlinProof challenge e = H(n || c || c' || phi_x || phi_a) (verlin_proof.rs:78-84, 102-108) and, for prove,
// the three integer responses z = x e + a (:85-87).
struct VerlinHashArgs {
  const uint32_t* n; uint64_t n_stride; const uint32_t* v[4];   // c, c', phi_x, phi_a : [B][2kw]
  uint32_t kw; uint64_t batch; uint32_t* e;
  const uint32_t* x[3]; const uint32_t* a[3]; uint32_t* z[3]; uint32_t zw;   // nullable
};
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_verlin_hash(VerlinHashArgs a) {
  __shared__ uint32_t shabuf[16 * 256];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const int kw = (int)a.kw;
  Sha256 s;
  s.init(shabuf + threadIdx.x, 256);
  s.put_bigint(a.n + b * a.n_stride, kw);
  for (int k = 0; k < 4; k++) s.put_bigint(a.v[k] + b * 2 * kw, 2 * kw);
  uint32_t d[8], e[8];
  s.finish(d);
#pragma unroll
  for (int k = 0; k < 8; k++) { e[k] = d[7 - k]; a.e[b * 8 + k] = e[k]; }
  for (int q = 0; q < 3; q++) {
    if (!a.z[q]) continue;
    const uint32_t* x = a.x[q] + b * kw;
    const uint32_t* ad = a.a[q] + b * kw;
    uint32_t* z = a.z[q] + b * a.zw;
    for (uint32_t w = 0; w < a.zw; w++) z[w] = w < (uint32_t)kw ? ad[w] : 0u;
    for (int i = 0; i < 8; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < kw; j++) {
        const uint64_t t = (uint64_t)e[i] * x[j] + z[i + j] + carry;
        z[i + j] = (uint32_t)t; carry = t >> 32;
      }
      for (uint32_t w = i + kw; carry && w < a.zw; w++) { const uint64_t t = (uint64_t)z[w] + carry; z[w] = (uint32_t)t; carry = t >> 32; }
    }
  }
}
#endif

// verdict[b] = (lhs[b] == rhs[b]) word for word (`c_z == c_z_test`, zero_enc_proof.rs:90, correct_ciphertext.rs:93)
struct WordsCmpArgs { const uint32_t* lhs; const uint32_t* rhs; const uint32_t* consts; uint64_t const_stride; int st_off; uint32_t words; uint64_t batch; uint8_t* verdict; };
#ifndef ZKP_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(256) k_words_compare(WordsCmpArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  bool same = true;
  for (uint32_t w = 0; w < a.words; w++) same = same && a.lhs[b * a.words + w] == a.rhs[b * a.words + w];
  uint8_t v = same ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
  if (a.consts[b * a.const_stride + a.st_off] != 0) v = ZKP_VERDICT_MALFORMED;   // even key: Montgomery path undefined
  a.verdict[b] = v;
}
#endif

}  // namespace zkp
