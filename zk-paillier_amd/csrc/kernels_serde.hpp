// kernels_serde.hpp — the reference's wire format for big integers is the DECIMAL string (src/serialize.rs:1-78:
// `bigint` and `vecbigint` serde adapters over BigInt::to_str_radix(10) / from_str_radix(s, 10), i.e. GMP
// mpz_get_str / mpz_set_str).  A batch of 4096 RangeProofNi proofs is ~1.5 M such strings (2.3 GB of JSON text); the
// radix conversion in both directions is per-number independent work and runs here, one number per lane, straight
// from the uploaded text into the SoA limb buffers the proof kernels read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/zkp_hip.h"

namespace zkp {

constexpr int SERDE_LANES = 64;

// ---- decimal text -> little-endian 32-bit words
// Horner over 9-digit groups: x = x * 10^k + group.  The accumulator of each lane lives in thread-interleaved LDS
// (word w of lane t at acc[w * 64 + t]); finished numbers are written out cooperatively (64 lanes x consecutive words).
// Accepted exactly as mpz_set_str(s, 10) accepts: an optional leading '-', white space anywhere (ignored), digits.
__global__ void __launch_bounds__(SERDE_LANES) k_dec2bin(const char* __restrict__ text, const zkp_dec_item* __restrict__ items, uint64_t count,
                                                         uint32_t* __restrict__ dst, uint8_t* __restrict__ status, int max_words) {
  extern __shared__ __align__(16) uint32_t acc[];
  const int lane = threadIdx.x;
  const uint64_t item0 = (uint64_t)blockIdx.x * SERDE_LANES;
  const uint64_t idx = item0 + lane;
  const bool live = idx < count;
  uint32_t* x = acc + lane;
  int words = 0;
  uint8_t st = ZKP_DEC_OK;
  if (live) {
    const zkp_dec_item it = items[idx];
    words = (int)it.words;
    const char* s = text + it.text_off;
    const uint32_t len = it.len;
    int n_live = 0;
    uint32_t group = 0, mult = 1;
    bool neg = false, seen = false;
    for (uint32_t p = 0; p <= len && st == ZKP_DEC_OK; p++) {
      bool flush = p == len;
      if (!flush) {
        const char ch = s[p];
        if (ch >= '0' && ch <= '9') {
          group = group * 10u + (uint32_t)(ch - '0'); mult *= 10u; seen = true;
          flush = mult == 1000000000u;
        } else if (ch == ' ' || (ch >= '\t' && ch <= '\r')) {
          continue;
        } else if (ch == '-' && !seen && !neg) {
          neg = true; continue;
        } else {
          st = ZKP_DEC_INVALID; break;
        }
      }
      if (flush && mult != 1) {
        uint64_t carry = group;
        for (int w = 0; w < n_live; w++) {
          const uint64_t t = (uint64_t)x[w * SERDE_LANES] * mult + carry;
          x[w * SERDE_LANES] = (uint32_t)t;
          carry = t >> 32;
        }
        if (carry) {
          if (n_live < words) x[n_live++ * SERDE_LANES] = (uint32_t)carry;
          else st = ZKP_DEC_OVERFLOW;
        }
        group = 0; mult = 1;
      }
    }
    if (st == ZKP_DEC_OK && !seen) st = ZKP_DEC_INVALID;          // "" or "-": mpz_set_str fails
    if (st == ZKP_DEC_OK && neg && n_live > 0) st = ZKP_DEC_NEGATIVE;   // "-0" is zero
    for (int w = (st == ZKP_DEC_OK ? n_live : 0); w < max_words; w++) x[w * SERDE_LANES] = 0;
    status[idx] = st;
  }
  __syncthreads();
  // cooperative, coalesced write-out: item j of the block by all lanes
  for (int j = 0; j < SERDE_LANES; j++) {
    const uint64_t id = item0 + j;
    if (id >= count) break;
    const zkp_dec_item it = items[id];
    for (uint32_t w = lane; w < it.words; w += SERDE_LANES) dst[it.dst_off + w] = acc[w * SERDE_LANES + j];
  }
}

// ---- little-endian 32-bit words -> decimal text (mpz_get_str(.., 10, x): no leading zeros, "0" for zero)
// Repeated short division by 10^9, top word down; nine digits per pass, written from the end of the item's row
// (out_text + idx * pitch + pitch - len is the string).  pitch >= ceil(words * 9.633) + 1.
__global__ void __launch_bounds__(SERDE_LANES) k_bin2dec(const uint32_t* __restrict__ src, uint64_t src_stride, int words, uint64_t count,
                                                         char* __restrict__ out_text, uint32_t pitch, uint32_t* __restrict__ out_len) {
  extern __shared__ __align__(16) uint32_t acc[];
  const int lane = threadIdx.x;
  const uint64_t item0 = (uint64_t)blockIdx.x * SERDE_LANES;
  // cooperative, coalesced read-in
  for (int j = 0; j < SERDE_LANES; j++) {
    const uint64_t id = item0 + j;
    if (id >= count) break;
    for (int w = lane; w < words; w += SERDE_LANES) acc[w * SERDE_LANES + j] = src[id * src_stride + w];
  }
  __syncthreads();
  const uint64_t idx = item0 + lane;
  if (idx >= count) return;
  uint32_t* x = acc + lane;
  int n_live = words;
  while (n_live > 0 && x[(n_live - 1) * SERDE_LANES] == 0) n_live--;
  char* row = out_text + idx * (uint64_t)pitch;
  uint32_t pos = pitch;                        // next digit goes to row[--pos]
  if (n_live == 0) row[--pos] = '0';
  while (n_live > 0) {
    uint64_t rem = 0;
    for (int w = n_live - 1; w >= 0; w--) {
      const uint64_t cur = (rem << 32) | x[w * SERDE_LANES];
      const uint64_t q = cur / 1000000000ull;
      rem = cur - q * 1000000000ull;
      x[w * SERDE_LANES] = (uint32_t)q;
    }
    while (n_live > 0 && x[(n_live - 1) * SERDE_LANES] == 0) n_live--;
    uint32_t r = (uint32_t)rem;
    for (int d = 0; d < 9; d++) {
      if (n_live == 0 && r == 0 && d > 0) break;       // most significant group: no leading zeros
      row[--pos] = (char)('0' + r % 10u);
      r /= 10u;
    }
  }
  out_len[idx] = pitch - pos;
}

}  // namespace zkp
