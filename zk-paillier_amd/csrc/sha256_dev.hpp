// sha256_dev.hpp — per-thread streaming SHA-256 for the Fiat-Shamir transcripts
// (src/zkproofs/utils.rs:9-22 compute_digest: SHA-256 over the concatenation of the minimal
// big-endian byte strings of each BigInt, no separators).  One hash per thread; the 16-word
// block buffer of each thread lives in LDS (word-major: conflict-free) so that the write
// cursor can be a run-time index without spilling a register array to scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zkp {

__device__ __constant__ const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct Sha256 {
  uint32_t h[8];
  uint32_t* buf;        // LDS: word i of this thread at buf[i * stride]
  int stride;
  int widx;             // words in the block buffer
  uint32_t pend;        // pending bytes (right aligned)
  int npend;            // 0..3
  uint64_t nbytes;

  __device__ __forceinline__ static uint32_t ror(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
  // a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96); the compiler alone emits two v_xor_b32
  __device__ __forceinline__ static uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
  // Maj and Ch as ONE v_bitop3_b32 each (truth tables 0xE8, 0xCA).  Written as (a & b) ^ (a & c) ^ (b & c) the compiler shares a & b with the
  // next round's b & c and spends v_and + v_xor + v_bitop3 per round: 16.7 instructions per round of the serial chain instead of 14.7 —
  // and a transcript hash on one wavefront IS that chain (k_range_hash_wave: 2052 blocks per proof, 4 cycles per instruction).
  __device__ __forceinline__ static uint32_t maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
  __device__ __forceinline__ static uint32_t ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }

  __device__ __forceinline__ void init(uint32_t* lds_buf, int lds_stride) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    buf = lds_buf; stride = lds_stride; widx = 0; pend = 0; npend = 0; nbytes = 0;
  }

  __device__ __forceinline__ void compress() {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = buf[i * stride];
    compress_words(w);
  }

  // one 64-byte block given as its 16 big-endian words (clobbers w)
  __device__ __forceinline__ void compress_words(uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) {
        wi = w[i];
      } else {
        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        const uint32_t s0 = xor3(ror(w15, 7), ror(w15, 18), w15 >> 3);
        const uint32_t s1 = xor3(ror(w2, 17), ror(w2, 19), w2 >> 10);
        wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        w[i & 15] = wi;
      }
      const uint32_t S1 = xor3(ror(e, 6), ror(e, 11), ror(e, 25));
      const uint32_t t1 = hh + S1 + ch(e, f, g) + SHA_K[i] + wi;
      const uint32_t S0 = xor3(ror(a, 2), ror(a, 13), ror(a, 22));
      const uint32_t t2 = S0 + maj(a, b, c);
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }

  __device__ __forceinline__ void emit_word(uint32_t x) {
    buf[widx * stride] = x;
    widx++;
    if (widx == 16) { compress(); widx = 0; }
  }

  // k (1..4) bytes, right aligned in x (higher bits of x must be zero)
  __device__ __forceinline__ void put_bytes(uint32_t x, int k) {
    nbytes += (uint64_t)k;
    const uint64_t t = ((uint64_t)pend << (8 * k)) | x;
    const int np = npend + k;
    if (np >= 4) {
      const int rem = np - 4;
      emit_word((uint32_t)(t >> (8 * rem)));
      pend = (uint32_t)(t & ((1ull << (8 * rem)) - 1));
      npend = rem;
    } else {
      pend = (uint32_t)t;
      npend = np;
    }
  }
  __device__ __forceinline__ void put_word(uint32_t x) { put_bytes(x, 4); }

  // minimal big-endian encoding of the little-endian limb array v[0..nwords): zero -> one 00 byte
  // ([upstream] curv BigInt::to_bytes over GMP: (sizeinbase(x,2)+7)/8 bytes).
  // The top word goes through the byte path (1..4 bytes).  Below it the value is whole words: once the block buffer is empty
  // they are consumed 16 at a time straight into registers — each message word is one funnel shift of two neighbouring source
  // words by the 0..3 pending bytes — without the per-word trip through the LDS block buffer; head (until the buffer is empty)
  // and tail (< 16 words) take the word path.  Transcripts are ~2 000 blocks per proof: this is where the time goes.
  __device__ __forceinline__ void put_bigint(const uint32_t* v, int nwords) {
    int top = nwords - 1;
    while (top > 0 && v[top] == 0) top--;
    const uint32_t tw = v[top];
    const int k = tw == 0 ? 1 : (4 - (__clz(tw) >> 3));
    put_bytes(tw, k);
    int i = top - 1;
    while (i >= 0 && widx != 0) { put_word(v[i]); i--; }
    if (i >= 15) {
      const int sh = 8 * npend;                       // 0, 8, 16, 24: the same for the whole value
      uint32_t prev = pend;                           // npend pending bytes, right aligned
      while (i >= 15) {
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = v[i - q];
        if (sh) {
#pragma unroll
          for (int q = 0; q < 16; q++) { const uint32_t cur = w[q]; w[q] = (prev << (32 - sh)) | (cur >> sh); prev = cur; }
        }
        compress_words(w);
        nbytes += 64;
        i -= 16;
      }
      if (sh) pend = prev & ((1u << sh) - 1);
    }
    for (; i >= 0; i--) put_word(v[i]);
  }

  __device__ __forceinline__ void finish(uint32_t (&out)[8]) {
    const uint64_t bits = nbytes * 8;
    put_bytes(0x80, 1);
    // pad with zero bytes until 8 bytes remain in the block
    while (npend != 0) put_bytes(0, 1);
    while (widx != 14) emit_word(0);
    nbytes = 0;   // (length already captured)
    emit_word((uint32_t)(bits >> 32));
    emit_word((uint32_t)bits);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = h[i];
  }
};

}  // namespace zkp
