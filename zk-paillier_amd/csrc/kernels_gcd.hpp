// kernels_gcd.hpp — word-batched binary GCD / modular inverse, one big integer pair per lane.
//
// Replaces what the reference reaches as BigInt::egcd / mod_inv (GMP mpz_gcdext / mpz_invert): the coprimality tests of
// CompositeDLogProof::verify (wi_dlog_proof.rs:72-73) and BigInt::mod_inv (multiplication_proof.rs:95,133, correct_message.rs:53,76,141).
//
// Algorithm (Pornin, "Optimized Binary GCD for Modular Inversion", 2020): the classic binary GCD does one bit of work per pass
// over the operands.  Here K = 30 binary-GCD steps are first run on 64-bit APPROXIMATIONS of (a, b) — their low 30 bits, which
// decide every parity exactly, glued to their top 34 bits, which decide the comparisons — while a 2x2 matrix (f0 g0; f1 g1) of
// 31-bit signed coefficients records what the steps did; then the matrix is applied to the full operands in ONE pass:
//     a' = (f0 a + g0 b) / 2^30,   b' = (f1 a + g1 b) / 2^30          (exact divisions; a sign is fixed up afterwards)
// and, for the inverse, to the cofactors modulo m in one more pass (u' = (f0 u + g0 v + c m) / 2^30 with the balanced
// c = -(f0 u + g0 v) / m mod 2^30: |u'| <= max(|u|, |v|) + m/2 — the cofactors stay within a small multiple of m, observed
// < 1.13 m, NOT inside (-m, m); the caller reduces the one it uses).  len(a) + len(b) shrinks by about 30 bits per round: 258 rounds for
// 4096-bit operands instead of ~5800 bit-serial ones.  b stays odd throughout; the loop ends when a == 0, gcd = b.
// tools/wbgcd_model.py is the word-for-word Python model of this file (int64 ranges asserted).
//
// Data layout: every operand lives in thread-interleaved LDS (word w of this lane at p[w * S], conflict-free); all loads of a
// chunk of CH words are issued before its first store.  kw (words per operand) is a multiple of CH.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zkp {

constexpr int GCD_K = 30;                      // binary-GCD steps per round
constexpr uint32_t GCD_MK = (1u << GCD_K) - 1;
constexpr int GCD_CH = 8;

// bits [p, p + 34) of the kw-word number x (p >= 0)
__device__ __forceinline__ uint64_t gcd_top34(const uint32_t* x, int S, int kw, int p) {
  const int w0 = p >> 5, off = p & 31;
  const uint64_t lo = x[w0 * S];
  const uint64_t mid = w0 + 1 < kw ? x[(w0 + 1) * S] : 0u;
  const uint64_t hi = w0 + 2 < kw ? x[(w0 + 2) * S] : 0u;
  uint64_t v = (lo | (mid << 32)) >> off;
  if (off) v |= hi << (64 - off);
  return v & ((1ull << 34) - 1);
}

// x = -x over nw words (two's complement)
__device__ __forceinline__ void gcd_negate(uint32_t* x, int S, int nw) {
  uint32_t carry = 1;
  for (int w0 = 0; w0 < nw; w0 += GCD_CH) {
    uint32_t t[GCD_CH];
#pragma unroll
    for (int k = 0; k < GCD_CH; k++) t[k] = x[(w0 + k) * S];
#pragma unroll
    for (int k = 0; k < GCD_CH; k++) {
      const uint64_t s = (uint64_t)(~t[k]) + carry;
      x[(w0 + k) * S] = (uint32_t)s;
      carry = (uint32_t)(s >> 32);
    }
  }
}

// The round loop.  a, b: kw words each (any values, b odd, not both zero).  COF: u, v (kw + 1 words, two's complement) are the
// cofactors modulo the odd modulus m (kw words; the caller sets u, v and b == m's relation: a == y u, b == y v mod m) and minv30 =
// -m^-1 mod 2^30.  Returns the live word count of b; afterwards a == 0 and b == gcd.
template <bool COF>
__device__ __forceinline__ int wb_gcd(uint32_t* a, uint32_t* b, uint32_t* u, uint32_t* v, const uint32_t* m, uint32_t minv30, int kw, int S) {
  int na = kw, nb = kw;
  for (;;) {
    while (na > 0 && a[(na - 1) * S] == 0) na--;
    if (na == 0) break;
    while (nb > 1 && b[(nb - 1) * S] == 0) nb--;
    const int la = 32 * na - __builtin_clz(a[(na - 1) * S]);
    const int lb = 32 * nb - __builtin_clz(b[(nb - 1) * S]);
    const int n = la > lb ? la : lb;
    uint64_t xa, xb;
    if (n <= 64) {
      xa = (uint64_t)a[0] | ((uint64_t)(kw > 1 ? a[S] : 0u) << 32);
      xb = (uint64_t)b[0] | ((uint64_t)(kw > 1 ? b[S] : 0u) << 32);
    } else {
      xa = (gcd_top34(a, S, kw, n - 34) << GCD_K) | (a[0] & GCD_MK);
      xb = (gcd_top34(b, S, kw, n - 34) << GCD_K) | (b[0] & GCD_MK);
    }
    // ---- K steps on the approximations
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
    for (int i = 0; i < GCD_K; i++) {
      const bool odd = (xa & 1) != 0;
      const bool sw = odd && xa < xb;
      const uint64_t ta = sw ? xb : xa, tb = sw ? xa : xb;
      const int32_t tf0 = sw ? f1 : f0, tg0 = sw ? g1 : g0, tf1 = sw ? f0 : f1, tg1 = sw ? g0 : g1;
      xa = (odd ? ta - tb : ta) >> 1;
      xb = tb;
      f0 = odd ? tf0 - tf1 : tf0;
      g0 = odd ? tg0 - tg1 : tg0;
      f1 = tf1 * 2;
      g1 = tg1 * 2;
    }
    // ---- (a, b) <- ((f0 a + g0 b) >> K, (f1 a + g1 b) >> K) in one pass; |f| + |g| <= 2^30 keeps every sum inside int64
    const int nw0 = na > nb ? na : nb;
    const int nw = (nw0 + GCD_CH - 1) / GCD_CH * GCD_CH;          // (words above a live count are zero: a chunk may run past it)
    {
      int64_t accA = 0, accB = 0;
      uint32_t pA = 0, pB = 0;
      for (int w0 = 0; w0 < nw; w0 += GCD_CH) {
        uint32_t A[GCD_CH], Bv[GCD_CH], Ta[GCD_CH], Tb[GCD_CH];
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) { A[k] = a[(w0 + k) * S]; Bv[k] = b[(w0 + k) * S]; }
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) {
          const int64_t aw = (int64_t)(uint64_t)A[k], bw = (int64_t)(uint64_t)Bv[k];
          accA += (int64_t)f0 * aw + (int64_t)g0 * bw;
          accB += (int64_t)f1 * aw + (int64_t)g1 * bw;
          Ta[k] = (uint32_t)accA; accA >>= 32;
          Tb[k] = (uint32_t)accB; accB >>= 32;
        }
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) {
          if (w0 + k > 0) {
            a[(w0 + k - 1) * S] = ((k ? Ta[k - 1] : pA) >> GCD_K) | (Ta[k] << (32 - GCD_K));
            b[(w0 + k - 1) * S] = ((k ? Tb[k - 1] : pB) >> GCD_K) | (Tb[k] << (32 - GCD_K));
          }
        }
        pA = Ta[GCD_CH - 1]; pB = Tb[GCD_CH - 1];
      }
      a[(nw - 1) * S] = (pA >> GCD_K) | ((uint32_t)accA << (32 - GCD_K));
      b[(nw - 1) * S] = (pB >> GCD_K) | ((uint32_t)accB << (32 - GCD_K));
      // the approximated comparisons may have taken the smaller operand for the larger one: a result can come out negative
      if (accA < 0) { gcd_negate(a, S, nw); f0 = -f0; g0 = -g0; }
      if (accB < 0) { gcd_negate(b, S, nw); f1 = -f1; g1 = -g1; }
      na = nb = nw;                                               // either result may be as long as the longer input
    }
    if constexpr (COF) {
      // ---- (u, v) <- ((f0 u + g0 v + cu m) >> K, (f1 u + g1 v + cv m) >> K): cu, cv in [-2^29, 2^29) make the sums divisible
      const uint32_t u0 = u[0], v0 = v[0];
      int32_t cu = (int32_t)((((uint32_t)f0 * u0 + (uint32_t)g0 * v0) * minv30) & GCD_MK);
      int32_t cv = (int32_t)((((uint32_t)f1 * u0 + (uint32_t)g1 * v0) * minv30) & GCD_MK);
      if (cu >> (GCD_K - 1)) cu -= 1 << GCD_K;
      if (cv >> (GCD_K - 1)) cv -= 1 << GCD_K;
      int64_t accU = 0, accV = 0;
      uint32_t pU = 0, pV = 0;
      for (int w0 = 0; w0 < kw; w0 += GCD_CH) {
        uint32_t U[GCD_CH], V[GCD_CH], M[GCD_CH], Tu[GCD_CH], Tv[GCD_CH];
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) { U[k] = u[(w0 + k) * S]; V[k] = v[(w0 + k) * S]; M[k] = m[(w0 + k) * S]; }
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) {
          const int64_t uw = (int64_t)(uint64_t)U[k], vw = (int64_t)(uint64_t)V[k], mw = (int64_t)(uint64_t)M[k];
          accU += (int64_t)f0 * uw + (int64_t)g0 * vw + (int64_t)cu * mw;
          accV += (int64_t)f1 * uw + (int64_t)g1 * vw + (int64_t)cv * mw;
          Tu[k] = (uint32_t)accU; accU >>= 32;
          Tv[k] = (uint32_t)accV; accV >>= 32;
        }
#pragma unroll
        for (int k = 0; k < GCD_CH; k++) {
          if (w0 + k > 0) {
            u[(w0 + k - 1) * S] = ((k ? Tu[k - 1] : pU) >> GCD_K) | (Tu[k] << (32 - GCD_K));
            v[(w0 + k - 1) * S] = ((k ? Tv[k - 1] : pV) >> GCD_K) | (Tv[k] << (32 - GCD_K));
          }
        }
        pU = Tu[GCD_CH - 1]; pV = Tv[GCD_CH - 1];
      }
      // the sign word (index kw) of the two's complement cofactors
      const int64_t ut = (int64_t)(int32_t)u[kw * S], vt = (int64_t)(int32_t)v[kw * S];
      accU += (int64_t)f0 * ut + (int64_t)g0 * vt;
      accV += (int64_t)f1 * ut + (int64_t)g1 * vt;
      const uint32_t tu = (uint32_t)accU, tv = (uint32_t)accV;
      accU >>= 32; accV >>= 32;
      u[(kw - 1) * S] = (pU >> GCD_K) | (tu << (32 - GCD_K));
      v[(kw - 1) * S] = (pV >> GCD_K) | (tv << (32 - GCD_K));
      u[kw * S] = (tu >> GCD_K) | ((uint32_t)accU << (32 - GCD_K));
      v[kw * S] = (tv >> GCD_K) | ((uint32_t)accV << (32 - GCD_K));
    }
  }
  while (nb > 1 && b[(nb - 1) * S] == 0) nb--;
  return nb;
}

// gcd(x, N) == 1 for an odd N (x any value of kw words, zero included).  a, b: kw words of LDS each at stride S.
__device__ __forceinline__ bool wb_coprime_to_odd(const uint32_t* __restrict__ x, const uint32_t* __restrict__ N, int kw, uint32_t* a, uint32_t* b, int S) {
  for (int w = 0; w < kw; w++) { a[w * S] = x[w]; b[w * S] = N[w]; }
  const int nb = wb_gcd<false>(a, b, nullptr, nullptr, nullptr, 0u, kw, S);
  bool one = b[0] == 1;
  for (int w = 1; w < nb; w++) one = one && b[w * S] == 0;
  return one;
}

}  // namespace zkp
