// kernels_inv.hpp — modular inverse (binary extended GCD, one lane per item) and the small per-proof kernels of
// MulProof (multiplication_proof.rs) and CorrectMessageProof (correct_message.rs): SURVEY §8(f) rank 4.
// The exponentiations and products of these proofs run on the kernels of kernels_modexp.hpp; what is here is
// latency-class work (hashing, 256-bit sums, word compares) plus BigInt::mod_inv.
#pragma once
#include "kernels_proofs.hpp"

namespace zkp {

// ------------------------------------------------------------------------------------------
// out = a^-1 mod M (curv BigInt::mod_inv -> mpz_invert; multiplication_proof.rs:95,135, correct_message.rs:53,76,141).
// Word-batched binary extended GCD (kernels_gcd.hpp): ~258 rounds of two passes over 4096-bit operands; the cofactors are kept
// modulo M as two's complement numbers of kw + 1 words (|u|, |v| stay within a small multiple of M; the result is reduced at the end).  a, b, u, v and a copy of M live in thread-interleaved
// LDS (word w of lane t at base[w * LANES + t]: conflict-free).  Data-dependent trip counts: lanes of a wavefront wait for the
// slowest one (the counts differ by a few rounds).
struct ModinvArgs {
  const uint32_t* a; uint64_t a_stride;        // words between consecutive items
  const uint32_t* mod; uint64_t mod_stride;    // modulus words [kw] (0 = shared)
  uint32_t* out; uint64_t out_stride;
  uint8_t* status;                             // ZKP_INV_*
  uint64_t count; int kw;
};
constexpr size_t modinv_lds_words_per_lane(int kw) { return 5 * (size_t)kw + 2; }

__global__ void __launch_bounds__(64) k_modinv(ModinvArgs a) {
  extern __shared__ __align__(16) uint32_t inv_lds[];
  const int S = (int)blockDim.x, kw = a.kw;
  const uint64_t item = (uint64_t)blockIdx.x * S + threadIdx.x;
  if (item >= a.count) return;
  const uint32_t* A = a.a + item * a.a_stride;
  const uint32_t* M = a.mod + item * a.mod_stride;
  uint32_t* out = a.out + item * a.out_stride;
  uint32_t* pa = inv_lds + threadIdx.x;
  uint32_t* pb = pa + (size_t)kw * S;
  uint32_t* pu = pb + (size_t)kw * S;
  uint32_t* pv = pu + (size_t)(kw + 1) * S;
  uint32_t* pm = pv + (size_t)(kw + 1) * S;            // this lane's copy of the modulus
  // domain: M odd, M >= 3, 0 <= a < M
  int cmp = 0;
  bool a_zero = true, m_small = M[0] < 3;
  for (int w = kw - 1; w >= 0; w--) {
    const uint32_t aw = A[w], mw = M[w];
    if (cmp == 0 && aw != mw) cmp = aw > mw ? 1 : -1;
    a_zero = a_zero && aw == 0;
    if (w > 0) m_small = m_small && mw == 0;
  }
  for (int w = 0; w < kw; w++) out[w] = 0;
  if (!(M[0] & 1) || m_small || cmp >= 0) { a.status[item] = ZKP_INV_DOMAIN; return; }
  if (a_zero) { a.status[item] = ZKP_INV_NONE; return; }
  uint32_t minv = M[0];
#pragma unroll
  for (int it = 0; it < 5; it++) minv *= 2u - M[0] * minv;
  const uint32_t minv30 = (0u - minv) & GCD_MK;        // -M^-1 mod 2^30
  // a == A * u, b == A * v (mod M) with (a, b, u, v) = (A, M, 1, 0)
  for (int w = 0; w < kw; w++) { pa[w * S] = A[w]; pb[w * S] = M[w]; pm[w * S] = M[w]; pu[w * S] = w == 0; pv[w * S] = 0; }
  pu[kw * S] = 0; pv[kw * S] = 0;
  const int nb = wb_gcd<true>(pa, pb, pu, pv, pm, minv30, kw, S);
  bool one = pb[0] == 1;
  for (int w = 1; w < nb; w++) one = one && pb[w * S] == 0;
  if (!one) { a.status[item] = ZKP_INV_NONE; return; }
  // 1 == A * v (mod M).  The balanced correction of a round only keeps |v'| <= max(|u|, |v|) + M/2, not |v| < M: values a little
  // above M in magnitude occur (a = M - 14: v = -1.07 M), so the residue is reduced properly — add M while v is negative, then
  // subtract M while v >= M, over the kw words and the sign word (a handful of passes at most; each one moves v towards [0, M)).
  for (;;) {
    const int32_t top = (int32_t)pv[kw * S];
    bool neg = top < 0, ge = top > 0;
    if (top == 0) {                                    // 0 <= v < 2^(32 kw): compare with M
      ge = true;                                       // (v == M cannot happen for gcd 1, but would reduce to 0)
      for (int w = kw - 1; w >= 0; w--) {
        const uint32_t vw = pv[w * S], mw = pm[w * S];
        if (vw != mw) { ge = vw > mw; break; }
      }
    }
    if (!neg && !ge) break;
    if (neg) {
      uint32_t carry = 0;
      for (int w = 0; w < kw; w++) { const uint64_t t = (uint64_t)pv[w * S] + pm[w * S] + carry; pv[w * S] = (uint32_t)t; carry = (uint32_t)(t >> 32); }
      pv[kw * S] += carry;
    } else {
      uint32_t borrow = 0;
      for (int w = 0; w < kw; w++) { const uint64_t t = (uint64_t)pv[w * S] - pm[w * S] - borrow; pv[w * S] = (uint32_t)t; borrow = (uint32_t)(t >> 63); }
      pv[kw * S] -= borrow;
    }
  }
  for (int w = 0; w < kw; w++) out[w] = pv[w * S];
  a.status[item] = ZKP_INV_OK;
}

// n -> n^2 as 32-bit words (one thread per key; schoolbook, kw x kw words)
__global__ void __launch_bounds__(64) k_square_words(const uint32_t* __restrict__ n, uint64_t n_stride, int kw, uint64_t count, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t* s = n + i * n_stride;
  uint32_t* o = out + i * 2 * kw;
  for (int w = 0; w < 2 * kw; w++) o[w] = 0;
  for (int p = 0; p < kw; p++) {
    uint64_t carry = 0;
    const uint64_t ap = s[p];
    for (int q = 0; q < kw; q++) {
      const uint64_t t = ap * s[q] + o[p + q] + carry;
      o[p + q] = (uint32_t)t;
      carry = t >> 32;
    }
    o[p + kw] = (uint32_t)carry;
  }
}

// ------------------------------------------------------------------------------------------
// e = SHA-256 over a list of BigInt operands (compute_digest, utils.rs:9-22), little-endian words of the digest
// taken as a number.  Operand k contributes reps values per proof: p + b*item_stride + r*rep_stride, `words` each.
struct HashOp { const uint32_t* p; uint64_t item_stride; uint64_t rep_stride; uint32_t words; uint32_t reps; };
struct HashListArgs { HashOp op[8]; int nops; uint64_t batch; uint32_t* e; };

__global__ void __launch_bounds__(256) k_hash_list(HashListArgs a) {
  __shared__ uint32_t shabuf[16 * 256];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  Sha256 s;
  s.init(shabuf + threadIdx.x, 256);
  for (int k = 0; k < a.nops; k++) {
    const HashOp& o = a.op[k];
    for (uint32_t r = 0; r < o.reps; r++) s.put_bigint(o.p + b * o.item_stride + r * o.rep_stride, (int)o.words);
  }
  uint32_t d[8];
  s.finish(d);
#pragma unroll
  for (int k = 0; k < 8; k++) a.e[b * 8 + k] = d[7 - k];
}

// out = (x + y) mod n for x, y < n (kw words, one thread per item): BigInt::mod_add, multiplication_proof.rs:88
__global__ void __launch_bounds__(256) k_modadd(const uint32_t* __restrict__ x, const uint32_t* __restrict__ y, const uint32_t* __restrict__ n,
                                                uint64_t n_stride, int kw, uint64_t batch, uint32_t* __restrict__ out) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  const uint32_t *xp = x + b * kw, *yp = y + b * kw, *np = n + b * n_stride;
  uint32_t* o = out + b * kw;
  uint32_t cy = 0;
  for (int w = 0; w < kw; w++) { const uint64_t s = (uint64_t)xp[w] + yp[w] + cy; o[w] = (uint32_t)s; cy = (uint32_t)(s >> 32); }
  int ge = cy ? 1 : 0;
  if (!ge) {
    ge = 1;
    for (int w = kw - 1; w >= 0; w--) if (o[w] != np[w]) { ge = o[w] > np[w]; break; }
  }
  if (ge) {
    uint32_t bw = 0;
    for (int w = 0; w < kw; w++) { const uint64_t d = (uint64_t)o[w] - np[w] - bw; o[w] = (uint32_t)d; bw = (uint32_t)(d >> 63); }
  }
}

// MulProof::prove epilogue: a failed mod_inv (the reference's unwrap panic, :95) blanks f, z1, z2 of that proof
struct MulFinishArgs { const uint8_t* inv_status; const uint32_t* consts; uint64_t const_stride; int st_off; int kw; uint64_t batch;
                       uint32_t* f; uint32_t* z1; uint32_t* z2; uint8_t* status; };
__global__ void __launch_bounds__(256) k_mul_finish(MulFinishArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const bool bad = a.inv_status[b] != ZKP_INV_OK || a.consts[b * a.const_stride + a.st_off] != 0;
  a.status[b] = bad ? ZKP_VERDICT_MALFORMED : 0;
  if (!bad) return;
  for (int w = 0; w < a.kw; w++) a.f[b * a.kw + w] = 0;
  for (int w = 0; w < 2 * a.kw; w++) { a.z1[b * 2 * a.kw + w] = 0; a.z2[b * 2 * a.kw + w] = 0; }
}

// MulProof::verify: `e_a_e_e_d == enc_f_z1 && e_b_f_e_db_e_c_e_inv == enc_0_z2` (:139-145)
struct MulVerdictArgs { const uint32_t *l1, *c1, *l2, *c2; const uint8_t* inv_status; const uint32_t* consts; uint64_t const_stride; int st_off;
                        uint32_t words; uint64_t batch; uint8_t* verdict; };
__global__ void __launch_bounds__(256) k_mul_verdict(MulVerdictArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  bool same = true;
  for (uint32_t w = 0; w < a.words; w++) same = same && a.l1[b * a.words + w] == a.c1[b * a.words + w] && a.l2[b * a.words + w] == a.c2[b * a.words + w];
  uint8_t v = same ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT;
  if (a.inv_status[b] != ZKP_INV_OK || a.consts[b * a.const_stride + a.st_off] != 0) v = ZKP_VERDICT_MALFORMED;
  a.verdict[b] = v;
}

// ------------------------------------------------------------------------------------------
// CorrectMessageProof (correct_message.rs:35-162).  A proof has K "rows" (one per valid message); the row-level
// exponentiations, products and inverses run as batches of B*K items, the kernels below do the bookkeeping.

// out[row] = src[row / K]   (per-proof values seen per row: n, ciphertext)
__global__ void __launch_bounds__(256) k_repeat_rows(const uint32_t* __restrict__ src, uint64_t src_stride, uint32_t words, uint32_t K, uint64_t rows,
                                                     uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * words) return;
  const uint64_t row = i / words, w = i % words;
  out[i] = src[(row / K) * src_stride + w];
}

// v += 1 in place (gm = m*n + 1: the product is a multiple of n below n^2, so the sum stays below n^2; :51, :137)
__global__ void __launch_bounds__(256) k_add_one(uint32_t* __restrict__ v, uint32_t words, uint64_t rows) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  uint32_t* p = v + r * words;
  for (uint32_t w = 0; w < words; w++) { if (++p[w] != 0) break; }
}

// prove, per proof: which rows hold the encrypted message (:68, :100, :113), which simulated (e, z) pair the other rows
// take (the running index j of :66-80), and the reference's index-out-of-bounds panic when no row matches.
// row_sim[row] = j for a simulated row, 0xFFFFFFFF for a real one.
struct CmPlanArgs { const uint32_t* valid; const uint32_t* message; uint32_t kw, K; uint64_t batch; uint32_t* row_sim; uint8_t* panic; };
__global__ void __launch_bounds__(256) k_cm_plan(CmPlanArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  uint32_t j = 0;
  bool panic = false;
  for (uint32_t i = 0; i < a.K; i++) {
    const uint32_t* m = a.valid + (b * a.K + i) * a.kw;
    bool same = true;
    for (uint32_t w = 0; w < a.kw; w++) same = same && m[w] == a.message[b * a.kw + w];
    if (same) a.row_sim[b * a.K + i] = 0xFFFFFFFFu;
    else { if (j >= a.K - 1) panic = true; a.row_sim[b * a.K + i] = j++; }
  }
  a.panic[b] = panic ? 1 : 0;
}

// prove, per row: base = w (real row) or z_sim[j]; exponent = 0 (real row: u^0 = 1, whose inverse is 1, so that
// a_i = w^n comes out of the same product as the simulated rows' z^n * (u^e)^-1, :69-77) or e_sim[j]
struct CmGatherArgs { const uint32_t* row_sim; const uint8_t* panic; const uint32_t* w; const uint32_t* z_sim; const uint32_t* e_sim; uint32_t kw, K; uint64_t rows;
                      uint32_t* base; uint32_t* exp; };
__global__ void __launch_bounds__(256) k_cm_gather(CmGatherArgs a) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.rows) return;
  const uint64_t b = r / a.K;
  const uint32_t j = a.row_sim[r];
  const bool real = j == 0xFFFFFFFFu || a.panic[b];      // a panicking proof is blanked at the end; keep its rows harmless
  const uint32_t* bs = real ? a.w + b * a.kw : a.z_sim + (b * (a.K - 1) + j) * a.kw;
  for (uint32_t w = 0; w < a.kw; w++) a.base[r * a.kw + w] = bs[w];
  for (uint32_t w = 0; w < 8; w++) a.exp[r * 8 + w] = real ? 0u : a.e_sim[(b * (a.K - 1) + j) * 8 + w];
}

// prove, per proof: ei = (chal - sum of ALL K-1 simulated challenges) mod 2^256 (:87-93)
__global__ void __launch_bounds__(256) k_cm_ei(const uint32_t* __restrict__ chal, const uint32_t* __restrict__ e_sim, uint32_t K, uint64_t batch, uint32_t* __restrict__ ei) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  uint32_t acc[8];
  for (int w = 0; w < 8; w++) acc[w] = chal[b * 8 + w];
  for (uint32_t k = 0; k + 1 < K; k++) {
    uint32_t bw = 0;
    for (int w = 0; w < 8; w++) { const uint64_t d = (uint64_t)acc[w] - e_sim[(b * (K - 1) + k) * 8 + w] - bw; acc[w] = (uint32_t)d; bw = (uint32_t)(d >> 63); }
  }
  for (int w = 0; w < 8; w++) ei[b * 8 + w] = acc[w];
}

// prove, per proof: e_vec / z_vec rows from (ei, zi) or the simulated pairs (:97-121); a proof whose reference run
// panics (no matching row, or a mod_inv without result) is blanked
struct CmScatterArgs { const uint32_t* row_sim; const uint8_t* panic; const uint8_t* inv1; const uint8_t* inv2; const uint32_t* consts; uint64_t const_stride; int st_off;
                       const uint32_t* ei; const uint32_t* zi; const uint32_t* e_sim; const uint32_t* z_sim; uint32_t kw, K; uint64_t batch;
                       uint32_t* e_vec; uint32_t* z_vec; uint32_t* a_vec; uint8_t* status; };
__global__ void __launch_bounds__(256) k_cm_scatter(CmScatterArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  bool bad = a.panic[b] != 0;
  for (uint32_t i = 0; i < a.K; i++) bad = bad || a.inv1[b * a.K + i] != ZKP_INV_OK || a.inv2[b * a.K + i] != ZKP_INV_OK || a.consts[(b * a.K + i) * a.const_stride + a.st_off] != 0;
  a.status[b] = bad ? ZKP_VERDICT_MALFORMED : 0;
  for (uint32_t i = 0; i < a.K; i++) {
    const uint64_t r = b * a.K + i;
    const uint32_t j = a.row_sim[r];
    for (uint32_t w = 0; w < 8; w++) a.e_vec[r * 8 + w] = bad ? 0u : (j == 0xFFFFFFFFu ? a.ei[b * 8 + w] : a.e_sim[(b * (a.K - 1) + j) * 8 + w]);
    for (uint32_t w = 0; w < a.kw; w++) a.z_vec[r * a.kw + w] = bad ? 0u : (j == 0xFFFFFFFFu ? a.zi[b * a.kw + w] : a.z_sim[(b * (a.K - 1) + j) * a.kw + w]);
    if (bad) for (uint32_t w = 0; w < 2 * a.kw; w++) a.a_vec[r * 2 * a.kw + w] = 0;
  }
}

// verify, per proof: assert_eq!(chal, sum of e_vec mod 2^256) (:126-132) -> MALFORMED; every row's
// u^e * a == z^n (:144-156) -> ACCEPT / REJECT
struct CmVerdictArgs { const uint32_t* chal; const uint32_t* e_vec; const uint32_t* lhs; const uint32_t* rhs; const uint8_t* inv1; const uint32_t* consts;
                       uint64_t const_stride; int st_off; uint32_t kw, K; uint64_t batch; uint8_t* verdict; };
__global__ void __launch_bounds__(256) k_cm_verdict(CmVerdictArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool same = true, bad = false;
  for (uint32_t i = 0; i < a.K; i++) {
    const uint64_t r = b * a.K + i;
    uint32_t cy = 0;
    for (int w = 0; w < 8; w++) { const uint64_t s = (uint64_t)acc[w] + a.e_vec[r * 8 + w] + cy; acc[w] = (uint32_t)s; cy = (uint32_t)(s >> 32); }
    for (uint32_t w = 0; w < 2 * a.kw; w++) same = same && a.lhs[r * 2 * a.kw + w] == a.rhs[r * 2 * a.kw + w];
    bad = bad || a.inv1[r] != ZKP_INV_OK || a.consts[r * a.const_stride + a.st_off] != 0;
  }
  for (int w = 0; w < 8; w++) bad = bad || acc[w] != a.chal[b * 8 + w];
  a.verdict[b] = bad ? ZKP_VERDICT_MALFORMED : (same ? ZKP_VERDICT_ACCEPT : ZKP_VERDICT_REJECT);
}

}  // namespace zkp
